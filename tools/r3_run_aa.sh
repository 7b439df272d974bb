cd $GRAFT_REPO_ROOT
python tools/ab.py --cfg cfg1 --rounds 3 --no-verify base=sprintz_amd/variants/base.so uni_nostore=sprintz_amd/variants/uni_nostore.so 2>&1 | grep MEDIAN
