cd $GRAFT_REPO_ROOT
python tools/ab.py --cfg headline --rounds 3 --no-verify base=sprintz_amd/variants/base.so ab_il=sprintz_amd/variants/ab_il.so ablstore=sprintz_amd/variants/ablstore.so 2>&1 | grep -E "MEDIAN" | awk '{print $1,$2,$3,$4,$5}'
