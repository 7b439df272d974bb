cd $GRAFT_REPO_ROOT
python tools/ab.py --cfg headline --rounds 3 --no-verify base=sprintz_amd/variants/base.so ablwin=sprintz_amd/variants/ablwin.so st0=sprintz_amd/variants/st0.so st1=sprintz_amd/variants/st1.so st3=sprintz_amd/variants/st3.so ablstore=sprintz_amd/variants/ablstore.so 2>&1 | grep -E "MEDIAN" | awk '{print $1,$2,$3,$4,$5}'
