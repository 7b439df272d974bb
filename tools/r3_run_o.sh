cd $GRAFT_REPO_ROOT
python tools/ab.py --cfg headline --rounds 3 --no-verify base=sprintz_amd/variants/base.so ablstage=sprintz_amd/variants/ablstage.so ablfetch=sprintz_amd/variants/ablfetch.so ablboth=sprintz_amd/variants/ablboth.so ablstore=sprintz_amd/variants/ablstore.so 2>&1 | grep -E "round|MEDIAN" | awk '{print $1,$2,$3,$4,$5}'
