cd $GRAFT_REPO_ROOT
python tools/ab.py --cfg cfg4_800000 --rounds 2 --reps 8 --no-verify base=sprintz_amd/variants/base.so h_nostore=sprintz_amd/variants/h_nostore.so h_plain=sprintz_amd/variants/h_plain.so 2>&1 | grep -E "round" | awk '{print $3, $5, $9}'
