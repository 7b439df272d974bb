#!/bin/bash
# the transforms' decode under several builds: tools/tr_ab.sh variants/a.so ...   ("" = the shipped library)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for L in "$@"; do
  if [ "$L" != default ]; then export SPRINTZ_MI355X_LIB=$PWD/sprintz_amd/$L; else unset SPRINTZ_MI355X_LIB; fi
  timeout 200 python tools/transforms_ab.py 20 2>&1 | tail -4
done
