#!/bin/bash
# an experiment build of the library next to the shipped one:
#   tools/build_variant.sh <name> "<units to recompile, e.g. decode_w16 encode_w16>" [-DFLAG ...]
# Objects of every other unit are taken from sprintz_amd/csrc/build_base (built once, no flags).
# -> sprintz_amd/variants/<name>.so (git-ignored; travels to the GPU box; select with SPRINTZ_MI355X_LIB or tools/ab.py)
set -e
cd "$(dirname "$0")/.."
name=$1; units=$2; shift 2
mkdir -p sprintz_amd/variants
C=sprintz_amd/csrc
[ -d $C/build_base ] || make -s -C $C -j8 OBJDIR=build_base OUT=../variants/base.so
if [ "$name" != base ]; then
  rm -rf $C/build_$name; mkdir -p $C/build_$name
  cp $C/build_base/*.o $C/build_$name/
  for u in $units; do rm -f $C/build_$name/$u.o; done
  make -s -C $C -j8 OBJDIR=build_$name OUT=../variants/$name.so EXTRA="$*"
fi
ls -la sprintz_amd/variants/$name.so
