#!/bin/bash
# tools/kernel_regs.sh <unit> [name-filter]: VGPR / SGPR / spill counts of the gfx950 kernels in sprintz_amd/csrc/build/<unit>.o
# (unbundles the device code object into a scratch directory and reads its metadata notes)
set -e
unit=$1; filt=${2:-.}
here=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
cp "$here/sprintz_amd/csrc/build/$unit.o" "$tmp/u.o"
(cd "$tmp" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading u.o >/dev/null 2>&1 && /opt/rocm/lib/llvm/bin/llvm-readelf --notes u.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 > notes.txt)
python3 - "$tmp/notes.txt" "$filt" <<'PY'
import re, sys
t = open(sys.argv[1]).read()
for b in t.split('- .agpr_count')[1:]:
    nm = re.search(r'\.name:\s*(\S+)', b).group(1)
    if re.search(sys.argv[2], nm):
        g = lambda k: re.search(r'\.%s:\s*(\d+)' % k, b).group(1)
        print(nm[:110], 'vgpr', g('vgpr_count'), 'sgpr', g('sgpr_count'), 'spill', g('vgpr_spill_count'))
PY
rm -rf "$tmp"
