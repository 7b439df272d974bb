#!/usr/bin/env python3
"""where huf0_tree_wave_kernel's first wave spends its time (a build with -DHUF0_TREE_TIMING:
SPRINTZ_MI355X_LIB=sprintz_amd/variants/tree_timing.so python tools/tree_phases.py [chunks])"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sprintz_amd  # noqa: E402
from sprintz_amd import _lib  # noqa: E402
from synth import synth_torch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
x = synth_torch("walk", 2, n, 640, 8, "cuda:0", seed=123, step=8)
cd = sprintz_amd.ChunkedCodec("xff", 2, 8, 5120, device="cuda:0")
b = cd.compress(x)
gb, gbo = sprintz_amd.huf0_compress(b)
goo = torch.zeros(n + 1, dtype=torch.int64, device="cuda:0")
goo[1:] = torch.cumsum(b.sizes.to(torch.int64), 0)
rets = torch.empty(n, dtype=torch.int64, device="cuda:0")
st = sprintz_amd.huf0_decompress(gb, gbo, goo, rets=rets)
for _ in range(5):
    sprintz_amd.huf0_decompress(gb, gbo, goo, out=st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    sprintz_amd.huf0_decompress(gb, gbo, goo, out=st)
e1.record()
torch.cuda.synchronize()
print(f"huff0 stage, {n} chunks: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
fn = getattr(_lib.lib, "sprintz_mi355x_dbg_tree_stamps", None) if hasattr(_lib, "lib") else None
if fn is None:
    lib = C.CDLL(os.environ.get("SPRINTZ_MI355X_LIB", os.path.join(ROOT, "sprintz_amd", "libsprintz_mi355x.so")))
    fn = getattr(lib, "sprintz_mi355x_dbg_tree_stamps", None)
if fn is None:
    sys.exit("not a HUF0_TREE_TIMING build")
ts = np.zeros(16, np.uint64)
fn(C.c_void_p(ts.ctypes.data))
names = ["offsets + header copy", "FSE_readNCount", "FSE_buildDTable", "weight decode", "followers' compare", "statistics", "counting sort", "descriptor stores"]
for k, nm in enumerate(names):
    print(f"{nm:24s} {(int(ts[k + 1]) - int(ts[k])) * 10:7d} ns")
print(f"{'total':24s} {(int(ts[8]) - int(ts[0])) * 10:7d} ns")

