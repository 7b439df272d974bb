#!/usr/bin/env python3
"""where huf_build_kernel's first workgroup spends its time (a build with -DHUF_BUILD_TIMING:
tools/build_variant.sh build_timing huf -DHUF_BUILD_TIMING; SPRINTZ_MI355X_LIB=sprintz_amd/variants/build_timing.so python tools/build_phases.py [chunks])"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sprintz_amd  # noqa: E402
from synth import synth_torch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
x = synth_torch("walk", 2, n, 640, 8, "cuda:0", seed=123, step=8)
cd = sprintz_amd.ChunkedCodec("xff", 2, 8, 5120, device="cuda:0")
b = cd.compress(x)
for _ in range(3):
    sprintz_amd.huf0_compress(b)
torch.cuda.synchronize()
lib = C.CDLL(os.environ.get("SPRINTZ_MI355X_LIB", os.path.join(ROOT, "sprintz_amd", "libsprintz_mi355x.so")))
fn = getattr(lib, "sprintz_mi355x_dbg_build_stamps", None)
if fn is None:
    sys.exit("this build has no stamps: build with -DHUF_BUILD_TIMING")
ts = np.zeros(16, np.uint64)
fn(ts.ctypes.data_as(C.c_void_p))
names = ["start", "histogram", "rank sort", "two-queue merge", "depth walk", "length-limit repair", "codes + stores"]
for k in range(1, 7):
    print(f"{names[k]:22s} {(int(ts[k]) - int(ts[k - 1])) / 100.0:8.2f} us")
print(f"{'total':22s} {(int(ts[6]) - int(ts[0])) / 100.0:8.2f} us")
