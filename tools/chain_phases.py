#!/usr/bin/env python
"""where a tile of the one-pass transform decode spends its time: needs a -DTR_CHAIN_TIMING build (tools/build_variant.sh chain_ts transforms
-DTR_CHAIN_TIMING; SPRINTZ_MI355X_LIB=.../variants/chain_ts.so).  Stamps (10 ns ticks): 0 ticket, 1 tile in registers, 2 wave summaries,
3 tile summary out, 4 state in (look-back done, state out), 5 rows stored."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import sprintz_amd
from sprintz_amd import _lib
from synth import synth_torch

kind = sys.argv[1] if len(sys.argv) > 1 else "delta"
dev = torch.device("cuda:0")
D, rows = 8, (64 << 20) // 8
x = synth_torch("walk", 2, 1, rows, D, dev, seed=123, step=8).reshape(-1)
y, back = torch.empty_like(x), torch.empty_like(x)
sprintz_amd.transform_device(kind, x, D, out=y)
for _ in range(5):
    sprintz_amd.transform_device(kind, y, D, inverse=True, out=back)
torch.cuda.synchronize()
lib = C.CDLL(os.environ["SPRINTZ_MI355X_LIB"])
nt = int(os.environ.get("TILES", "1024"))
buf = np.zeros((nt, 8), np.uint64)
assert lib.sprintz_mi355x_dbg_chain_stamps(buf.ctypes.data_as(C.c_void_p), C.c_uint32(nt)) == 0
t = buf[:, :6].astype(np.int64)
t0 = t[:, 0].min()
print("tiles", nt, "span %.1f us" % ((t[:, 5].max() - t0) / 100.0))
names = ["load", "fold", "publish", "look-back", "store"]
for k in range(5):
    d = (t[:, k + 1] - t[:, k]) / 100.0
    print("%-10s mean %6.2f us  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (names[k], d.mean(), *np.percentile(d, [10, 50, 90]), d.max()))
life = (t[:, 5] - t[:, 0]) / 100.0
print("tile life  mean %6.2f us; mean tiles in flight %.0f" % (life.mean(), life.sum() / ((t[:, 5].max() - t0) / 100.0)))
st = (t[:, 0] - t0) / 100.0
print("ticket times (us) of tiles 0, 128, 256, 384, 512, 768, last:", [round(float(st[i]), 1) for i in (0, 128, 256, 384, 512, 768, nt - 1) if i < nt])
