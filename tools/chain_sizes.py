#!/usr/bin/env python
"""the one-pass decoders against the forms they replace over STREAM SIZES (the switch-over point): delta / double delta decode of uint16 x 8 and
dynamic-delta unpack, 0.5 MB .. 128 MB streams; us"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import sprintz_amd
from sprintz_amd import _lib

dev = torch.device("cuda:0")
w = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
for _ in range(200):
    w.add_(1)
torch.cuda.synchronize()
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    return best * 1e3


def both(env, fn):
    res = []
    for v in ("1", "0"):
        os.environ[env] = v
        res.append(timed(fn))
    os.environ.pop(env, None)
    return res


g = torch.Generator(device="cuda")
g.manual_seed(3)
for mb in [float(v) for v in os.environ.get("SIZES_MB", "0.5,1,2,4,8,16,32,64,128").split(",")]:
    n = int(mb * (1 << 20)) // 2
    x = torch.randint(0, 1 << 16, (n,), generator=g, device="cuda", dtype=torch.int32).to(torch.uint16)
    line = "%6.1f MB " % mb
    for kind in ("delta", "doubledelta"):
        y = sprintz_amd.transform_device(kind, x, 8)
        back = torch.empty_like(x)
        tmp = torch.empty(int(_lib.transform_tmp_bytes(0 if kind == "delta" else 1, 2, n, 8)), dtype=torch.uint8, device=dev)
        k = 0 if kind == "delta" else 1
        a, b = both("SPRINTZ_MI355X_TRANSFORM_CHAIN", lambda: _lib.check(_lib.transform_decode_device(k, 2, y.data_ptr(), n, 8, back.data_ptr(), tmp.data_ptr(), st)))
        line += " %s: one pass %6.1f  two %6.1f |" % (kind, a, b)
    dest = torch.zeros(int(_lib.online_bound(0, n)) + 64, dtype=torch.uint8, device=dev)
    tmp = torch.empty(int(_lib.online_tmp_bytes(0, n)) + 64, dtype=torch.uint8, device=dev)
    ret = torch.zeros(1, dtype=torch.int64, device=dev)
    out = torch.zeros(n + 16, dtype=torch.int16, device=dev)
    xs = x.view(torch.int16)
    _lib.check(_lib.online_pack_device(0, xs.data_ptr(), n, dest.data_ptr(), ret.data_ptr(), tmp.data_ptr(), st))
    a, b = both("SPRINTZ_MI355X_ONLINE_CHAIN", lambda: _lib.check(_lib.online_unpack_device(0, dest.data_ptr(), n, out.data_ptr(), ret.data_ptr(), tmp.data_ptr(), st)))
    line += " dyn-delta unpack: one pass %6.1f  three launches %6.1f" % (a, b)
    print(line, flush=True)
