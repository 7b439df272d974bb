#!/usr/bin/env python
"""time the online coders' unpack (and pack) on the bench's stream for the library SPRINTZ_MI355X_LIB names: tools/online_ab.py [reps [kind]]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from sprintz_amd import _lib
from synth import synth_torch

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kinds = [int(sys.argv[2])] if len(sys.argv) > 2 else [0, 3]
names = {0: "dynamic_delta", 1: "dynamic_delta_alt", 2: "zigzag", 3: "sprintzpack", 4: "sprintzpack_zigzag"}
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
w = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
for _ in range(200):
    w.add_(1)
torch.cuda.synchronize()


def timed(fn):
    for _ in range(3):
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
    return best


for n in (64 << 20, 512 << 20):
    x = synth_torch("walk", 2, 1, n, 1, dev, seed=123, step=8).reshape(-1)
    for kind in kinds:
        dest = torch.zeros(int(_lib.online_bound(kind, n)) + 64, dtype=torch.uint8, device=dev)
        tmp = torch.empty(int(_lib.online_tmp_bytes(kind, n)) + 64, dtype=torch.uint8, device=dev)
        ret = torch.zeros(1, dtype=torch.int64, device=dev)
        out = torch.zeros(n + 16, dtype=torch.int16, device=dev)
        pk = lambda: _lib.check(_lib.online_pack_device(kind, x.data_ptr(), n, dest.data_ptr(), ret.data_ptr(), tmp.data_ptr(), st))
        up = lambda: _lib.check(_lib.online_unpack_device(kind, dest.data_ptr(), n, out.data_ptr(), ret.data_ptr(), tmp.data_ptr(), st))
        tp = timed(pk)
        r = int(ret.item())
        tu = timed(up)
        ok = int(ret.item()) == n and torch.equal(out[:n].view(torch.uint16), x.view(torch.uint16))
        byt = 2 * n + 2 * r
        print("%-18s %4d Mi  pack %.4f ms frac %.3f  unpack %.4f ms frac %.3f %s" % (names[kind], n >> 20, tp, byt / (tp * 1e-3) / 8e12, tu, byt / (tu * 1e-3) / 8e12, "ok" if ok else "MISMATCH"), flush=True)
        del dest, tmp, out
    del x
