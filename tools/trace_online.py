#!/usr/bin/env python3
"""the online coders' unpack / pack launches on 64 Mi samples, for a kernel trace:
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tt -o r -- python tools/trace_online.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch  # noqa: E402

from sprintz_amd import _lib  # noqa: E402
from synth import synth_torch  # noqa: E402

dev = torch.device("cuda", 0)
n = 64 << 20
x = synth_torch("walk", 2, 1, n, 1, dev, seed=123, step=8)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for kind in (0, 3):
    dest = torch.zeros(int(_lib.online_bound(kind, n)) + 64, dtype=torch.uint8, device=dev)
    tmp = torch.zeros(int(_lib.online_tmp_bytes(kind, n)) + 64, dtype=torch.uint8, device=dev)
    back = torch.empty(n, dtype=torch.uint16, device=dev)
    ret = torch.zeros(2, dtype=torch.int64, device=dev)
    for _ in range(10):
        _lib.check(_lib.online_pack_device(kind, x.data_ptr(), n, dest.data_ptr(), ret.data_ptr(), tmp.data_ptr(), st))
        _lib.check(_lib.online_unpack_device(kind, dest.data_ptr(), n, back.data_ptr(), ret.data_ptr() + 8, tmp.data_ptr(), st))
    torch.cuda.synchronize()
