#!/usr/bin/env python3
"""decode and encode rate of one shape at several chunk lengths: python tools/bench_shape.py codec esz D chunk_len[,chunk_len...] [mb]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import sprintz_amd
from synth import synth_torch
codec, esz, D = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lens = [int(v) for v in sys.argv[4].split(",")]
mb = int(sys.argv[5]) if len(sys.argv) > 5 else 512
dev = torch.device("cuda", 0)
for cl in lens:
    rows = cl // D
    cl = rows * D
    n = max(256, (mb << 20) // (cl * esz))
    x = synth_torch("walk", esz, n, rows, D, dev, step=2 if esz == 1 else 8)
    if esz == 2: x = x.view(torch.int16)
    cd = sprintz_amd.ChunkedCodec(codec, esz, D, cl, device=dev)
    b = cd.compress(x)
    out = torch.empty(n * cl, dtype=x.dtype, device=dev)
    cd.decompress_into(b.data, b.offsets, n, out); torch.cuda.synchronize()
    assert torch.equal(out, x)
    for _ in range(3): cd.decompress_into(b.data, b.offsets, n, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): cd.decompress_into(b.data, b.offsets, n, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    raw = n * cl * esz
    # the write path: encoder alone (into slots), and encode + container (what the bench times)
    src = cd._padded_view(x.contiguous())
    ws = cd.workspace(n)
    ws, dense, offs = cd.compress_dense(src, n * cl, ws)
    def timed(f):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20
    enc_ms = timed(lambda: cd.compress_to_slots(src, n * cl, ws))
    wr_ms = timed(lambda: cd.compress_dense(src, n * cl, ws, dense, offs))
    print(f"{codec} u{8*esz} D={D} chunk {cl*esz:7d} B x {n:7d}: ratio {raw / b.stream_bytes():.3f} decode {ms:.4f} ms = {raw / ms / 1e6:.0f} GB/s raw, {(raw + b.stream_bytes()) / ms / 1e6 / 8000:.3f} of HBM peak; "
          f"encode {enc_ms:.4f} ms, encode + container {wr_ms:.4f} ms = {(raw + b.stream_bytes()) / wr_ms / 1e6 / 8000:.3f}", flush=True)
    del ws, dense, offs, src
    del x, b, out; cd._ws = {}; torch.cuda.empty_cache()
