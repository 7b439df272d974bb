#!/usr/bin/env python3
"""Throughput of the batched codec over the BASELINE.json configurations and a few
stress shapes (not the headline bench: that is bench.py).  Prints a markdown table.
    python tools/bench_configs.py [--mb 512]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import sprintz_amd  # noqa: E402
from sprintz_amd import _lib  # noqa: E402


def make(kind, nchunks, rows, ndims, esz, device, seed=1):
    g = torch.Generator(device=device).manual_seed(seed)
    top = 1 << (8 * esz)
    if kind == "uniform":
        x = torch.randint(0, top, (nchunks, rows, ndims), device=device, generator=g, dtype=torch.int32)
    else:
        step = {"walk8": 8, "walk2": 2, "walk300": 300, "walkflat": 8}[kind]
        x = torch.randint(-step, step + 1, (nchunks, rows, ndims), device=device, generator=g, dtype=torch.int32)
        if kind == "walkflat":
            x[:, (torch.arange(rows, device=device) // 64) % 2 == 0] = 0
        x = torch.cumsum(x, dim=1, dtype=torch.int32) + torch.randint(0, top, (nchunks, 1, ndims), device=device, generator=g, dtype=torch.int32)
    x = x & (top - 1)
    if esz == 1:
        return x.to(torch.uint8).reshape(-1)
    return torch.where(x >= 32768, x - 65536, x).to(torch.int16).reshape(-1)


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=512, help="raw MB per configuration")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfgs = [
        ("cfg1  u8  D=1  delta 1KB (low-dim)", "delta", 1, 1, 1024, "walk2"),
        ("cfg2  u16 D=8  xff 10KB walk8", "xff", 2, 8, 5120, "walk8"),
        ("cfg2  u16 D=8  xff 10KB uniform", "xff", 2, 8, 5120, "uniform"),
        ("cfg2  u16 D=8  xff 10KB walk+runs", "xff", 2, 8, 5120, "walkflat"),
        ("      u16 D=8  delta 10KB walk8", "delta", 2, 8, 5120, "walk8"),
        ("      u8  D=8  xff 8KB walk2", "xff", 1, 8, 8192, "walk2"),
        ("cfg3  u8  D=80 delta 1KB (raw passthrough)", "delta", 1, 80, 1024 + 16, "walk2"),
        ("cfg3  u8  D=80 delta 10KB", "delta", 1, 80, 10240, "walk2"),
        ("      u16 D=80 xff 20KB (MSRC-12 shape at 16 bits)", "xff", 2, 80, 10240, "walk8"),
        ("cfg5  u16 D=32 xff 10KB", "xff", 2, 32, 5120, "walk8"),
        ("      u16 D=16 xff 10KB", "xff", 2, 16, 5120, "walk8"),
        ("      u16 D=64 xff 16KB", "xff", 2, 64, 8192, "walk8"),
        ("      u16 D=8  delta, run-less codec 10KB", "delta_norle", 2, 8, 5120, "walk8"),
        ("      u8  D=8  xff, run-less codec 8KB", "xff_norle", 1, 8, 8192, "walk2"),
        ("      u16 D=2  xff 8KB (low-dim)", "xff", 2, 2, 4096, "walk8"),
        ("      u8  D=4  xff 4KB (low-dim)", "xff", 1, 4, 4096, "walk2"),
        ("      u8  D=2  delta 2KB (low-dim)", "delta", 1, 2, 2048, "walk2"),
        ("      u8  D=3  xff 3KB (low-dim)", "xff", 1, 3, 3072, "walk2"),
    ]
    print("| config | ratio | compress GB/s | decompress GB/s | decode ms |")
    print("|---|---|---|---|---|")
    for name, codec, esz, D, chunk_len, kind in cfgs:
        rows = chunk_len // D
        chunk_len = rows * D
        nchunks = max(256, (a.mb << 20) // (chunk_len * esz))
        x = make(kind, nchunks, rows, D, esz, dev)
        cd = sprintz_amd.ChunkedCodec(codec, esz, D, chunk_len, device=dev)
        batch = cd.compress(x)
        out = torch.empty(nchunks * chunk_len, dtype=x.dtype, device=dev)
        cd.decompress_into(batch.data, batch.offsets, nchunks, out)
        torch.cuda.synchronize()
        assert torch.equal(out, x), name
        src = cd._padded_view(x)
        ws = cd.workspace(nchunks)
        dense = torch.empty(nchunks * cd.slot_stride + 16, dtype=torch.uint8, device=dev)
        offs = torch.empty(nchunks + 1, dtype=torch.int64, device=dev)
        tc = timeit(lambda: (cd.compress_to_slots(src, x.numel(), ws), cd.compact(ws, nchunks, dense, offs)), a.reps)
        td = timeit(lambda: cd.decompress_into(batch.data, batch.offsets, nchunks, out), a.reps)
        raw = nchunks * chunk_len * esz
        print(f"| {name} | {raw / batch.stream_bytes():.3f} | {raw / tc / 1e6:.0f} | {raw / td / 1e6:.0f} | {td:.3f} |", flush=True)
        del x, batch, out, src, dense, offs
        cd._ws = {}
        torch.cuda.empty_cache()

    # BASELINE config 5 as stated: column-major matrix, 32 variables, 160-row (10 KB) chunks
    for name, nrows in (("cfg5  u16 D=32 xff COLUMN-MAJOR 1M rows (64 MiB)", 1 << 20), ("cfg5  same, 8M rows (512 MiB)", 1 << 23)):
        D, rpc, esz = 32, 160, 2
        g = torch.Generator(device=dev).manual_seed(5)
        cols = (torch.cumsum(torch.randint(-8, 9, (D, nrows), generator=g, device=dev, dtype=torch.int32), dim=1) & 0xffff).to(torch.uint16)
        cd = sprintz_amd.ChunkedCodec("xff", esz, D, rpc * D, device=dev)
        batch = cd.compress_colmajor(cols)
        nchunks = batch.nchunks
        out = torch.empty((D, nchunks * rpc), dtype=torch.uint16, device=dev)
        assert torch.equal(cd.decompress_colmajor(batch, out=out), cols)
        ws = cd.workspace(nchunks)
        dense = torch.empty(nchunks * cd.slot_stride + 16, dtype=torch.uint8, device=dev)
        offs = torch.empty(nchunks + 1, dtype=torch.int64, device=dev)
        st = cd._stream()

        def enc():
            _lib.check(_lib.compress_batch_colmajor(_lib.CODEC_XFF, esz, cols.data_ptr(), nrows, nrows, rpc, D, ws["slots"].data_ptr(),
                                                    cd.slot_stride, ws["sizes"].data_ptr(), ws["rets"].data_ptr(), st))
            cd.compact(ws, nchunks, dense, offs)

        def dec():
            _lib.check(_lib.decompress_batch_colmajor(_lib.CODEC_XFF, esz, batch.data.data_ptr(), batch.offsets.data_ptr(), nchunks,
                                                      rpc, D, int(out.shape[1]), out.data_ptr(), None, st))
        tc, td = timeit(enc, a.reps), timeit(dec, a.reps)
        raw = nrows * D * esz
        print(f"| {name} | {raw / batch.stream_bytes():.3f} | {raw / tc / 1e6:.0f} | {raw / td / 1e6:.0f} | {td:.3f} |", flush=True)
        del cols, batch, out, dense, offs
        cd._ws = {}
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
