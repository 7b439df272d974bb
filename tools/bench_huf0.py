#!/usr/bin/env python3
"""Throughput of the Huff0 wire-format decoder (huf0.hip) on the headline shape: the Sprintz
streams of cfg2 (uint16, 8 columns, FIRE, 10 KB chunks) entropy-coded chunk by chunk with the
system libzstd's HUF_compress (the input generator here; absent -> exit), decoded on the GPU and
chained into the Sprintz decoder.    python tools/bench_huf0.py [--chunks 131072]"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sprintz_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=131072)
    ap.add_argument("--distinct", type=int, default=4096, help="chunks actually entropy-coded on the CPU (tiled to --chunks)")
    a = ap.parse_args()
    try:
        z = C.CDLL("libzstd.so.1")
        z.HUF_compress.restype = C.c_size_t
        z.HUF_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        z.HUF_isError.restype = C.c_uint
        z.HUF_isError.argtypes = [C.c_size_t]
    except (OSError, AttributeError):
        print("no libzstd with HUF_compress here")
        return
    dev = torch.device("cuda", 0)
    chunk_len, D = 5120, 8
    n0 = a.distinct
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randint(-8, 9, (n0, chunk_len // D, D), device=dev, generator=g, dtype=torch.int32)
    x = (torch.cumsum(x, dim=1, dtype=torch.int32) + torch.randint(0, 65536, (n0, 1, D), device=dev, generator=g, dtype=torch.int32)) & 0xffff
    x = torch.where(x >= 32768, x - 65536, x).to(torch.int16).reshape(-1).view(torch.uint16)
    cd = sprintz_amd.ChunkedCodec("xff", 2, D, chunk_len, device=dev)
    batch = cd.compress(x)
    comp, offs, sizes = batch.data.cpu().numpy(), batch.offsets.cpu().numpy(), batch.sizes.cpu().numpy()
    blocks, out = [], np.zeros(1 << 16, np.uint8)
    t0 = time.perf_counter()
    for c in range(n0):
        s = np.ascontiguousarray(comp[offs[c]:offs[c] + sizes[c]])
        r = z.HUF_compress(out.ctypes.data, out.size, s.ctypes.data, s.size)
        blocks.append(s.copy() if (r == 0 or z.HUF_isError(r)) else out[:r].copy())
    cpu_s = time.perf_counter() - t0
    reps = (a.chunks + n0 - 1) // n0
    blocks = (blocks * reps)[: a.chunks]
    plain_sizes = np.tile(sizes, reps)[: a.chunks].astype(np.int64)
    bo = np.zeros(a.chunks + 1, np.int64)
    bo[1:] = np.cumsum([b.size for b in blocks])
    # the streams are laid out byte-dense (the decoder takes a chunk's size as an offset difference); the
    # Sprintz decoder takes unaligned stream starts as well
    oo_dense = np.zeros(a.chunks + 1, np.int64)
    oo_dense[1:] = np.cumsum(plain_sizes)
    oo_d = torch.from_numpy(oo_dense).to(dev)
    d = torch.from_numpy(np.concatenate(blocks + [np.zeros(16, np.uint8)])).to(dev)
    bo_d = torch.from_numpy(bo).to(dev)
    rets = torch.empty(a.chunks, dtype=torch.int64, device=dev)
    streams = sprintz_amd.huf0_decompress(d, bo_d, oo_d, rets=rets)
    assert bool((rets.cpu().numpy() == plain_sizes).all())
    raw = torch.empty(a.chunks * chunk_len, dtype=torch.uint16, device=dev)
    cd.decompress_into(streams, oo_d, a.chunks, raw)
    assert torch.equal(raw[: n0 * chunk_len], x)

    def timeit(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    raw_bytes = a.chunks * chunk_len * 2
    t_h = timeit(lambda: sprintz_amd.huf0_decompress(d, bo_d, oo_d, out=streams))
    t_s = timeit(lambda: cd.decompress_into(streams, oo_d, a.chunks, raw))
    raw_bytes = a.chunks * chunk_len * 2
    # the GPU writer on the same container (tiled like the blocks above)
    big = cd.compress(x.repeat(reps)[: a.chunks * chunk_len])
    gb, gbo = sprintz_amd.huf0_compress(big)
    t_w = timeit(lambda: sprintz_amd.huf0_compress(big), reps=5)
    gtotal = int(gbo[-1].item())
    goo = torch.zeros(a.chunks + 1, dtype=torch.int64, device=dev)
    goo[1:] = torch.cumsum(big.sizes.to(torch.int64), 0)
    gst = sprintz_amd.huf0_decompress(gb, gbo, goo, rets=rets)
    assert bool((rets == big.sizes.to(torch.int64)).all())
    t_gd = timeit(lambda: sprintz_amd.huf0_decompress(gb, gbo, goo, out=gst))
    print(f"GPU writer: {t_w:.3f} ms = {int(goo[-1].item())/t_w/1e6:.0f} GB/s of stream bytes in, ratio {raw_bytes/gtotal:.3f}; "
          f"its blocks decode in {t_gd:.3f} ms")
    print(f"chunks {a.chunks}  raw {raw_bytes/1e6:.0f} MB  sprintz {oo_dense[-1]/1e6:.0f} MB  huff0 {bo[-1]/1e6:.0f} MB  "
          f"ratio {raw_bytes/bo[-1]:.3f} (sprintz alone {raw_bytes/oo_dense[-1]:.3f})")
    print(f"libzstd HUF_compress on 1 host thread: {sizes.sum()/cpu_s/1e6:.0f} MB/s of stream bytes")
    print(f"huff0 decode  {t_h:.3f} ms = {oo_dense[-1]/t_h/1e6:.0f} GB/s of stream bytes out")
    print(f"sprintz decode (byte-dense streams) {t_s:.3f} ms;  chain {t_h+t_s:.3f} ms = {raw_bytes/(t_h+t_s)/1e6:.0f} GB/s decompressed")


if __name__ == "__main__":
    main()
