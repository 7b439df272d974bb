#!/usr/bin/env python3
"""cfg2 / cfg4 at different batch sizes: where the path turns from launch-bound to bandwidth-bound
(1 250 chunks = one GPU's share of a 10 000-chunk batch over 8 GPUs, ... 800 000).
Prints a markdown table.  Times are per call, HIP events, data resident in HBM."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import sprintz_amd  # noqa: E402
from sprintz_amd import _lib  # noqa: E402
from sprintz_amd.codec import huf_compress  # noqa: E402


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
    dev = torch.device("cuda", 0)
    D, chunk_len, esz = 8, 5120, 2
    print("| chunks | raw MB | Sprintz decode µs | GB/s | Huffman decode µs | chain GB/s | query(sum) µs |")
    print("|---|---|---|---|---|---|---|")
    for n in (1250, 10000, 80000, 131072, 800000):
        g = torch.Generator(device=dev).manual_seed(n)
        x = (torch.cumsum(torch.randint(-8, 9, (n, 640, D), generator=g, device=dev, dtype=torch.int32), dim=1) & 0xffff)
        x = x.to(torch.uint16).reshape(-1)
        cd = sprintz_amd.ChunkedCodec("xff", esz, D, chunk_len, device=dev)
        batch = cd.compress(x)
        out = torch.empty(n * chunk_len, dtype=torch.uint16, device=dev)
        hb = huf_compress(batch)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        cap = batch.total_bytes() + 16 * n
        d_buf = torch.zeros(cap + 16, dtype=torch.uint8, device=dev)
        d_offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
        d_sizes = torch.empty(n, dtype=torch.int32, device=dev)
        tmp = torch.empty(int(_lib.huf_tmp_bytes(n)), dtype=torch.uint8, device=dev)
        part = torch.empty((n, D), dtype=torch.int64, device=dev)
        res = torch.empty(D, dtype=torch.int64, device=dev)

        def dec():
            cd.decompress_into(batch.data, batch.offsets, n, out)

        def hdec():
            _lib.check(_lib.huf_decompress_batch(hb.data.data_ptr(), hb.offsets.data_ptr(), hb.tables.data_ptr(), n, 16, d_buf.data_ptr(),
                                                 cap, d_offs.data_ptr(), d_sizes.data_ptr(), None, tmp.data_ptr(), st))

        def query():
            _lib.check(_lib.query_batch(_lib.CODEC_XFF, esz, batch.data.data_ptr(), batch.offsets.data_ptr(), n, chunk_len, D,
                                        _lib.QUERY_SUM, 0, 0, None, part.data_ptr(), None, st))
            _lib.check(_lib.query_reduce(_lib.QUERY_SUM, part.data_ptr(), n, D, res.data_ptr(), st))

        reps = 200 if n <= 10000 else 30
        td, th, tq = timeit(dec, reps), timeit(hdec, reps), timeit(query, reps)
        raw = n * chunk_len * esz
        print(f"| {n} | {raw / 1e6:.1f} | {td * 1e3:.1f} | {raw / td / 1e6:.0f} | {th * 1e3:.1f} | {raw / (td + th) / 1e6:.0f} | {tq * 1e3:.1f} |", flush=True)
        del x, batch, out, hb, d_buf, part
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
