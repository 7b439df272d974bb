#!/usr/bin/env python3
"""cfg4 (Huff0 -> Sprintz) at bandwidth-sized batches: does it pay to cut the batch into P parts and run the Huff0 stage of part
i + 1 on one HIP stream while the Sprintz decoder takes part i on another?  The Huff0 stage is bound by dependent LDS look-ups (VALU
and LDS a third busy), the Sprintz decoder by how its stores meet the memory system: different resources, and a part's streams
(chunks * 3.6 KB) may stay in the 256 MB memory-side cache between the two.  Caller-side experiment on the unchanged C-ABI: the
parts are the same calls with the offset arrays and the output advanced by part * S chunks (S a multiple of 64, the writer's segment).

    python tools/cfg4_pipeline.py [--chunks 800000] [--parts 1,2,4,8,16,32] [--reps 5]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import sprintz_amd  # noqa: E402
from sprintz_amd import _lib  # noqa: E402
from bench_configs import make  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=800000)
    ap.add_argument("--parts", default="1,2,4,8,16,32")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    chunk_len, D, n = 5120, 8, a.chunks
    base = 8192                                              # distinct chunks, tiled (as the bench's large cfg4 batches are)
    x0 = make("walk8", base, chunk_len // D, D, 2, dev).view(torch.uint16)
    x = x0.repeat((n + base - 1) // base)[: n * chunk_len].contiguous()
    cd = sprintz_amd.ChunkedCodec("xff", 2, D, chunk_len, device=dev)
    big = cd.compress(x)
    blocks, boffs = sprintz_amd.huf0_compress(big)
    sizes = big.sizes.to(torch.int64)
    soffs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    soffs[1:] = torch.cumsum(sizes, 0)
    stream_bytes = int(soffs[-1].item())
    hbytes = int(boffs[-1].item())
    sbuf = torch.zeros(stream_bytes + _lib.READ_SLACK, dtype=torch.uint8, device=dev)
    out = torch.empty(n * chunk_len, dtype=torch.uint16, device=dev)
    rets = torch.empty(n, dtype=torch.int64, device=dev)
    tmp = torch.empty(int(_lib.huf0_decode_tmp_bytes(n)) + 4096, dtype=torch.uint8, device=dev)
    hint = int((boffs[1:] - boffs[:-1]).max().item())
    codec = _lib.CODEC_XFF
    raw = n * chunk_len * 2
    algo = hbytes + 16 * n + raw
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    p1, p2 = C.c_void_p(s1.cuda_stream), C.c_void_p(s2.cuda_stream)

    def chain(parts):
        S = ((n + parts - 1) // parts + 63) // 64 * 64
        evs = []
        for i in range(parts):
            c0 = i * S
            if c0 >= n:
                break
            m = min(S, n - c0)
            t = tmp.data_ptr()                               # one descriptor workspace: the Huff0 launches of all parts are ordered on their stream
            _lib.check(_lib.huf0_decompress_batch_hint(blocks.data_ptr(), boffs.data_ptr() + 8 * c0, m, sbuf.data_ptr(), soffs.data_ptr() + 8 * c0,
                                                       rets.data_ptr() + 8 * c0, t, hint, p1))
            ev = torch.cuda.Event()
            ev.record(s1)
            s2.wait_event(ev)
            _lib.check(_lib.decompress_batch(codec, 2, sbuf.data_ptr(), soffs.data_ptr() + 8 * c0, m, chunk_len, D,
                                             out.data_ptr() + 2 * c0 * chunk_len, None, p2))
            evs.append(ev)

    def timed(parts):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s1.wait_stream(torch.cuda.current_stream(dev))
        e0.record(s1)
        s2.wait_event(e0)
        for _ in range(a.reps):
            chain(parts)
            # the next repetition's Huff0 stage must not overwrite streams the Sprintz decoder is still reading
            ev = torch.cuda.Event()
            ev.record(s2)
            s1.wait_event(ev)
        s1.wait_stream(s2)
        e1.record(s1)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    # correctness of the partitioned calls first
    out.zero_()
    chain(8)
    torch.cuda.synchronize()
    assert torch.equal(out, x), "partitioned chain != input"
    assert torch.equal(rets, sizes)
    print(f"chunks {n}  raw {raw/1e6:.0f} MB  sprintz streams {stream_bytes/1e6:.0f} MB  Huff0 blocks {hbytes/1e6:.0f} MB  algorithmic {algo/1e6:.0f} MB")
    for rnd in range(2):
        for parts in [int(p) for p in a.parts.split(",")]:
            ms = timed(parts)
            print(f"round {rnd} parts {parts:3d} ({(n + parts - 1) // parts:7d} chunks, {stream_bytes / parts / 1e6:7.1f} MB of streams a part): "
                  f"chain {ms:.3f} ms = {algo / ms / 1e6:.0f} GB/s = {algo / ms / 1e6 / 8000:.3f}", flush=True)


if __name__ == "__main__":
    main()
