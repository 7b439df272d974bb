#!/usr/bin/env python3
"""cfg4's chain with the two stages overlapped: the batch cut into slices, the Huff0 stage of slice i+1 on one HIP stream
while the Sprintz decoder takes slice i on another.   python tools/overlap_probe.py [--chunks 800000] [--slices 8]"""
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
from synth import synth_torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=800000)
    ap.add_argument("--slices", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    n, chunk_len, D = a.chunks, 5120, 8
    x = synth_torch("walk", 2, n, chunk_len // D, D, dev, seed=123, step=8).view(torch.int16)
    cd = sprintz_amd.ChunkedCodec("xff", 2, D, chunk_len, device=dev)
    batch = cd.compress(x.view(torch.uint16))
    blocks, bo = sprintz_amd.huf0_compress(batch)
    sizes = batch.sizes.to(torch.int64)
    oo = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    oo[1:] = torch.cumsum(sizes, 0)
    streams = torch.zeros(int(oo[-1].item()) + 64, dtype=torch.uint8, device=dev)
    out = torch.empty(n * chunk_len, dtype=torch.int16, device=dev)
    rets = torch.empty(n, dtype=torch.int64, device=dev)
    tmp = torch.empty(int(_lib.huf0_decode_tmp_bytes(n)), dtype=torch.uint8, device=dev)

    def stage1(c0, c1, st, tmp_t):
        _lib.check(_lib.huf0_decompress_batch_ws(blocks.data_ptr(), bo[c0:].data_ptr(), c1 - c0, streams.data_ptr(), oo[c0:].data_ptr(),
                                                 rets[c0:].data_ptr(), tmp_t.data_ptr(), C.c_void_p(st.cuda_stream)))

    def stage2(c0, c1, st):
        with torch.cuda.stream(st):
            cd.decompress_into(streams, oo[c0:], c1 - c0, out[c0 * chunk_len:])

    def timeit(fn, reps=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    cur = torch.cuda.current_stream(dev)

    def serial():
        stage1(0, n, cur, tmp)
        stage2(0, n, cur)
    t_serial = timeit(serial)
    assert torch.equal(out, x)

    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    S = a.slices
    per = (n + S - 1) // S // 64 * 64
    cuts = [min(n, i * per) for i in range(S)] + [n]
    tmps = [torch.empty(int(_lib.huf0_decode_tmp_bytes(per + 64)), dtype=torch.uint8, device=dev) for _ in range(2)]

    def overlapped():
        start = torch.cuda.Event(); start.record(cur)
        sa.wait_event(start); sb.wait_event(start)
        evs = []
        for i in range(S):
            if i >= 2:
                sa.wait_event(done2[i - 2])                # its workspace is free again
            stage1(cuts[i], cuts[i + 1], sa, tmps[i & 1])
            e = torch.cuda.Event(); e.record(sa); evs.append(e)
            sb.wait_event(e)
            stage2(cuts[i], cuts[i + 1], sb)
            d = torch.cuda.Event(); d.record(sb); done2.append(d)
        cur.wait_event(done2[-1])
        cur.wait_event(evs[-1])
    done2 = []

    def run_overlapped():
        done2.clear()
        overlapped()
    out.zero_()
    t_over = timeit(run_overlapped)
    assert torch.equal(out, x)
    print(f"chunks {n}: serial {t_serial:.3f} ms, {S} slices on two streams {t_over:.3f} ms")


if __name__ == "__main__":
    main()
