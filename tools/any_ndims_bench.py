"""csrc/any_ndims.hip at a few widths: decode / encode ms and the fraction of 8 TB/s ((samples + stream bytes) / time), round trip checked.
usage: python tools/any_ndims_bench.py [D ...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
import sprintz_amd  # noqa: E402
from synth import synth_torch  # noqa: E402


def timed(fn, reps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    dev = torch.device("cuda:0")
    for D in [int(v) for v in sys.argv[1:]] or [600, 1000, 1024, 1500, 2047]:
        for codec, esz in (("xff", 2), ("delta", 1)):
            rows, n = 256, 1024
            x = synth_torch("walk", esz, n, rows, D, dev, seed=123, step=8)
            x = x.view(torch.int16) if esz == 2 else x
            cd = sprintz_amd.ChunkedCodec(codec, esz, D, rows * D, device=dev)
            batch = cd.compress(x)
            out = torch.empty_like(x)
            te = timed(lambda: cd.compress(x), 5, 1)
            td = timed(lambda: cd.decompress(batch, out=out), 5, 1)
            assert torch.equal(out, x)
            raw, sb = x.numel() * esz, batch.stream_bytes()
            print(f"D {D} {codec} u{8 * esz}: dec {td:.3f} ms = {(raw + sb) / td / 8e9:.3f}   enc {te:.3f} ms = {(raw + sb) / te / 8e9:.3f}   ratio {raw / sb:.3f}", flush=True)


main()
