#!/usr/bin/env python3
"""Throughput of the stand-alone transforms on one long stream resident in HBM (GB/s of the
stream: bytes in = bytes out).  Prints a markdown table."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import sprintz_amd  # noqa: E402


def timeit(fn, reps=20):
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
    print("| stream | kind | encode GB/s | decode GB/s | decode ms |")
    print("|---|---|---|---|---|")
    for name, dt, D, rows in (("u16 D=8, 84 M rows (1.34 GB)", torch.uint16, 8, 83_886_080), ("u16 D=32, 1 M rows (64 MiB)", torch.uint16, 32, 1 << 20),
                              ("u16 D=32, 16 M rows (1 GiB)", torch.uint16, 32, 1 << 24), ("u8 D=80, 16 M rows (1.34 GB)", torch.uint8, 80, 1 << 24),
                              ("u16 D=3, 100 M rows", torch.uint16, 3, 100_000_000), ("u8 D=1, 1 G rows", torch.uint8, 1, 1 << 30)):
        n = rows * D
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randint(0, 256 if dt == torch.uint8 else 65536, (n,), generator=g, device=dev, dtype=torch.int32).to(dt)
        nbytes = n * x.element_size()
        for kind in ("delta", "doubledelta"):
            y = torch.empty_like(x)
            back = torch.empty_like(x)
            te = timeit(lambda: sprintz_amd.transform_device(kind, x, D, out=y))
            td = timeit(lambda: sprintz_amd.transform_device(kind, y, D, inverse=True, out=back))
            assert torch.equal(back, x)
            print(f"| {name} | {kind} | {nbytes / te / 1e6:.0f} | {nbytes / td / 1e6:.0f} | {td:.3f} |", flush=True)
        del x, y, back
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
