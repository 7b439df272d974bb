#!/usr/bin/env python3
"""Interleaved A/B of library builds (and/or env knobs) on ONE box -- box-to-box spread of the same
binary is +-4 %, so variants are only ever compared inside one gpurun call.

    python tools/ab.py --cfg headline|cfg1|cfg3_10k|... [--rounds 3] NAME=path/to/lib.so[,ENV=VAL...] ...

Prints decompress / compress ms of every variant per round and the medians."""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="headline")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--no-verify", action="store_true", help="ablated builds that decode garbage on purpose")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    res = {}
    for r in range(a.rounds):
        for v in a.variants:
            name, spec = v.split("=", 1)
            parts = spec.split(",")
            env = dict(os.environ)
            if parts[0]:
                env["SPRINTZ_MI355X_LIB"] = os.path.join(ROOT, parts[0])
            for kv in parts[1:]:
                k, val = kv.split("=", 1)
                env[k] = val
            if a.cfg == "headline":
                cmd = [sys.executable, "bench.py", "--configs", "none", "--no-extras", "--no-cpu-baseline", "--steps", str(a.reps)]
            else:
                cmd = [sys.executable, "bench.py", "--only", a.cfg, "--no-cpu-baseline", "--config-reps", str(a.reps)]
            if a.no_verify:
                cmd.append("--no-verify")
            p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
            try:
                d = json.loads(p.stdout.strip().splitlines()[-1])
            except Exception:
                print(name, "FAILED", p.stderr[-800:])
                continue
            if a.cfg == "headline":
                dec, enc = d["kernel_ms"], d["compress"]["ms_per_step_max_rank"]
                extra = ""
            else:
                dec, enc = d["decompress_ms"], d["compress_ms"]
                extra = " ".join(f"{k}={d[k]}" for k in ("huff0_decode_ms", "sprintz_decode_ms", "huff0_encode_ms") if k in d)
            res.setdefault(name, []).append((dec, enc))
            print(f"round {r} {name:12s} decode {dec:.4f} ms  encode {enc:.4f} ms  {extra}", flush=True)
    for name, v in res.items():
        print(f"MEDIAN {name:12s} decode {statistics.median(x[0] for x in v):.4f}  encode {statistics.median(x[1] for x in v):.4f}")


if __name__ == "__main__":
    main()
