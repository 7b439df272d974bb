#!/usr/bin/env python3
"""single legs of bench.py's extras: python tools/online_bench.py [online|transforms|any_ndims ...]  (default: online)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch  # noqa: E402

import bench  # noqa: E402

cx = bench.Ctx()
cx.torch, cx.device, cx.timer = torch, torch.device("cuda", 0), bench.Timer(torch)
for leg in (sys.argv[1:] or ["online"]):
    print(json.dumps({leg: getattr(bench, leg + "_leg")(cx)}, indent=1))
