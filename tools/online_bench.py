#!/usr/bin/env python3
"""bench.py's online_coders leg alone (the 2020 coders of online.cpp on one 128 MiB uint16 stream): python tools/online_bench.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch  # noqa: E402

import bench  # noqa: E402

cx = bench.Ctx()
cx.torch, cx.device, cx.timer = torch, torch.device("cuda", 0), bench.Timer(torch)
print(json.dumps(bench.online_leg(cx), indent=1))
