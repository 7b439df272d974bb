import os, sys, torch
sys.path.insert(0, "/root/repo")
import sprintz_amd
from bench import make_data
dev = torch.device("cuda", 0)
x = make_data(torch, "walk8", 131072, 640, 8, dev, 123)
cd = sprintz_amd.ChunkedCodec("xff", 2, 8, 5120, device=dev)
src = cd._padded_view(x); ws = cd.workspace(131072)
for _ in range(5): cd.compress_to_slots(src, x.numel(), ws)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): cd.compress_to_slots(src, x.numel(), ws)
e1.record(); torch.cuda.synchronize()
print("encode kernel ms", e0.elapsed_time(e1) / 20)
