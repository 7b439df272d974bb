import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import torch, sprintz_amd
from synth import synth_torch
D, rows = 8, (64 << 20) // 8
x = synth_torch("walk", 2, 1, rows, D, "cuda:0", seed=123, step=8).reshape(-1)
y, back = torch.empty_like(x), torch.empty_like(x)
sprintz_amd.transform_device("delta", x, D, out=y)
for _ in range(20): sprintz_amd.transform_device("delta", y, D, inverse=True, out=back)
torch.cuda.synchronize()
