# probe: achievable HBM write-only / copy bandwidth on this GPU (torch kernels), for roofline context
import torch, time
n = 1342177280
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty(n, dtype=torch.uint8, device="cuda")
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
ms = t(lambda: a.zero_()); print(f"fill  1.342 GB: {ms:.4f} ms  -> {n/ms/1e9:.2f} TB/s written")
ms = t(lambda: b.copy_(a)); print(f"copy  1.342 GB: {ms:.4f} ms  -> {2*n/ms/1e9:.2f} TB/s read+written")
av = a.view(torch.int32); ms = t(lambda: av.sum()); print(f"read  1.342 GB: {ms:.4f} ms  -> {n/ms/1e9:.2f} TB/s read")
