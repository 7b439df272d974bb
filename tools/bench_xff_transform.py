import torch, time, sys
sys.path.insert(0, '.')
import sprintz_amd as sz
for esz, D, rows in ((2, 8, 1 << 20), (2, 80, 1 << 18), (1, 1, 1 << 20), (2, 1024, 1 << 15), (2, 16384, 1 << 12)):
    n = D * rows
    x = torch.randint(0, 1 << (8 * esz), (n,), device="cuda", dtype=torch.int32).to(torch.uint8 if esz == 1 else torch.uint16)
    for inv in (False, True):
        y = sz.transform_device("xff", x, D, inverse=inv)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = sz.transform_device("xff", x, D, inverse=inv)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"xff esz={esz} D={D} rows={rows} {'decode' if inv else 'encode'}: {dt*1e3:.2f} ms  {n*esz/dt/1e9:.2f} GB/s")
