"""Run the Huffman stage a few times on a cfg2 batch (for rocprofv3 --kernel-trace --stats)."""
import sys, torch
sys.path.insert(0, ".")
from sprintz_amd.codec import ChunkedCodec, huf_compress, huf_decompress
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(123)
steps = torch.randint(-8, 9, (n * 640, 8), generator=g, device=dev, dtype=torch.int32)
x = (torch.cumsum(steps.view(n, 640, 8), dim=1) & 0xffff).to(torch.uint16).reshape(-1)
codec = ChunkedCodec("xff", 2, 8, 5120)
cb = codec.compress(x)
for _ in range(3):
    hb = huf_compress(cb)
    back = huf_decompress(hb, cb.stream_bytes())
torch.cuda.synchronize()
print("sprintz bytes", cb.stream_bytes(), "huf bytes", hb.total_bytes())
