"""Time the Huffman stage kernels with HIP events (cfg2 batch)."""
import sys, torch, ctypes as C
sys.path.insert(0, ".")
from sprintz_amd import _lib
from sprintz_amd.codec import ChunkedCodec, huf_compress, huf_decompress
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(123)
steps = torch.randint(-8, 9, (n * 640, 8), generator=g, device=dev, dtype=torch.int32)
x = (torch.cumsum(steps.view(n, 640, 8), dim=1) & 0xffff).to(torch.uint16).reshape(-1)
codec = ChunkedCodec("xff", 2, 8, 5120)
cb = codec.compress(x)
hb = huf_compress(cb)
stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
dense = torch.zeros(cb.stream_bytes() + 16 + 16 * n, dtype=torch.uint8, device=dev)
offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
sizes = torch.empty(n, dtype=torch.int32, device=dev)
tmp = torch.empty(int(_lib.compact_tmp_bytes(n)) + 64, dtype=torch.uint8, device=dev)
def dec():
    _lib.check(_lib.huf_decompress_batch(hb.data.data_ptr(), hb.offsets.data_ptr(), hb.tables.data_ptr(), n, 16,
                                         dense.data_ptr(), dense.numel() - 16, offs.data_ptr(), sizes.data_ptr(), None,
                                         tmp.data_ptr(), stream))
for _ in range(5): dec()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): dec()
e1.record(); torch.cuda.synchronize()
print("huf decode ms", e0.elapsed_time(e1) / 20, "sprintz bytes", cb.stream_bytes(), "huf bytes", hb.total_bytes())
