import torch, sys
sys.path.insert(0,'.')
import sprintz_amd as sz
dev=torch.device('cuda',0)
for esz, D, chunk_len in ((1,1,1024),(2,2,4096)):
    nchunks = (512<<20)//(chunk_len*esz)
    g=torch.Generator(device=dev).manual_seed(1)
    x=torch.randint(-2,3,(nchunks,chunk_len//D,D),device=dev,generator=g,dtype=torch.int32)
    x=(torch.cumsum(x,dim=1,dtype=torch.int32)+100)&((1<<(8*esz))-1)
    x=x.to(torch.uint8).reshape(-1) if esz==1 else torch.where(x>=32768,x-65536,x).to(torch.int16).reshape(-1).view(torch.uint16)
    cd=sz.ChunkedCodec("delta" if esz==1 else "xff",esz,D,chunk_len,device=dev)
    b=cd.compress(x)
    def run(): return cd.query(b,"sum",materialize=False)
    r=run(); torch.cuda.synchronize()
    want = x.view(nchunks*chunk_len//D, D).to(torch.int64).sum(0)
    got = r[0] if isinstance(r, tuple) else r
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(f"esz={esz} D={D}: reduce-only sum {ms:.3f} ms = {x.numel()*esz/ms/1e6:.0f} GB/s scanned; ok={torch.equal(torch.as_tensor(got).to(torch.int64).cpu().reshape(-1), want.cpu())}")
