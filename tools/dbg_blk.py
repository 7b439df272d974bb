import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import sprintz_amd
from sprintz_amd import _lib
from harness import gen_walk, DTYPES
esz, ndims, chunk_len, nchunks = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rng=np.random.default_rng(5)
data=gen_walk(rng, nchunks*chunk_len, ndims, esz, 3)
cd=sprintz_amd.ChunkedCodec("delta", esz, ndims, chunk_len, device="cuda:0")
_lib.set_option(_lib.OPT_LAT_CHUNKS,0); _lib.set_option(_lib.OPT_BLK_CHUNKS,0)
t=torch.from_numpy(data.view(np.int8 if esz==1 else np.int16)).cuda().view(cd.dtype)
batch=cd.compress(t)
_lib.set_option(_lib.OPT_BLK_CHUNKS,1)
rets=torch.full((nchunks,),-77,dtype=torch.int64,device="cuda:0")
out=torch.full((nchunks*chunk_len,),0x5a,dtype=cd.dtype,device="cuda:0")
cd.decompress_into(batch.data,batch.offsets,nchunks,out,rets)
torch.cuda.synchronize()
o=out.cpu().numpy().view(DTYPES[esz]); r=rets.cpu().numpy()
print("rets",r[:8], "sizes", batch.sizes.cpu().numpy()[:4], "offs", batch.offsets.cpu().numpy()[:4])
bad=np.flatnonzero(o!=data)
print("mismatches",bad.size,"of",data.size, "first", bad[:20])
if bad.size:
    i=bad[0]; print("at",i,"chunk",i//chunk_len,"row",(i%chunk_len)//ndims,"col",i%ndims,"got",o[i:i+20],"want",data[i:i+20])
    # error pattern per row for chunk 0
    c0=o[:chunk_len].reshape(-1,ndims); w0=data[:chunk_len].reshape(-1,ndims)
    rows=np.flatnonzero((c0!=w0).any(axis=1)); print("bad rows chunk0", rows[:40])
    cols=np.flatnonzero((c0!=w0).any(axis=0)); print("bad cols chunk0", cols[:40])
