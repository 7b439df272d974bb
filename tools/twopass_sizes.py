import os, sys, ctypes as C
sys.path.insert(0, "/root/repo")
import torch, sprintz_amd
from sprintz_amd import _lib
dev=torch.device("cuda:0")
w=torch.empty(1<<28,dtype=torch.uint8,device=dev)
for _ in range(200): w.add_(1)
torch.cuda.synchronize()
st=C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
def timed(fn,reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); best=1e9
    for _ in range(3):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize(); best=min(best,a.elapsed_time(b)/reps)
    return best*1e3
g=torch.Generator(device="cuda"); g.manual_seed(3)
for mb in (0.03125,0.125,0.5,2,8,32,128):
    n=int(mb*(1<<20))
    line="%7.3f MB" % mb
    for esz,D in ((1,8),(1,80),(2,6)):
        m=n//esz//D*D
        x=torch.randint(0,1<<(8*esz),(m,),generator=g,device="cuda",dtype=torch.int32).to(torch.uint8 if esz==1 else torch.uint16)
        for kind,k in (("delta",0),("doubledelta",1)):
            y=sprintz_amd.transform_device(kind,x,D); back=torch.empty_like(x)
            tmp=torch.empty(int(_lib.transform_tmp_bytes(k,esz,m,D)),dtype=torch.uint8,device=dev)
            t=timed(lambda:_lib.check(_lib.transform_decode_device(k,esz,y.data_ptr(),m,D,back.data_ptr(),tmp.data_ptr(),st)))
            assert torch.equal(back,x)
            line+="  u%dx%d %s %6.1f" % (8*esz,D,kind[:2],t)
    print(line,flush=True)
