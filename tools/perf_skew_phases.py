import os, sys, time, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from conftest import load_golden
from mathlib_amd import _lib
lib=_lib.load(); dev=torch.device("cuda",0); st=torch.cuda.current_stream().cuda_stream
gen=torch.Generator(device=dev); gen.manual_seed(5)
g=load_golden("BLS12-381"); cid=g["curve_id"]; fpb,g1b,g2b,gtb=_lib.sizes(cid); n=1<<20
rnd=lambda k: torch.randint(-(1<<63),(1<<63)-1,(k,4),dtype=torch.int64,generator=gen,device=dev).view(torch.uint8).reshape(k,32).contiguous()
base=torch.frombuffer(bytearray(bytes.fromhex(g["g1_gen"])),dtype=torch.uint8).to(dev)
P=torch.empty(n*g1b,dtype=torch.uint8,device=dev)
_lib.check(lib.mlhip_scalar_mul_device(cid,1,base.data_ptr(),0,rnd(n).data_ptr(),0,n,P.data_ptr(),st))
S=rnd(n)
sk=S.clone().view(torch.int64).reshape(n,4); sk[:,1:]=0; sk[:,0]&=0xFFFFFFFF; sk[::100]=sk[0]; sk=sk.view(torch.uint8).reshape(n,32).contiguous()
Pk=P.clone().reshape(n,g1b); Pk[::100]=Pk[0]; Pk=Pk.reshape(-1).contiguous()
eq=S.clone().view(torch.int64).reshape(n,4); eq[:]=eq[0]; eq=eq.view(torch.uint8).reshape(n,32).contiguous()
plan=_lib.MsmPlan(cid,1,n,16); plan.set_profiling(True)
for name,pp,ss in (("uniform",P,S),("skewed",Pk,sk),("all equal",P,eq)):
    for _ in range(3): plan.run(pp.data_ptr(),ss.data_ptr(),n,False,st)
    ts=[]
    for _ in range(5):
        t0=time.perf_counter(); plan.run(pp.data_ptr(),ss.data_ptr(),n,False,st); ts.append((time.perf_counter()-t0)*1e3)
    print(name,"%.3f ms"%sorted(ts)[2],{k:round(v,3) for k,v in plan.timings().items()},flush=True)
