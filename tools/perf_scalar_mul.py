#!/usr/bin/env python3
"""Batched single-scalar multiplication (G1.Mul for many independent pairs), 2^20 points, one MI355X."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from conftest import load_golden
from mathlib_amd import _lib
lib=_lib.load(); dev=torch.device("cuda",0); st=torch.cuda.current_stream().cuda_stream
gen=torch.Generator(device=dev); gen.manual_seed(1)
rnd=lambda k: torch.randint(-(1<<63),(1<<63)-1,(k,4),dtype=torch.int64,generator=gen,device=dev).view(torch.uint8).reshape(k,32).contiguous()
for name in ("BLS12-381","BN254"):
    g=load_golden(name); cid=g["curve_id"]; fpb,g1b,g2b,gtb=_lib.sizes(cid); n=1<<20
    base=torch.frombuffer(bytearray(bytes.fromhex(g["g1_gen"])),dtype=torch.uint8).to(dev)
    P=torch.empty(n*g1b,dtype=torch.uint8,device=dev); S=rnd(n)
    for _ in range(3):
        torch.cuda.synchronize(); t0=time.perf_counter()
        _lib.check(lib.mlhip_scalar_mul_device(cid,1,base.data_ptr(),0,S.data_ptr(),0,n,P.data_ptr(),st)); torch.cuda.synchronize()
        dt=time.perf_counter()-t0
    print(name,"batched G1 scalar mul 2^20: %.1f ms -> %.3e /s"%(dt*1e3,n/dt))
