import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minivllm_amd import hip_ops as H
DEV="cuda:0"
n, S, Hq, Hkv, D = 4, 4096, 28, 4, 128
T = n*S
q = torch.randn(T,Hq,D,device=DEV).bfloat16(); k = torch.randn(T,Hkv,D,device=DEV).bfloat16(); v = torch.randn(T,Hkv,D,device=DEV).bfloat16()
cu = (torch.arange(n+1,device=DEV)*S).int()
for _ in range(3):
    H.prefill_varlen(q,k,v,cu,cu,S,S,D**-0.5)
torch.cuda.synchronize()
