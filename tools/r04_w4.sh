#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
python3 - <<'PY'
import torch, sys
sys.path.insert(0,'.')
from ruart_amd import hip
lib=hip.load(); d=torch.device('cuda:0')
g=torch.Generator().manual_seed(1)
for (M,N,K,act,res,of) in [(512,768,768,hip.ACT_NONE,True,True),(768,512,128,hip.ACT_NONE,False,False),(1024,2304,768,hip.ACT_NONE,False,False),(512,3072,768,hip.ACT_GELU,False,False),(512,768,3072,hip.ACT_NONE,True,True)]:
  for dt in (hip.DT_F16, hip.DT_BF16):
    td=hip.TORCH_DTYPE[dt]
    A=torch.randn(M,K,generator=g).to(td).to(d); W=(torch.randn(N,K,generator=g)*0.05).to(td).to(d); b=torch.randn(N,generator=g).to(d)
    R=torch.randn(M,N,generator=g).to(d) if res else None
    outs=[]
    for v in (5,7):
        assert lib.ruart_gemm_set_variant(v)==0
        C=torch.full((M,N),float('nan'),dtype=torch.float32 if of else td,device=d)
        rc=lib.ruart_gemm_16_nt(hip.ptr(A),K,hip.ptr(W),K,hip.ptr(b),hip.ptr(R),N,hip.DT_F32,hip.ptr(C),N,hip.DT_F32 if of else dt,M,N,K,act,dt,hip.stream_ptr())
        assert rc==0; torch.cuda.synchronize(); outs.append(C.float())
    err=(outs[0]-outs[1]).abs().max().item(); print(M,N,K,act,res,dt,'max diff v5 vs v7', err, 'nan', torch.isnan(outs[1]).any().item())
    assert err < 1e-2 and not torch.isnan(outs[1]).any()
lib.ruart_gemm_set_variant(5)
print("w4 correct")
PY
python3 tools/gemm_bench.py --variants 5,7 --orders 8 --iters 20 2>&1 | grep -v amdgpu
python3 tools/gemm_bench.py --variants 5,7 --orders 8 --iters 20 --rows 32768 2>&1 | grep -v amdgpu
