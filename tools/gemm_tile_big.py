import sys, torch
sys.path.insert(0, ".")
from icl_speech_text_llm_amd.runtime import binding as B
B.load_library()
DEV="cuda"
shapes=[("wh qkv",384000,3840,1280,"bf16"),("wh o",384000,1280,1280,"res"),("wh fc1",384000,5120,1280,"gelu"),("wh fc2",384000,1280,5120,"res"),("be o",382976,768,768,"res"),("ll o",48128,4096,4096,"res")]
for name,M,N,K,kind in shapes:
    g=torch.Generator().manual_seed(1)
    a=(torch.randn(M,K,generator=g)*0.5).to(torch.bfloat16).to(DEV)
    w=(torch.randn(N,K,generator=g)*0.05).to(torch.bfloat16).to(DEV)
    bias=torch.randn(N,device=DEV)
    res=torch.randn(M,N,device=DEV) if kind=="res" else None
    out=torch.empty(M,N,dtype=torch.float32 if kind=="res" else torch.bfloat16,device=DEV)
    line=[]
    for tile in (3,1):
        def run(): B.gemm(a,w,out if res is None else res,bias=bias,gelu=kind=="gelu",residual=res,tile=tile)
        for _ in range(2): run()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        us=e0.elapsed_time(e1)/5*1e3
        line.append(f"tile{tile}: {us:8.1f} us {2*M*N*K/us/1e6:7.1f} TF/s")
    print(name, M,N,K, " | ".join(line), flush=True)
