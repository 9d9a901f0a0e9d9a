import torch, time
dev=torch.device('cuda:0')
M,K,N=32768,3136,512
a=torch.randn(M,K,device=dev); W=torch.randn(N,K,device=dev); Wt=W.t().contiguous(); b=torch.randn(N,device=dev)
def bench(f,reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1)*1e3/reps
fl=2*M*K*N
for lib in ("default","hipblaslt","hipblas"):
    try:
        if lib!="default": torch.backends.cuda.preferred_blas_library(lib)
    except Exception as ex: print(lib,ex); continue
    for name,f in [("linear(a,W,b)",lambda: torch.nn.functional.linear(a,W,b)),("linear(a,W)",lambda: torch.nn.functional.linear(a,W)),("a@Wt",lambda: a@Wt),("addmm(b,a,Wt)",lambda: torch.addmm(b,a,Wt)),
                   ("(W@a.T)",lambda: W@a.t()),("dgrad g@W", None),("wgrad g.T@a",None)]:
        if f is None: continue
        us=bench(f); print(lib,name,round(us,1),"us",round(fl/us/1e6,1),"TF")
    g=torch.randn(M,N,device=dev)
    us=bench(lambda: g@W); print(lib,"dgrad g@W",round(us,1),round(fl/us/1e6,1))
    us=bench(lambda: g.t()@a); print(lib,"wgrad g.T@a",round(us,1),round(fl/us/1e6,1))
    # small rollout shapes
    a2=torch.randn(1024,K,device=dev)
    us=bench(lambda: torch.nn.functional.linear(a2,W,b),50); print(lib,"rollout linear 1024",round(us,1),round(2*1024*K*N/us/1e6,1))
print("--- wgrad variants")
torch.backends.cuda.preferred_blas_library("default")
g=torch.randn(M,N,device=dev)
for lib in ("default","hipblaslt"):
    if lib!="default": torch.backends.cuda.preferred_blas_library(lib)
    for name,f in [("g.t()@a",lambda: g.t()@a),("(a.t()@g)",lambda: a.t()@g),("mm(gT_contig,a)",None)]:
        if f is None:
            gT=g.t().contiguous(); f=lambda: gT@a
        us=bench(f); print(lib,name,round(us,1),round(fl/us/1e6,1))
    aT=a.t().contiguous()
    us=bench(lambda: aT@g); print(lib,"aT_contig@g",round(us,1),round(fl/us/1e6,1))
    out=torch.empty(N,K,device=dev)
    us=bench(lambda: torch.mm(g.t(),a,out=out)); print(lib,"mm out=",round(us,1),round(fl/us/1e6,1))
    try:
        us=bench(lambda: torch._addmm_activation(b,a,Wt)); print(lib,"_addmm_activation relu",round(us,1),round(fl/us/1e6,1))
    except Exception as ex: print("addmm_activation failed",ex)
print("--- split-K wgrad via bmm")
torch.backends.cuda.preferred_blas_library("default")
for S in (2,3,4,8,16):
    if M % S: continue
    f=lambda: torch.bmm(g.view(S,M//S,N).transpose(1,2), a.view(S,M//S,K)).sum(0)
    us=bench(f); print("S",S,round(us,1),round(fl/us/1e6,1))
ref=(g.t()@a); got=torch.bmm(g.view(4,M//4,N).transpose(1,2), a.view(4,M//4,K)).sum(0)
print("maxdiff",(ref-got).abs().max().item(), ref.abs().max().item())
