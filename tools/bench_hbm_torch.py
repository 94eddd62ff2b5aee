"""What the chip gives plain torch kernels on a decoder-sized fp32 tensor (1.47 GB): sum / dot / copy / add / fill / relu_ in TB/s --
the yardstick for the attention streams (docs/EXPERIMENTS.md, round 6).    python tools/bench_hbm_torch.py"""
import torch
dev=torch.device("cuda:0")
n=1835008*200
a=torch.randn(n,device=dev); b=torch.randn(n,device=dev); c=torch.empty_like(a); d=torch.empty_like(a)
def t(fn,reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/reps*1e-3
G=n*4/1e12
x=t(lambda: a.sum()); print(f"sum (1R): {x*1e6:.0f} us {G/x:.2f} TB/s")
x=t(lambda: torch.dot(a,b)); print(f"dot (2R): {x*1e6:.0f} us {2*G/x:.2f} TB/s")
x=t(lambda: c.copy_(a)); print(f"copy (1R1W): {x*1e6:.0f} us {2*G/x:.2f} TB/s")
x=t(lambda: torch.add(a,b,out=c)); print(f"add (2R1W): {x*1e6:.0f} us {3*G/x:.2f} TB/s")
x=t(lambda: c.zero_()); print(f"fill (1W): {x*1e6:.0f} us {G/x:.2f} TB/s")
x=t(lambda: torch.relu_(c)); print(f"relu_ (1R1W same): {x*1e6:.0f} us {2*G/x:.2f} TB/s")
