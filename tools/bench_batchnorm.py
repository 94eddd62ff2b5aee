"""BatchNorm forward / backward on the row counts of the B = 32 train step: the one-launch slab kernels (NSDP_BN_SLAB=1,
default) against stats + finalize + apply (three launches per direction), through the C ABI, with the bytes each direction
has to move (x [+ addend] in, y out; dy, x [, y, addend] in, dx out)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsdp_amd import _lib, hip_batchnorm as H
dev = torch.device("cuda:0")
L = _lib.lib()
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
dtype = torch.bfloat16 if "--bf16" in sys.argv else torch.float32
print(f"storage {dtype}; us per direction (back-to-back launches of the same op: includes the kernel boundaries inside the op)")
for (R, C, addend, relu) in [(3200, 256, False, False), (3200, 256, True, False), (3200, 256, False, True), (16000, 120, False, False),
                             (16000, 120, True, False), (16000, 256, False, True), (16000, 256, True, False), (8000, 256, False, False), (65536, 120, False, False)]:
    x = torch.randn(R, C, device=dev).to(dtype); a = torch.randn(R, C, device=dev).to(dtype) if addend else None
    go = torch.randn(R, C, device=dev).to(dtype)
    row = []
    for slab in (1, 0):
        L.nsdp_debug_set(11, slab)
        bn = torch.nn.BatchNorm1d(C).to(dev)
        xx = x.clone().requires_grad_(True)
        y = H.batch_norm(xx, bn, addend=a, relu=relu)
        tf = timeit(lambda: H.batch_norm(x, bn, addend=a, relu=relu))
        tb = timeit(lambda: torch.autograd.grad(y, [xx, bn.weight, bn.bias], go, retain_graph=True))
        row.append((tf, tb))
    L.nsdp_debug_set(11, 1)
    es = x.element_size()
    bf = R * C * es * (2 + addend) / 1e3; bb = R * C * es * (3 + addend + relu) / 1e3     # KB... / us = GB/s
    (f1, b1), (f0, b0) = row
    print(f"R={R:6d} C={C:3d} addend={int(addend)} relu={int(relu)}: fwd slab {f1:6.1f} us ({bf/f1/1e3:5.2f} TB/s) 3-launch {f0:6.1f} us | "
          f"bwd slab {b1:6.1f} us ({bb/b1/1e3:5.2f} TB/s) 3-launch {b0:6.1f} us")
