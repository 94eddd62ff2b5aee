#!/bin/bash
# phase timing of the bf16x3 weight-gradient kernel (instrumented build, GPU box only): tools/time_wgrad.sh M N K
cd "$(dirname "$0")/.."
OBJ=nsdp_amd/lib/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=fast -DWG3_TIMING $WG3_EXTRA \
    -c nsdp_amd/csrc/wgrad_bf16x3.hip -o $OBJ/wgrad_bf16x3.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o nsdp_amd/lib/libnsdp_hip.so $OBJ/*.o || exit 1
python - "$@" <<'PY'
import sys, ctypes, torch
sys.path.insert(0, ".")
from nsdp_amd import hip_linear
from nsdp_amd._lib import lib
M, N, K = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
dy = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev); ones = torch.ones(M, N, device=dev)
buf = (ctypes.c_ulonglong * 8)()
for name, m in (("no mask", None), ("mask", ones)):
    for _ in range(2): hip_linear._wgrad_x3(dy, x, m, False, True)
    lib().nsdp_debug_wg3_timers(buf, 1)
    n = 5
    for _ in range(n): hip_linear._wgrad_x3(dy, x, m, False, True)
    lib().nsdp_debug_wg3_timers(buf, 1)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): hip_linear._wgrad_x3(dy, x, m, False, True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    names = ["producer steps", "plain steps", "bottom wait", "barrier"]
    tot = buf[4]
    print(name, {nm: f"{100.0 * buf[i] / tot:.1f}%" for i, nm in enumerate(names)}, "ticks/wave/launch", tot // n // 1024, f"{ms:.3f} ms -> {tot / n / 1024 / ms / 1e3:.0f} MHz tick")
PY
