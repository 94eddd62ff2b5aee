#!/bin/bash
# phase timing of the bf16x3 forward kernel (instrumented build, GPU box only): tools/time_x3.sh M K N
cd "$(dirname "$0")/.."
OBJ=nsdp_amd/lib/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=fast -DNSDP_X3_TIMING \
    -c nsdp_amd/csrc/gemm_bf16x3.hip -o $OBJ/gemm_bf16x3.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o nsdp_amd/lib/libnsdp_hip.so $OBJ/*.o || exit 1
python - "$@" <<'PY'
import sys, ctypes, torch
sys.path.insert(0, ".")
from nsdp_amd.hip_linear import _fwd_x3, pack_weight_x3
from nsdp_amd._lib import lib
M, K, N = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
w3 = pack_weight_x3(w)[0]
buf = (ctypes.c_ulonglong * 8)()
for dbg in (0, 64, 32, 96):
    lib().nsdp_debug_set(6, dbg)
    for _ in range(2): _fwd_x3(x, w3, N, b, None, None, None, False, True)
    lib().nsdp_debug_x3_timers(buf, 1)
    n = 5
    for _ in range(n): _fwd_x3(x, w3, N, b, None, None, None, False, True)
    lib().nsdp_debug_x3_timers(buf, 1)
    names = ["mfma steps", "bottom wait", "barrier", "epilogue", "tile prologue", "total"]
    tot = buf[5]
    print("dbg", dbg, {nm: f"{100.0 * buf[i] / tot:.1f}%" for i, nm in enumerate(names[:5])}, "ticks/wave/launch", tot // n)
PY
