#!/bin/bash
# timing-only ablations of the bf16x3 weight-gradient kernel (wrong results): tools/ablate_wgrad.sh M N K
cd "$(dirname "$0")/.."
OBJ=nsdp_amd/lib/obj
for flags in "" "-DWG3_ABLATE_NO_SPLIT" "-DWG3_ABLATE_NO_LOADS" "-DWG3_ABLATE_NO_SPLIT -DWG3_ABLATE_NO_LOADS"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=fast $flags \
      -c nsdp_amd/csrc/wgrad_bf16x3.hip -o $OBJ/wgrad_bf16x3.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o nsdp_amd/lib/libnsdp_hip.so $OBJ/*.o || exit 1
  python - "$@" "$flags" <<'PY'
import sys, torch
sys.path.insert(0, ".")
from nsdp_amd import hip_linear
M, N, K = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
dy = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev)
for _ in range(3): hip_linear._wgrad_x3(dy, x, None, False, True)
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): hip_linear._wgrad_x3(dy, x, None, False, True)
e.record(); torch.cuda.synchronize()
print(f"{sys.argv[4] or '(full)':50s} {s.elapsed_time(e) / 10:.3f} ms")
PY
done
# leave the in-tree library as the normal build
touch nsdp_amd/csrc/wgrad_bf16x3.hip && python -m nsdp_amd.build > /dev/null
