#!/bin/bash
# A/B the compile-time knobs of the fused decoder kernel on the GPU box:  tools/tune_decoder.sh "-DNSDP_DEC_PREFETCH=4" "-DNSDP_DEC_PREFETCH=6" ...
cd "$(dirname "$0")/.."
OBJ=nsdp_amd/lib/obj
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=fast $flags \
      -c nsdp_amd/csrc/decoder_fused.hip -o $OBJ/decoder_fused.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o nsdp_amd/lib/libnsdp_hip.so $OBJ/*.o || exit 1
  echo "== $flags"
  timeout 300 python bench.py --workload dense_inference --batch 8 --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']['decoder_fwd_kernel']; print(d['ms_per_step'], k)"
done
