#!/bin/bash
# Same-box A/B of two builds of libnsdp_hip.so on the default bench (box-to-box spread is ~0.5 ms, more than most kernel
# changes): interleaved rounds, the in-tree library (A) against another build (B), restored at the end.
#     gpurun -- 'bash tools/ab_libs.sh path/to/libB.so [rounds] [bench args...]'
set -e
B=${1:?usage: ab_libs.sh libB.so [rounds] [bench args...]}; ROUNDS=${2:-3}; shift; shift || true
LIB=nsdp_amd/lib/libnsdp_hip.so
cp $LIB /tmp/ab_A.so; cp "$B" /tmp/ab_B.so
trap 'cp /tmp/ab_A.so $LIB' EXIT
run() {
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 "${@:2}" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$1', d['ms_per_step'], 'ms/step  frac', r['frac'], ' launch ms', r['avg_launch_ms'], ' frac_replayed', r.get('frac_replayed'))"
}
for i in $(seq $ROUNDS); do
  cp /tmp/ab_A.so $LIB; run A "$@"
  cp /tmp/ab_B.so $LIB; run B "$@"
done
