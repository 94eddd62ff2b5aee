cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
timeout 1200 python -m pytest tests/test_decoder_gpu.py tests/test_attention_gpu.py tests/test_linear_gpu.py -x -q > gpurun_out/r3/tests10.txt 2>&1; tail -8 gpurun_out/r3/tests10.txt
for rep in 1 2; do for v in 0 1; do NSDP_DECODER_TRAIN_FUSED=$v python bench.py --no-cpu-baseline --steps 20 --warmup 3 --reps 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']; print('TRAIN_FUSED=$v', d['ms_per_step'], 'loss', d['final_loss'], 'decoder_fwd', k.get('decoder_fwd_kernel'), 'x3', k.get('linear_bf16x3_kernel'))"; done; done
NSDP_DECODER_TRAIN_FUSED=1 python bench.py --no-cpu-baseline --steps 10 --warmup 2 --reps 1 --batch 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('B=8 TRAIN_FUSED=1', d['ms_per_step'])"
