#!/bin/bash
# Round profile on the GPU box: bench lines + rocprofv3 kernel stats + HBM PMC passes -> gpurun_out/, then
# `python tools/summarize_profile.py <tag>` (here) condenses them into profiles/.
#   tools/profile_round.sh r1
tag=${1:-r1}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
python bench.py --eager --no-cpu-baseline > gpurun_out/${tag}_bench_default_eager.json 2>/dev/null
python bench.py --batch 8 --eager --no-cpu-baseline > gpurun_out/${tag}_bench_b8_eager.json 2>/dev/null
python bench.py --dtype bf16 --eager --no-cpu-baseline > gpurun_out/${tag}_bench_forward_bf16_eager.json 2>/dev/null
python bench.py --dtype bf16 --workload arbitrary_train --eager --no-cpu-baseline > gpurun_out/${tag}_bench_arbitrary_bf16_eager.json 2>/dev/null
# one rank, RCCL communicator of one: the flat-bucket all-reduce on the device (two graphs around the collective)
python bench.py --force-reducer --no-cpu-baseline > gpurun_out/${tag}_bench_force_reducer_nccl.json 2>/dev/null
# ... and with the backward cut at the decoder's inputs: head / tail / update graphs, bucket 0's all-reduce under the encoder's backward
python bench.py --force-reducer --dp-overlap on --no-cpu-baseline > gpurun_out/${tag}_bench_force_reducer_nccl_overlap.json 2>/dev/null
python bench.py --batch 8 --no-cpu-baseline > gpurun_out/${tag}_bench_b8.json 2>/dev/null
python bench.py --workload arbitrary_train --no-cpu-baseline > gpurun_out/${tag}_bench_arbitrary.json 2>/dev/null
python bench.py --workload dense_inference --no-cpu-baseline > gpurun_out/${tag}_bench_dense_inference.json 2>/dev/null
python bench.py --workload forward_eval --no-cpu-baseline > gpurun_out/${tag}_bench_forward_eval.json 2>/dev/null
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/${tag}_bench_forward_bf16.json 2>/dev/null
python bench.py --dtype bf16 --workload arbitrary_train --no-cpu-baseline > gpurun_out/${tag}_bench_arbitrary_bf16.json 2>/dev/null
python bench.py --dtype bf16 --batch 8 --no-cpu-baseline > gpurun_out/${tag}_bench_b8_bf16.json 2>/dev/null
# config 3 with FlowArbitrary's first network in fp32 storage (eval L2 against the reference 1e-2 instead of 1.3e-1)
python bench.py --dtype bf16 --workload arbitrary_train --canonicalize-f32 --no-cpu-baseline > gpurun_out/${tag}_bench_arbitrary_bf16_net1f32.json 2>/dev/null
# the middle point: network 1 with a bf16 encoder and an fp32-storage decoder (round 6: no accuracy gain)
python bench.py --dtype bf16 --workload arbitrary_train --canonicalize-decoder-f32 --no-cpu-baseline > gpurun_out/${tag}_bench_arbitrary_bf16_net1dec32.json 2>/dev/null
# every (shape, flags) the step sends to the dense kernels, timed in isolation: the algorithmic bytes of `roofline` are this table's
python tools/profile_linear_shapes.py --json gpurun_out/${tag}_x3_operands.json > gpurun_out/${tag}_linear_shapes.txt 2>/dev/null
# the self-launch path: two ranks on this one GPU over gloo (the RCCL path needs a multi-GPU node: driver-run)
python bench.py --gpus 2 --backend gloo --batch 8 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/${tag}_bench_2ranks_gloo.json
python bench.py --gpus 8 --backend gloo --batch 4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/${tag}_bench_8ranks_gloo.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag} -o ${tag} -- \
  python $R/bench.py --eager --reps 1 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_${tag}.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_replay -o ${tag}_replay -- \
  python $R/bench.py --reps 1 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_${tag}_replay.log 2>&1
NSDP_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_iso -o ${tag}_iso -- \
  python $R/bench.py --eager --reps 1 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_${tag}_iso.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- \
  python $R/bench.py --eager --reps 1 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o w -- \
  python $R/bench.py --eager --reps 1 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch_bf16 -o f -- \
  python $R/bench.py --dtype bf16 --eager --reps 1 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch_bf16.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write_bf16 -o w -- \
  python $R/bench.py --dtype bf16 --eager --reps 1 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_write_bf16.log 2>&1
NSDP_WGRAD_STREAM=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -o s -- \
  python $R/bench.py --eager --reps 1 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_sq.log 2>&1
NSDP_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_bf16 -o ${tag}_bf16 -- \
  python $R/bench.py --dtype bf16 --workload arbitrary_train --eager --reps 1 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_${tag}_bf16.log 2>&1
# only the summaries travel back (gpurun merges at most 64 MiB): the per-dispatch kernel traces are dropped
find $R/gpurun_out -name "*_kernel_trace.csv" -delete
find $R/gpurun_out -name "*_agent_info.csv" -delete
ls -la $R/gpurun_out/prof_${tag} $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/pmc_sq
du -sh $R/gpurun_out
tail -c 600 $R/gpurun_out/${tag}_bench_default.json
