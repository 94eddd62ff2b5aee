cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests/test_rccl_gpu.py tests/test_bench_gpu.py "tests/test_model_gpu.py::test_full_shape_arbitrary_eval_and_train_step_match_golden" "tests/test_bf16_gpu.py::test_bf16_storage_against_the_full_size_reference_fixtures" -x -q -s > gpurun_out/r3/tests1.txt 2>&1
tail -30 gpurun_out/r3/tests1.txt
timeout 600 python tools/bench_grouping.py > gpurun_out/r3/grouping.txt 2>&1
timeout 600 python bench.py --force-reducer --no-cpu-baseline > gpurun_out/r3/bench_force_reducer_nccl.json 2> gpurun_out/r3/bench_force_reducer.err
tail -c 600 gpurun_out/r3/bench_force_reducer_nccl.json
