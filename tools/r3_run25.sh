cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf $R/gpurun_out/prof_r3_replay
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r3_replay -o r3_replay -- python $R/bench.py --reps 1 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_r3_replay.log 2>&1
find $R/gpurun_out/prof_r3_replay -name "*_kernel_trace.csv" -delete
find $R/gpurun_out/prof_r3_replay -name "*_agent_info.csv" -delete
ls -la $R/gpurun_out/prof_r3_replay; tail -c 400 $R/gpurun_out/prof_r3_replay.log
