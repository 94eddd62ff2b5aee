cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r3_graph
for cfg in "" "DEBUG_HIP_FORCE_GRAPH_QUEUES=4" "DEBUG_HIP_FORCE_GRAPH_QUEUES=8" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=64"; do
  for B in 32 8; do
    echo "=== cfg=[$cfg] B=$B" 
    env $cfg timeout 300 python tools/try_graph.py $B f32 2>&1 | grep -E "ms/step|graph replay"
  done
done > gpurun_out/r3_graph/probe.txt 2>&1
cat gpurun_out/r3_graph/probe.txt
