mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_plans.py -q 2>&1 | grep -v "^$" | grep "Error\|^E  \|passed\|failed\|FAILED" | head -20 | cut -c1-700 > gpurun_out/r04/test_plans.txt; cat gpurun_out/r04/test_plans.txt
for cfg in "DFINE_GRAPH_CHUNK=5" "DFINE_GRAPH_CHUNK=3" "DFINE_GRAPH_CHUNK=8 DFINE_GRAPH_GROUP_AT=4" "DFINE_GRAPH_CHUNK=5 DFINE_GRAPH_GROUP_AT=32"; do
echo "== $cfg"
env $cfg AB_BLOCKS=6 AB_STEPS=8 timeout 600 python tools/ab_step.py step.hip_graph 2>&1 | grep -v amdgpu | tail -3
done > gpurun_out/r04/ab_graph2.txt 2>&1
cat gpurun_out/r04/ab_graph2.txt
