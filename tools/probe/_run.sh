for cfg in "DFINE_GRAPH_PREFLUSH=0" "DFINE_GRAPH_PREFLUSH=1" "DFINE_GRAPH_PREFLUSH=0" "DFINE_GRAPH_PREFLUSH=1"; do
echo "== $cfg"; env $cfg timeout 600 python bench.py --steps 60 --warmup 10 --cpu-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['median_ms_per_step'])"
done
timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_dist_gpu.py -q 2>&1 | grep "passed\|failed\|Error" | head
