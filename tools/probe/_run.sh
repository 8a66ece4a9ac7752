timeout 900 python -m pytest tests/test_conv_mfma_gpu.py tests/test_conv_units_gpu.py tests/test_graph_gpu.py tests/test_optim_gpu.py -q -n 2 2>&1 | grep "passed\|failed\|Error" | head -5
for cfg in "DFINE_WGRAD1_GROUP_WGS=256" "DFINE_WGRAD1_GROUP_WGS=64" "DFINE_WGRAD1_GROUP_WGS=32" "DFINE_WGRAD1_GROUP_WGS=256" "DFINE_WGRAD1_GROUP_WGS=64" "DFINE_WGRAD1_GROUP_WGS=128"; do
echo "== $cfg"; env $cfg timeout 600 python bench.py --steps 60 --warmup 10 --cpu-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['median_ms_per_step'])"
done
