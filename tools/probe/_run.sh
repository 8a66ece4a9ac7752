for cfg in "DFINE_STEM_WGRAD_SIDE=1" "DFINE_STEM_WGRAD_SIDE=0" "DFINE_STEM_WGRAD_SIDE=1" "DFINE_STEM_WGRAD_SIDE=0"; do
echo "== $cfg"; env $cfg timeout 600 python bench.py --steps 60 --warmup 10 --cpu-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['median_ms_per_step'])"
done
