# Collects the evidence files of a round on the GPU box (copied into profiles/ afterwards):   bash tools/refresh_profiles.sh r05
R=${1:-r05}
O=gpurun_out/$R
mkdir -p $O
python bench.py > $O/bench_line.json 2> $O/bench_line.err
tools/rocprof_stats.sh ${R}_stats python /root/repo/bench.py --steps 50 --warmup 10 --cpu-steps 0 > $O/stats_cmd.log 2>&1
cp gpurun_out/${R}_stats/${R}_stats_kernel_stats.csv $O/bench_kernel_stats.csv
cp gpurun_out/${R}_stats/${R}_stats_domain_stats.csv $O/bench_domain_stats.csv 2>/dev/null
cp gpurun_out/${R}_stats/top_kernels.txt $O/bench_top_kernels.txt
# the bench line's roofline recomputed from the rocprofv3 summary (same kernel-name patterns): profiles/rNN_roofline_from_profile.json
python tools/roofline_from_stats.py $O/bench_kernel_stats.csv $O/bench_line.json | sed "s#$O/bench_kernel_stats.csv#profiles/${R}_bench_kernel_stats.csv#" > $O/roofline_from_profile.json
STEP_PROFILE_TOP=60 python tools/step_profile.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > $O/step_profile.txt
python tools/host_profile.py 2>&1 | grep -v -i "warn\|amdgpu.ids" | head -60 > $O/host_profile.txt
python tools/wgrad3_bench.py 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -7 > $O/wgrad3_bench.txt
DFINE_WGRAD3_ROWS=0 python tools/wgrad3_bench.py 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -7 >> $O/wgrad3_bench.txt
for a in 1 2 3; do DFINE_W3_ABLATE=$a python tools/wgrad3_bench.py 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -7 >> $O/wgrad3_bench.txt; done
RCCL_PROBE_GRAPH=1 timeout 300 python tools/probe/rccl_one_rank.py 2>&1 | grep "order of\|backend\|smoke" > $O/rccl_one_rank_order.txt
DFINE_ANCHOR_PRINT=1 timeout 900 python -m pytest tests/test_bf16_anchor_gpu.py -x -q -s 2>&1 | grep "totals\|worst\|gradients\|vs fp32\|vs ATen\|passed\|failed" | grep -v "print(" > $O/bf16_anchor.txt
python bench.py --model s --batch 16 --dtype fp32 --steps 30 --warmup 8 --cpu-steps 0 > $O/bench_line_s_fp32.json 2>/dev/null
python bench.py --model x --img 960 --batch 8 --mask 1 --steps 20 --warmup 6 --cpu-steps 0 > $O/bench_line_x_mask_960.json 2> $O/bench_line_x_mask_960.err
for f in bench_line bench_line_s_fp32 bench_line_x_mask_960; do python - "$f" "$O" <<'PY'
import json, sys
f, o = sys.argv[1], sys.argv[2]
try:
    d = json.load(open(f"{o}/{f}.json"))
    print(f, d["value"], d["ms_per_step"], d["median_ms_per_step"], d["roofline"]["frac"], d["roofline"].get("bound_frac"))
except Exception as e:
    print(f, "FAILED", e)
PY
done
