# Collects the evidence files of a round on the GPU box (copied into profiles/ afterwards):   bash tools/refresh_profiles.sh r04
R=${1:-r04}
mkdir -p gpurun_out/$R
python bench.py > gpurun_out/$R/bench_line.json 2> gpurun_out/$R/bench_line.err
tools/rocprof_stats.sh ${R}_stats python /root/repo/bench.py --steps 50 --warmup 10 --cpu-steps 0 > gpurun_out/$R/stats_cmd.log 2>&1
STEP_PROFILE_TOP=60 python tools/step_profile.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > gpurun_out/$R/step_profile.txt
python tools/host_profile.py 2>&1 | grep -v -i "warn\|amdgpu.ids" | head -60 > gpurun_out/$R/host_profile.txt
python tools/stream_timeline.py --graph 1 2>&1 | grep -v -i "warn\|amdgpu.ids" > gpurun_out/$R/stream_timeline.txt
python tools/graph_probe8.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > gpurun_out/$R/graph_segment_probe.txt
DFINE_HIPGRAPH=0 python bench.py --steps 50 --warmup 10 --cpu-steps 0 > gpurun_out/$R/bench_line_eager.json 2>/dev/null
DFINE_HIPGRAPH=0 DFINE_DEVICE_PLANS=0 python bench.py --steps 50 --warmup 10 --cpu-steps 0 > gpurun_out/$R/bench_line_eager_hostplans.json 2>/dev/null
python bench.py --model s --batch 16 --dtype fp32 --steps 30 --warmup 8 --cpu-steps 0 > gpurun_out/$R/bench_line_s_fp32.json 2>/dev/null
python bench.py --model x --img 960 --batch 8 --mask 1 --steps 20 --warmup 6 --cpu-steps 0 > gpurun_out/$R/bench_line_x_mask_960.json 2> gpurun_out/$R/bench_line_x_mask_960.err
for f in bench_line bench_line_eager bench_line_eager_hostplans bench_line_s_fp32 bench_line_x_mask_960; do python - "$f" "$R" <<'PY'
import json, sys
f, r = sys.argv[1], sys.argv[2]
try:
    d = json.load(open(f"gpurun_out/{r}/{f}.json"))
    print(f, d["value"], d["ms_per_step"], d["median_ms_per_step"], d["roofline"]["frac"], d["roofline"].get("bound_frac"))
except Exception as e:
    print(f, "FAILED", e)
PY
done
