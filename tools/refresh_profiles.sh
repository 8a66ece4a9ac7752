# Collects the evidence files of a round on the GPU box (copied into profiles/ afterwards):   bash tools/refresh_profiles.sh r05
R=${1:-r06}
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
python tools/conv1x1_table.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > $O/conv1x1_table.txt
python tools/conv1x1_table.py --ks 3 2>&1 | grep -v -i "warn\|amdgpu.ids" > $O/conv3x3_table.txt
python tools/bn_microbench.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > $O/bn_microbench.txt
python tools/step_events.py $O/step_events.jsonl > $O/step_events.log 2>&1
python tools/step_phases.py $O/step_events.jsonl > $O/step_phases.txt 2>&1
for sw in kernels.MLP_FUSED kernels.LN_DEFER kernels.CDN_KERNEL kernels.STACK_LAYER_OUTPUTS; do python tools/ab_step.py $sw 2>&1 | tail -3; done > $O/ab_launch_fusions.txt
python tools/f32_step_profile.py 2>&1 | grep -v -i "warn\|amdgpu.ids\|_warn_once" > $O/f32_step_profile.txt
python tools/f32_table.py --full 70 2>&1 | grep -v -i "warn\|amdgpu.ids" > $O/f32_table.txt
python tools/probe/aten_ops.py 2>&1 | grep -v -i "warn\|amdgpu.ids" | head -40 > $O/f32_aten_ops.txt
# HBM traffic counters of the family's reference layer (512 -> 512 @80x80 on the LDS-DMA 1x1 kernel), one --pmc pass per counter group
tools/conv_pmc.sh ${R}_conv_pmc 512 512 80 1 fwd conv1x1_glds > $O/conv1x1_pmc.txt 2>&1
python - $O/conv1x1_pmc.txt $R <<'PY' > $O/conv_pmc.json
import json, re, sys
txt = open(sys.argv[1]).read()
val = lambda k: float(re.search(k + r"\s+([0-9.]+) per dispatch", txt).group(1))
f, w = val("FETCH_SIZE"), val("WRITE_SIZE")
print(json.dumps({"kernel": "conv1x1_glds_kernel<4,2,8,false> (dfine_conv_fwd_bf16)", "shape": {"B": 32, "Cin": 512, "Cout": 512, "H": 80, "W": 80, "dtype": "bf16"},
                  "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "traffic_bytes_per_launch": int(f * 1024 * 2 + w * 1024),
                  "compulsory_bytes_per_launch": 2 * 32 * 6400 * (512 + 512) + 2 * 512 * 512,
                  "source": f"profiles/{sys.argv[2]}_conv1x1_pmc.txt (FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes; gfx950 FETCH_SIZE under-reports wide reads by up to 2x)"}, indent=1))
PY
RCCL_PROBE_GRAPH=1 timeout 300 python tools/probe/rccl_one_rank.py 2>&1 | grep "order of\|backend\|smoke" > $O/rccl_one_rank_order.txt
DFINE_ANCHOR_PRINT=1 timeout 900 python -m pytest tests/test_bf16_anchor_gpu.py -x -q -s 2>&1 | grep "totals\|worst\|gradients\|vs fp32\|vs ATen\|passed\|failed\|^0\.\|^1\." | grep -v "print(" > $O/bf16_anchor_run.txt
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
