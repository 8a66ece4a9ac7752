timeout 600 python -m pytest tests/test_gemm_attn_gpu.py -x -q 2>&1 | tail -1
DFINE_ATTN_DKDV_KT4=0 timeout 600 python -m pytest tests/test_gemm_attn_gpu.py -x -q 2>&1 | tail -1
for i in 1 2; do for cfg in 1 0; do echo "kt4 $cfg"; LAB_SKIP_LINEAR=1 DFINE_ATTN_DKDV_KT4=$cfg python tools/linear_attn_bench.py 2>/dev/null | tail -2; done; done
for cfg in 1 0 1 0; do DFINE_ATTN_DKDV_KT4=$cfg python bench.py --cpu-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kt4 $cfg', d['value'], d['ms_per_step'], d['median_ms_per_step'], d['max_ms_per_step'])"; done
