python -c "
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)
for p in (-2,-1,0,1,2):
    try:
        s=torch.cuda.Stream(priority=p); print(p, 'ok', s.priority)
    except Exception as e: print(p, 'err', str(e)[:80])
"
python tools/ab_step.py hip._SIDE_PRIORITY=0,1 2>/dev/null | tail -3
python tools/ab_step.py hip._SIDE_PRIORITY=0,-1 2>/dev/null | tail -3
timeout 600 python -m pytest tests/test_dist_gpu.py tests/test_optim_gpu.py -x -q 2>&1 | tail -1
