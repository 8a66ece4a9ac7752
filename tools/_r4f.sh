timeout 900 python -m pytest tests/test_model_gpu.py tests/test_dist_gpu.py tests/test_mask_path.py -x -q 2>&1 | tail -2
python tools/ab_step.py env:DFINE_DN_PREPARE 2>/dev/null | tail -3
python tools/ab_step.py env:DFINE_DN_PREPARE 2>/dev/null | tail -3
