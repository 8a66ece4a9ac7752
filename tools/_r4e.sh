timeout 900 python -m pytest tests/test_model_gpu.py tests/test_dist_gpu.py tests/test_optim_gpu.py -x -q 2>&1 | tail -2
python tools/ab_step.py hip.WGRAD_STREAM 2>/dev/null | tail -3
python tools/step_aten_shapes.py embedding 5 2>/dev/null | tail -4
