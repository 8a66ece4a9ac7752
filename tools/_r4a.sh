mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_msda_gpu.py tests/test_model_gpu.py tests/test_dist_gpu.py -x -q > gpurun_out/r4a/t1.log 2>&1; tail -4 gpurun_out/r4a/t1.log
python tools/ab_step.py hip.MSDA_SPLIT 2>/dev/null | tail -3
python tools/ab_step.py hip.MSDA_SPLIT 2>/dev/null | tail -3
