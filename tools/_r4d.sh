timeout 600 python -m pytest tests/test_conv_epilogue_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -2
python tools/ab_step.py hip._SIDE_GROUP_AT=12,48 2>/dev/null | tail -3
python tools/ab_step.py hip._SIDE_GROUP_AT=12,1000 2>/dev/null | tail -3
python tools/ab_step.py hip.WGRAD_STREAM 2>/dev/null | tail -3
python tools/ab_step.py env:DFINE_GRAD_FANIN 2>/dev/null | tail -3
