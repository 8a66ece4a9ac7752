python tools/ab_step.py hip.WGRAD_STREAM 2>/dev/null | tail -3
python tools/ab_step.py env:DFINE_GRAD_FANIN 2>/dev/null | tail -3
python tools/ab_step.py env:DFINE_CONV1X1_XIMG 2>/dev/null | tail -3
