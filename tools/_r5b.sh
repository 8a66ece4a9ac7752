python tools/ab_step.py main_high_priority 2>/dev/null | tail -3
python tools/ab_step.py main_high_priority 2>/dev/null | tail -3
