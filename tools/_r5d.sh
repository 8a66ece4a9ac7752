for mb in 16 40 1000; do echo "bucket_mb $mb"; DFINE_BUCKET_MB=$mb python tools/probe/ddp_mode_timing.py 2>&1 | tail -1; done
echo "overlap 0"; DFINE_GRAD_OVERLAP=0 python - <<'PY' 2>&1 | tail -1
import os, sys, time
sys.path.insert(0, os.getcwd())
exec(open("tools/probe/ddp_mode_timing.py").read().replace('assert step.fused.overlap, "data-parallel mode not active"', ''))
PY
