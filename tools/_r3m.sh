mkdir -p gpurun_out/r3s
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r3s/gpu_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3s/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3s/smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r3s/smoke.log
tail -14 gpurun_out/r3s/gpu_tests.log; tail -2 gpurun_out/r3s/smoke.log
