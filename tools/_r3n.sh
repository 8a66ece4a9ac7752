mkdir -p gpurun_out/r3p
timeout 1500 python -m pytest tests/test_conv_units_gpu.py tests/test_model_gpu.py tests/test_optim_gpu.py tests/test_dist_gpu.py tests/test_deploy.py tests/test_infer.py -x -q > gpurun_out/r3p/tests.log 2>&1; tail -3 gpurun_out/r3p/tests.log
python bench.py --cpu-steps 0 > gpurun_out/r3p/bench.json 2> gpurun_out/r3p/bench.err; tail -2 gpurun_out/r3p/bench.err
cut -c1-300 gpurun_out/r3p/bench.json
DFINE_FUSE_CONV_BN=0 python bench.py --cpu-steps 0 2>/dev/null | cut -c1-300
python tools/host_profile.py --rows 5 2>&1 | grep "un-profiled"
