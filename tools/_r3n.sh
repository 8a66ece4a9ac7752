mkdir -p gpurun_out/r3p
timeout 900 python -m pytest tests/test_conv_mfma_gpu.py tests/test_conv_units_gpu.py tests/test_optim_gpu.py tests/test_data_device.py -x -q > gpurun_out/r3p/tests.log 2>&1; tail -3 gpurun_out/r3p/tests.log
timeout 600 python tools/conv_survey.py > gpurun_out/r3p/conv_survey.txt 2>&1
tail -1 gpurun_out/r3p/conv_survey.txt
python bench.py --cpu-steps 0 > gpurun_out/r3p/bench.json 2> gpurun_out/r3p/bench.err; tail -2 gpurun_out/r3p/bench.err
cut -c1-300 gpurun_out/r3p/bench.json
