mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests/test_stem_gpu.py tests/test_train_trace.py tests/test_model_gpu.py tests/test_no_library_gpu.py -m gpu -q -n 2 2>&1 | grep -v "^$" | grep "Error\|^E  \|passed\|failed\|FAILED" | head -30 | cut -c1-1200 > gpurun_out/r04/test_all.txt; cat gpurun_out/r04/test_all.txt
