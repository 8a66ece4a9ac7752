mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_model_gpu.py -q -x -k "decoder_blocks" 2>&1 | grep -v "^$" | grep "Error\|^E  \|passed\|failed\|FAILED" | head -30 | cut -c1-900 > gpurun_out/r04/test_model.txt; cat gpurun_out/r04/test_model.txt
