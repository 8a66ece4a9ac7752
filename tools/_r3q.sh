mkdir -p gpurun_out/r3q
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r3q/tests.log 2>&1; tail -3 gpurun_out/r3q/tests.log
python bench.py > gpurun_out/r3q/bench.json 2> gpurun_out/r3q/bench.err; tail -c 600 gpurun_out/r3q/bench.json | head -c 300
python tools/step_profile.py > gpurun_out/r3q/step_profile.txt 2>&1; head -30 gpurun_out/r3q/step_profile.txt
python tools/step_aten_shapes.py add 40 > gpurun_out/r3q/aten_add.txt 2>&1
python tools/step_aten_shapes.py "" 60 > gpurun_out/r3q/aten_all.txt 2>&1
