mkdir -p gpurun_out/r04
STEP_PROFILE_TOP=60 timeout 600 python tools/step_profile.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > gpurun_out/r04/step_profile.txt
head -90 gpurun_out/r04/step_profile.txt
