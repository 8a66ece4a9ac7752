DFINE_HIPGRAPH=0 timeout 600 python tools/step_aten_sites.py add add_ mul copy_ to cat clone contiguous _to_copy sum 2>&1 | grep -v -i "warn\|amdgpu" | head -50
