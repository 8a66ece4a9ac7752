cd /root/repo
for a in 0 1 2 3 4 5 6 7; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDFINE_CONV1X1_ABLATE=$a -Icustom_d_fine_amd/csrc -Iinclude -o /tmp/abl$a tools/probe/conv1x1_ablate.hip custom_d_fine_amd/csrc/conv3s.hip custom_d_fine_amd/csrc/wgrad3.hip 2>/dev/null || echo build $a failed; /tmp/abl$a 512 512 6400 384 768 1600 1280 384 1600; done
