timeout 600 python -m pytest tests/test_dist_gpu.py -x -q 2>&1 | tail -1
python - <<'PY' 2>&1 | tail -3
import torch
from custom_d_fine_amd.d_fine import dfine
from custom_d_fine_amd.dl.fused_optim import FusedAdamWEMA
m = dfine.build_model("m", 80, False, "cuda", img_size=[640, 640])
opt = dfine.build_optimizer(m, lr=8e-4, backbone_lr=4e-4, betas=(0.9, 0.999), weight_decay=1.25e-4, base_lr=8e-4)
for mb in (16, 40):
    f = FusedAdamWEMA(m, opt, None, overlap=True, bucket_mb=mb)
    print(mb, [round((b["hi"] - b["lo"]) * 4 / 2**20, 1) for b in f._buckets])
PY
python tools/probe/rccl_one_rank.py 2>&1 | tail -1
