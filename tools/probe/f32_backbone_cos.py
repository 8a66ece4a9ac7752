import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from custom_d_fine_amd.d_fine import dfine
from custom_d_fine_amd import kernels
from tests import helpers
g = np.load(f"{helpers.GOLDEN_DIR}/backbone_encoder_m320.npz")
dev = torch.device("cuda", 0)
torch.backends.cuda.matmul.allow_tf32 = False
for mode in ("1", "0"):
    os.environ["DFINE_F32_CONV"] = mode
    kernels.reload_env()
    m = dfine.build_model("m", 80, False, "cpu", img_size=[320, 320])
    m.load_state_dict(helpers.seeded_state_dict(m.state_dict()))
    m = m.to(dev).train()
    feats = m.encoder(m.backbone(helpers.make_images(2, 320).to(dev)))
    loss = 0
    for i, f in enumerate(feats):
        ref = torch.tensor(g[f"feat{i}"].astype(np.float32))
        print(mode, "feat", i, ((f.detach().float().cpu()[:1] - ref).abs().max() / ref.abs().max()).item())
        loss = loss + (f.float() * helpers.make_cotangent(f.shape, 50 + i).to(dev)).sum()
    loss.backward()
    params = dict(m.named_parameters())
    for k in helpers.BACKBONE_ENCODER_GRAD_KEYS:
        ref = torch.tensor(g[f"grad/{k}"].astype(np.float32)) * float(g[f"gscale/{k}"])
        got = helpers.compact_rows(params[k].grad.float().cpu())
        print(mode, k, 1 - torch.nn.functional.cosine_similarity(got.flatten().double(), ref.flatten().double(), dim=0).item(), (got.norm() / ref.norm()).item())
