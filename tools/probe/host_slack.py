"""Is the eager decoder / criterion stretch of the step paced by the host?  A busy-wait of X ms is added on the host inside the
decoder's forward; the step time grows by ~X if the host is on the critical path there and stays put while the device still has
queued work (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from custom_d_fine_amd.dl.synthetic import make_batch
from custom_d_fine_amd.d_fine.arch import dfine_decoder
dev = torch.device("cuda", 0)
step = bench.build_step("m", 640, dev, torch.bfloat16)
images, targets = make_batch(32, 640, seed=42, device=dev)
DELAY = [0.0]
orig = dfine_decoder.TransformerDecoder.forward
def slowed(self, *a, **k):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < DELAY[0]:
        pass
    return orig(self, *a, **k)
dfine_decoder.TransformerDecoder.forward = slowed
for _ in range(8):
    step(images, list(targets))
for d in (0.0, 0.5, 1.0, 2.0, 4.0, 0.0):
    DELAY[0] = d * 1e-3
    for _ in range(3):
        step(images, list(targets))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        step(images, list(targets))
    torch.cuda.synchronize()
    print(f"host delay {d:4.1f} ms in the decoder forward: {(time.perf_counter() - t0) / 30 * 1e3:7.3f} ms per step")
