"""Which .contiguous() calls actually copy during one train step, by call site (GPU box):  python tools/contig_audit.py"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from custom_d_fine_amd.dl.synthetic import make_batch
dev = torch.device("cuda", 0)
step = bench.build_step("m", 640, dev, torch.bfloat16)
images, targets = make_batch(32, 640, seed=42, device=dev)
for _ in range(3): step(images, targets)
torch.cuda.synchronize()
log = collections.Counter(); mb = collections.Counter()
orig = torch.Tensor.contiguous
def patched(self, *a, **k):
    if self.is_cuda and not self.is_contiguous():
        st = traceback.extract_stack(limit=3)[0]
        key = (f"{os.path.basename(st.filename)}:{st.lineno} {st.name}", tuple(self.shape), str(self.dtype).replace("torch.", ""))
        log[key] += 1; mb[key] += self.numel() * self.element_size() / 1e6
    return orig(self, *a, **k)
torch.Tensor.contiguous = patched
step(images, targets)
torch.Tensor.contiguous = orig
torch.cuda.synchronize()
print(f"copying .contiguous() calls in one step: {sum(log.values())}, {sum(mb.values()):.0f} MB written")
for k, n in sorted(log.items(), key=lambda kv: -mb[kv[0]])[:40]:
    print(f"{mb[k]:8.1f} MB x{n:3d} {k[0]:45s} {list(k[1])} {k[2]}")
