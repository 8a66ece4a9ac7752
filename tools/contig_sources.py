"""Call sites of .contiguous() / .to(dtype) / .float() calls that really copy, and of torch.cat, in one steady-state train
step, with tensor shapes (GPU box).  Backward-pass calls are attributed to the autograd Function's backward frame."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from custom_d_fine_amd.dl.synthetic import make_batch
dev = torch.device("cuda", 0)
step = bench.build_step("m", 640, dev, torch.bfloat16)
images, targets = make_batch(32, 640, seed=42, device=dev)
for _ in range(4):
    step(images, targets)
torch.cuda.synchronize()
sites = collections.Counter()
byts = collections.Counter()
active = [False]

def site():
    out = []
    for f in reversed(traceback.extract_stack(limit=14)[:-2]):
        if "custom_d_fine_amd" in f.filename:
            out.append(f"{f.filename.split('custom_d_fine_amd/')[-1]}:{f.lineno}")
            if len(out) == 2:
                break
    return " <- ".join(out) or "?"

orig_contig = torch.Tensor.contiguous
def contiguous(self, *a, **k):
    if active[0] and self.is_cuda and not self.is_contiguous():
        key = ("contiguous", tuple(self.shape), str(self.dtype)[6:], site())
        sites[key] += 1; byts[key] += self.numel() * self.element_size() * 2
    return orig_contig(self, *a, **k)
torch.Tensor.contiguous = contiguous
orig_to = torch.Tensor.to
def to(self, *a, **k):
    if active[0] and self.is_cuda and a and isinstance(a[0], torch.dtype) and a[0] != self.dtype and self.numel() > 4096:
        key = ("to " + str(a[0])[6:], tuple(self.shape), str(self.dtype)[6:], site())
        sites[key] += 1; byts[key] += self.numel() * (self.element_size() + torch.empty(0, dtype=a[0]).element_size())
    return orig_to(self, *a, **k)
torch.Tensor.to = to
orig_float = torch.Tensor.float
def float_(self, *a, **k):
    if active[0] and self.is_cuda and self.dtype != torch.float32 and self.numel() > 4096:
        key = ("float", tuple(self.shape), str(self.dtype)[6:], site())
        sites[key] += 1; byts[key] += self.numel() * (self.element_size() + 4)
    return orig_float(self, *a, **k)
torch.Tensor.float = float_
orig_cat = torch.cat
def cat(ts, *a, **k):
    ts = list(ts)
    if active[0] and ts and ts[0].is_cuda:
        n = sum(t.numel() * t.element_size() for t in ts)
        if n > 1 << 16:
            key = ("cat", tuple(ts[0].shape), f"x{len(ts)}", site())
            sites[key] += 1; byts[key] += 2 * n
    return orig_cat(ts, *a, **k)
torch.cat = cat
active[0] = True
step(images, targets)
torch.cuda.synchronize()
active[0] = False
for key, b in sorted(byts.items(), key=lambda kv: -kv[1])[:50]:
    print(f"{b / 1e6:8.1f} MB x{sites[key]:3d} {key[0]:12s} {str(key[1]):24s} {key[2]:9s} {key[3]}")
print("total MB", sum(byts.values()) / 1e6)
