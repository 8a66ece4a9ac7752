import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from custom_d_fine_amd.dl.synthetic import make_batch
dev = torch.device("cuda", 0)
step = bench.build_step("m", 640, dev, torch.bfloat16)
images, targets = make_batch(32, 640, seed=42, device=dev)
for _ in range(5): step(images, targets)
torch.cuda.synchronize()
for seg in range(8):
    t0 = time.perf_counter()
    for _ in range(10): step(images, targets)
    torch.cuda.synchronize()
    print(f"steps {5+seg*10:3d}-{14+seg*10:3d}: {(time.perf_counter()-t0)*100:.2f} ms/step, mem {torch.cuda.memory_allocated()/1e9:.2f} GB reserved {torch.cuda.memory_reserved()/1e9:.2f} GB", flush=True)
