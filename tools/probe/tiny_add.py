import torch, time
dev = torch.device("cuda")
def t(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for shape in [(32, 300, 4), (32, 492, 4), (32, 492, 256), (32, 492, 8)]:
    a = torch.randn(shape, device=dev).bfloat16(); b = torch.randn(shape, device=dev)
    c = torch.randn(shape, device=dev)
    print(shape, "bf16+f32 %.1f us" % t(lambda: a + b), " f32+f32 %.1f us" % t(lambda: c + b), " bf16+bf16 %.1f us" % t(lambda: a + a),
          " a.float()+b %.1f us" % t(lambda: a.float() + b))
