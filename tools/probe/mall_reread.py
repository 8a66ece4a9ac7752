"""Does a second streaming read of a buffer come out of the 256 MB Infinity Cache (MALL) faster than out of HBM?  Read bandwidth of
`dfine_bn` statistics-style passes (torch.sum over a bf16 buffer; the ATen reduce kernel streams at ~the same rate as bn_stats) for
buffer sizes around the cache size: first read after a 1 GB flush vs immediate re-read.  GPU box."""
import torch
dev = torch.device("cuda", 0)
flush = torch.empty(1 << 30, dtype=torch.uint8, device=dev)


def t(fn, n=1):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


for mb in (13, 26, 52, 105, 157, 210, 420):
    x = torch.randn(mb * (1 << 20) // 2, device=dev).bfloat16() if False else torch.ones(mb * (1 << 20) // 2, device=dev, dtype=torch.bfloat16)
    y = torch.empty_like(x)
    for _ in range(3):
        x.float().sum(); y.copy_(x)
    cold, warm, cw, ww = [], [], [], []
    for _ in range(5):
        flush.fill_(1)
        cold.append(t(lambda: x.sum(dtype=torch.float32)))
        warm.append(t(lambda: x.sum(dtype=torch.float32)))
        flush.fill_(1)
        cw.append(t(lambda: y.copy_(x)))
        ww.append(t(lambda: y.copy_(x)))
    m = lambda v: sorted(v)[len(v) // 2]
    by = mb * (1 << 20)
    print(f"{mb:4d} MB  read cold {by / m(cold) / 1e6:6.2f} TB/s  re-read {by / m(warm) / 1e6:6.2f} TB/s   copy (1R1W) cold {2 * by / m(cw) / 1e6:6.2f}  again {2 * by / m(ww) / 1e6:6.2f} TB/s")
