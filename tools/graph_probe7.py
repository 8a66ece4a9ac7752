"""What does a HIP-graph replay cost on this stack?  (GPU box)
A: chains of tiny / medium launches, eager against captured: host time of the enqueue, total time, per-node figures;
   fork / join inside a capture through torch streams and through dfine_stream_fork.
B: backbone + encoder forward of D-FINE-m (bs 32, 640 x 640, bf16): eager against a captured forward (host time, device time).
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda", 0)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    host, total = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
    return min(host), min(total)


def part_a():
    for n, numel, label in ((1000, 1024, "tiny (4 KB add_)"), (300, 13 * 1024 * 1024, "medium (52 MB add_, ~20 us)")):
        x = torch.zeros(numel, device=dev)

        def chain():
            for _ in range(n):
                x.add_(1.0)
        h, t = timeit(chain)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            chain()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            chain()
        hg, tg = timeit(g.replay)
        print(f"A {label}: {n} launches  eager host {h:.2f} ms total {t:.2f} ms ({1e3 * t / n:.2f} us/launch) | "
              f"graph host {hg:.2f} ms total {tg:.2f} ms ({1e3 * tg / n:.2f} us/node)", flush=True)
    # fork / join inside a capture (torch streams)
    x = torch.zeros(1 << 20, device=dev); y = torch.zeros(1 << 20, device=dev)
    side = torch.cuda.Stream()

    def forked(nf=100):
        for _ in range(nf):
            x.add_(1.0)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                y.add_(1.0)
            x.add_(1.0)
        torch.cuda.current_stream().wait_stream(side)
    h, t = timeit(forked)
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            forked()
        hg, tg = timeit(g.replay)
        print(f"A fork/join x100 (torch streams, 4 MB adds): eager host {h:.2f} total {t:.2f} ms | graph host {hg:.2f} total {tg:.2f} ms", flush=True)
    except Exception as e:
        print("A fork/join capture (torch streams) FAILED:", repr(e)[:300], flush=True)
    # dfine_stream_fork inside a capture
    try:
        from custom_d_fine_amd import hip

        def forked2(nf=100):
            for _ in range(nf):
                x.add_(1.0)
                st = hip._side_fork(dev)
                with torch.cuda.stream(st.stream):
                    y.add_(1.0)
                x.add_(1.0)
            hip._SIDE_LIVE.append((x,))
            hip.side_join()
        h, t = timeit(forked2)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            forked2()
        hg, tg = timeit(g.replay)
        print(f"A fork/join x100 (dfine_stream_fork): eager host {h:.2f} total {t:.2f} ms | graph host {hg:.2f} total {tg:.2f} ms", flush=True)
    except Exception as e:
        print("A fork/join capture (dfine_stream_fork) FAILED:", repr(e)[:300], flush=True)


def part_b():
    import bench
    from custom_d_fine_amd import kernels
    from custom_d_fine_amd.dl.engine import _BackboneEncoder
    from custom_d_fine_amd.dl.synthetic import make_batch
    step = bench.build_step("m", 640, dev, torch.bfloat16)
    images, targets = make_batch(32, 640, seed=42, device=dev)
    be = _BackboneEncoder(step.model.backbone, step.model.encoder)

    def fwd():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return be(images)

    def fwd_grad():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return be(images)
    for name, f in (("no_grad", fwd), ("grad", fwd_grad)):
        for _ in range(3):
            f()
        h, t = timeit(f)
        print(f"B eager forward ({name}): host {h:.2f} ms total {t:.2f} ms", flush=True)
    # captured forward (no_grad: only the launch stream of kernels matters here); weight packs served from the caches
    kernels._CAPTURE_POSSIBLE, kernels._CAPTURE_FROZEN_WEIGHTS = True, True
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fwd()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fwd()
    hg, tg = timeit(g.replay)
    print(f"B captured forward (no_grad): replay host {hg:.2f} ms total {tg:.2f} ms", flush=True)
    ref = fwd()
    g.replay()
    torch.cuda.synchronize()
    print("B captured == eager:", [float((a.float() - b.float()).abs().max()) for a, b in zip(out, ref)], flush=True)
    kernels._CAPTURE_POSSIBLE, kernels._CAPTURE_FROZEN_WEIGHTS = False, False


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "ab"
    if "a" in which:
        part_a()
    if "b" in which:
        part_b()
