"""The bf16 train step runs on the build's own kernels only: one profiled step per configuration - D-FINE-n (head dim 16),
D-FINE-m (the headline model) and D-FINE-x + mask head (head dim 48 in the encoder's AIFI, six decoder layers, MaskDecoder) -
must launch no rocBLAS / hipBLASLt (Tensile `Cijk_`) GEMM, no MIOpen convolution and no library attention kernel
(the fp32 counterpart: tests/test_gemm_f32_gpu.py::test_fp32_train_step_launches_no_library_gemm)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

LIBRARY = ("Cijk_", "rocblas", "gemv", "miopen", "Miopen", "MIOpen", "igemm", "aotriton", "flash", "ck_tile", "batched_transpose",
           "naive_conv", "SubTensorOpWithScalar", "gemm_kernel", "attn_fwd_", "attn_bwd_")


@pytest.mark.parametrize("name,img,bs,mask,graph", [("n", 320, 4, False, False), ("m", 320, 2, False, True), ("x", 320, 2, True, False)])
def test_bf16_train_step_launches_no_library_kernel(cuda, name, img, bs, mask, graph):
    import bench
    from custom_d_fine_amd import kernels
    from custom_d_fine_amd.dl.synthetic import make_batch
    step = bench.build_step(name, img, cuda, torch.bfloat16, mask=mask)
    step.hip_graph = graph
    images, targets = make_batch(bs, img, seed=7, device=cuda, with_masks=mask)
    try:
        for _ in range(2):
            step(images, targets)
        torch.cuda.synchronize()
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
            loss, _ = step(images, targets)
            torch.cuda.synchronize()
    finally:
        kernels.flush_bn_counters()
        kernels.defer_bn_counters(False)
    assert torch.isfinite(loss)
    names = [e.key for e in prof.key_averages()]
    bad = sorted({n[:110] for n in names if any(p in n for p in LIBRARY) and "dfine::" not in n})
    assert not bad, bad
    assert any("dfine::attn_fwd" in n for n in names) and any("dfine::conv1x1_glds_kernel" in n or "dfine::conv" in n for n in names)
