"""Token-stream GEMM (csrc/gemm.hip) and attention (csrc/attn.hip) kernels against plain fp32 PyTorch on the same
bf16-rounded operands.  Tolerances: outputs are bf16 (8 mantissa bits) of fp32-accumulated sums -> 1e-2 relative to the
tensor's scale; gradients likewise.  Reference ops: F.linear (+ activation), F.scaled_dot_product_attention,
nn.MultiheadAttention (what src/d_fine/arch/dfine_decoder.py:33-46,200,214-271 and hybrid_encoder.py:243-290 call)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(got, ref, tol=1e-2, what=""):
    got, ref = got.float(), ref.float()
    scale = ref.abs().max().item() + 1e-6
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


ACTS = {0: lambda t: t, 1: F.relu, 2: F.gelu, 3: F.silu}


@pytest.mark.parametrize("M,K,N", [(15744, 256, 256), (1000, 4, 512), (777, 20, 64), (513, 64, 1), (4096, 1024, 256),
                                   (3000, 256, 132), (2048, 256, 80), (130, 512, 1024), (5, 33, 7)])
@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_linear_act_forward(cuda, M, K, N, act):
    from custom_d_fine_amd import hip
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).bfloat16().to(cuda)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().to(cuda)
    b = torch.randn(N, generator=g).to(cuda)
    ref = ACTS[act](F.linear(x.float(), w.float(), b))
    _close(hip.linear_act(x, w, b, act), ref, what="bf16 out")
    _close(hip.linear_act(x, w, None, act, out_f32=True), ACTS[act](F.linear(x.float(), w.float())), tol=2e-3, what="fp32 out, no bias")


def test_linear_act_strided_operands(cuda):
    """Row slices / column slices of larger buffers (packed in_proj weight, transposed shadows, packed qkv gradients)."""
    from custom_d_fine_amd import hip
    g = torch.Generator().manual_seed(3)
    big_w = (torch.randn(768, 256, generator=g) / 16).bfloat16().to(cuda)
    big_wt = big_w.t().contiguous()                                   # [256, 768]
    x = torch.randn(1999, 256, generator=g).bfloat16().to(cuda)
    _close(hip.linear_act(x, big_w[:512]), F.linear(x.float(), big_w[:512].float()))
    dy = torch.randn(1999, 768, generator=g).bfloat16().to(cuda)
    _close(hip.linear_act(dy[:, :512], big_wt[:, :512]), dy[:, :512].float() @ big_w[:512].float())
    _close(hip.linear_act(dy[:, 512:], big_wt[:, 512:]), dy[:, 512:].float() @ big_w[512:].float())


@pytest.mark.parametrize("act", [None, "relu", "gelu", "silu"])
def test_linear_autograd_matches_torch(cuda, act):
    from custom_d_fine_amd import kernels
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 333, 256, generator=g).to(cuda).requires_grad_(True)
    w = (torch.randn(132, 256, generator=g) / 16).to(cuda).requires_grad_(True)
    b = torch.randn(132, generator=g).to(cuda).requires_grad_(True)
    go = torch.randn(4, 333, 132, generator=g).to(cuda)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = kernels.linear(x, w, b, act=act)
    assert y.dtype == torch.bfloat16
    y.backward(go.bfloat16())
    xr = x.detach().bfloat16().float().requires_grad_(True)
    wr = w.detach().bfloat16().float().requires_grad_(True)
    br = b.detach().clone().requires_grad_(True)
    fn = {None: lambda t: t, "relu": F.relu, "gelu": F.gelu, "silu": F.silu}[act]
    yr = fn(F.linear(xr, wr, br))
    yr.backward(go.bfloat16().float())
    _close(y, yr, what="y")
    _close(x.grad, xr.grad, tol=2e-2, what="dx")
    _close(w.grad, wr.grad, tol=2e-2, what="dw")
    _close(b.grad, br.grad, tol=2e-2, what="db")


def _dn_mask(L, dn, groups):
    """Denoising-style mask (True = blocked): normal queries do not see dn queries, dn groups do not see each other."""
    m = torch.zeros(L, L, dtype=torch.bool)
    m[dn:, :dn] = True
    per = dn // groups
    for i in range(groups):
        m[i * per:(i + 1) * per, : i * per] = True
        m[i * per:(i + 1) * per, (i + 1) * per: dn] = True
    return m


def _attn_ref(q, k, v, mask, H):
    B, L, E = q.shape
    hd = E // H
    qh, kh, vh = (t.float().view(B, L, H, hd).transpose(1, 2) for t in (q, k, v))
    o = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=None if mask is None else ~mask)
    return o.transpose(1, 2).reshape(B, L, E)


@pytest.mark.parametrize("B,L,masked", [(2, 496, True), (3, 400, False), (2, 77, True), (1, 900, False), (2, 300, True), (1, 1030, True), (2, 498, True)])
def test_attention_forward_backward(cuda, B, L, masked):
    from custom_d_fine_amd import hip
    H, E = 8, 256
    g = torch.Generator().manual_seed(L)
    qk = (torch.randn(B, L, 2 * E, generator=g) * 1.5).bfloat16().to(cuda)        # packed projection output: q | k
    v = torch.randn(B, L, E, generator=g).bfloat16().to(cuda)
    do = torch.randn(B, L, E, generator=g).bfloat16().to(cuda)
    mask = _dn_mask(L, (L // 3) // 4 * 4, 4).to(cuda) if masked else None
    m8 = None if mask is None else mask.contiguous().view(torch.uint8)
    q, k = qk[..., :E], qk[..., E:]
    o, lse2 = hip.attn_forward(q, k, v, H, m8)
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    ref = _attn_ref(qr, kr, vr, mask, H)
    _close(o, ref, what="o")
    ref.backward(do.float())
    dqk = torch.empty_like(qk)
    dv = torch.empty_like(v)
    hip.attn_backward(q, k, v, o, do, lse2, H, dqk[..., :E], dqk[..., E:], dv, m8)
    _close(dqk[..., :E], qr.grad, tol=2e-2, what="dq")
    _close(dqk[..., E:], kr.grad, tol=2e-2, what="dk")
    _close(dv, vr.grad, tol=2e-2, what="dv")


@pytest.mark.parametrize("hd,B,L,masked", [(48, 2, 900, False), (48, 2, 300, True), (64, 2, 400, False), (64, 1, 498, True), (40, 1, 77, True)])
def test_attention_head_dims_up_to_64(cuda, hd, B, L, masked):
    """Head dims above 32 (48 = the AIFI layer of D-FINE-x, ref configs.py:182, hybrid_encoder.py:243-290) on the two-slab
    instantiations of the attention kernels (48 / 40 zero-padded to 64) vs fp32 softmax attention."""
    from custom_d_fine_amd import hip
    H = 8
    E = H * hd
    g = torch.Generator().manual_seed(L + hd)
    q, k, v, do = ((torch.randn(B, L, E, generator=g) * (1.2 if i < 2 else 1.0)).bfloat16().to(cuda) for i in range(4))
    mask = _dn_mask(L, (L // 3) // 4 * 4, 4).to(cuda) if masked else None
    m8 = None if mask is None else mask.contiguous().view(torch.uint8)
    o, lse2 = hip.attn_forward(q, k, v, H, m8)
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    ref = _attn_ref(qr, kr, vr, mask, H)
    _close(o, ref, what="o")
    ref.backward(do.float())
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    hip.attn_backward(q, k, v, o, do, lse2, H, dq, dk, dv, m8)
    _close(dq, qr.grad, tol=2e-2, what="dq")
    _close(dk, kr.grad, tol=2e-2, what="dk")
    _close(dv, vr.grad, tol=2e-2, what="dv")


def test_attention_online_softmax_rescale_branch(cuda):
    """Keys are streamed in blocks of 256 with a running max: force the max to jump in the LAST block (a key that matches
    its query far better than anything before) - a wrong rescale of the accumulated output would be O(1) wrong."""
    from custom_d_fine_amd import hip
    B, L, H, E = 1, 700, 8, 256
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, L, E, generator=g)
    k = torch.randn(B, L, E, generator=g)
    v = torch.randn(B, L, E, generator=g)
    k[0, 650] = q[0, 10] * 3.0            # query 10 meets its spike in block 3 of 3
    k[0, 5] = q[0, 20] * 3.0              # query 20 meets it in block 1
    q, k, v = (t.bfloat16().to(cuda) for t in (q, k, v))
    o, _ = hip.attn_forward(q, k, v, H, None)
    _close(o, _attn_ref(q, k, v, None, H), what="o with spikes")


def test_mha_block_matches_nn_multiheadattention(cuda):
    """The whole block (in-proj, attention, out-proj, all parameter gradients) vs nn.MultiheadAttention in fp32."""
    from custom_d_fine_amd import kernels
    E, H, B, L = 256, 8, 2, 496
    torch.manual_seed(0)
    mha = torch.nn.MultiheadAttention(E, H, batch_first=True).to(cuda)
    torch.nn.init.normal_(mha.in_proj_bias, std=0.1)
    torch.nn.init.normal_(mha.out_proj.bias, std=0.1)
    g = torch.Generator().manual_seed(2)
    qk_in = torch.randn(B, L, E, generator=g).to(cuda).requires_grad_(True)
    v_in = torch.randn(B, L, E, generator=g).to(cuda).requires_grad_(True)
    go = torch.randn(B, L, E, generator=g).to(cuda)
    mask = _dn_mask(L, 196, 4).to(cuda)
    params = [mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = kernels.self_attention(qk_in, v_in, *params, H, mask)
    y.backward(go.bfloat16())
    got = [qk_in.grad.clone(), v_in.grad.clone()] + [p.grad.clone() for p in params]
    for t in [qk_in, v_in] + params:
        t.grad = None
    yr, _ = mha(qk_in, qk_in, v_in, attn_mask=mask, need_weights=False)
    yr.backward(go)
    want = [qk_in.grad, v_in.grad] + [p.grad for p in params]
    _close(y, yr, tol=2e-2, what="y")
    for name, a, b in zip(["d_qk", "d_v", "d_in_w", "d_in_b", "d_out_w", "d_out_b"], got, want):
        _close(a, b, tol=3e-2, what=name)


@pytest.mark.parametrize("B,C,shapes", [(2, 256, ((80, 80), (40, 40), (20, 20))), (3, 64, ((6, 12), (5, 8))), (1, 8, ((2, 4),))])
def test_flatten_levels_matches_permute_concat(cuda, B, C, shapes):
    """Encoder maps -> decoder token memory (csrc/layout.hip) and its backward: bit-identical to
    concat(flatten(2).permute(0, 2, 1)) and to the autograd gradient of that composition (pure data movement)."""
    from custom_d_fine_amd import kernels
    torch.manual_seed(C)
    maps = [torch.randn(B, C, h, w, device=cuda).bfloat16().requires_grad_(True) for h, w in shapes]
    ref_maps = [m.detach().clone().requires_grad_(True) for m in maps]
    mem = kernels.flatten_levels(maps)
    want = torch.concat([m.flatten(2).permute(0, 2, 1) for m in ref_maps], 1)
    assert mem.shape == want.shape and torch.equal(mem, want)
    go = torch.randn_like(want)
    mem.backward(go)
    want.backward(go)
    for a, b in zip(maps, ref_maps):
        assert a.grad.is_contiguous() and torch.equal(a.grad, b.grad)


def test_attention_backward_bit_mask_equals_byte_mask(cuda, monkeypatch):
    """The dK / dV kernel's two mask forms (transposed bit-packed words made by dfine_attn_mask_bits, the byte mask itself) give
    the same gradients bit for bit; L not a multiple of 32 and a mask with fully blocked rows / columns included."""
    from custom_d_fine_amd import hip
    torch.manual_seed(9)
    B, H, L, E = 3, 8, 77, 256
    q, k, v, do = (torch.randn(B, L, E, device=cuda).bfloat16() for _ in range(4))
    m = torch.rand(L, L, device=cuda) < 0.3
    m[5, :] = True
    m[:, 11] = True
    m[5, 5] = False
    m8 = m.view(torch.uint8)
    o, lse2 = hip.attn_forward(q, k, v, H, m8)
    out = {}
    for form in ("bits", "bytes"):
        if form == "bytes":
            monkeypatch.setattr(hip, "_mask_bits", lambda mask: None)
        dq, dk, dv = (torch.full_like(q, float("nan")) for _ in range(3))
        hip.attn_backward(q, k, v, o, do, lse2, H, dq, dk, dv, m8)
        out[form] = (dq, dk, dv)
    for a, b in zip(out["bits"], out["bytes"]):
        assert torch.isfinite(a).all() and torch.equal(a, b)


@pytest.mark.parametrize("L,maxT,groups", [(480, 15, 6), (498, 9, 11), (340, 20, 1)])
def test_attention_mask_tile_summaries_are_exact(cuda, monkeypatch, L, maxT, groups):
    """The decoder's denoising mask (ref arch/utils.py:442-455: matching queries do not see the denoising ones, a denoising
    group sees itself and the matching queries) through the tile summaries (blocked tiles skipped, free tiles unmasked) must
    give bit-identical results to the plain masked kernels: a blocked tile only ever adds zeros."""
    from custom_d_fine_amd import hip
    dn = 2 * maxT * groups
    assert L == 300 + dn
    mask = torch.zeros(L, L, dtype=torch.bool, device=cuda)
    mask[dn:, :dn] = True
    for i in range(groups):
        a, b = 2 * maxT * i, 2 * maxT * (i + 1)
        mask[a:b, :a] = True
        mask[a:b, b:dn] = True
    m8 = mask.view(torch.uint8).contiguous()
    B, H, E = 2, 8, 256
    g = torch.Generator(device=cuda).manual_seed(L)
    qkv = torch.randn(B, L, 3 * E, device=cuda, generator=g).bfloat16()
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    do = torch.randn(B, L, E, device=cuda, generator=g).bfloat16()

    def run(summary):
        monkeypatch.setattr(hip, "_MASK_SUMMARY", summary)
        hip._MASK_BITS[0] = None
        o, lse2 = hip.attn_forward(q, k, v, H, m8)
        d = torch.zeros_like(qkv)
        hip.attn_backward(q, k, v, o, do, lse2, H, d[..., :E], d[..., E:2 * E], d[..., 2 * E:], m8)
        return o, lse2, d

    o0, l0, d0 = run(False)
    o1, l1, d1 = run(True)
    hip._MASK_BITS[0] = None
    assert torch.equal(o0, o1) and torch.equal(l0, l1) and torch.equal(d0, d1)
    # and against fp32 softmax attention
    qh, kh, vh = (t.reshape(B, L, H, 32).transpose(1, 2).float() for t in (q, k, v))
    ref = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh, attn_mask=~mask).transpose(1, 2).reshape(B, L, E)
    assert (o1.float() - ref).abs().max() <= 2 ** -6 * ref.abs().max()
