"""(f1) Inference runtime `Torch_model` (reference src/infer/torch_model.py:13-375) on the HIP forward path.
The reference's pre-processing is cv2 (opencv-python, a pip dependency that is not vendored and not installed here), so the
8-bit bilinear resize is pinned by properties + oracle/np_ref.resize_linear_u8 (PARITY UNPINNED against cv2 itself); the box
mapping is checked against a direct numpy restatement of the reference's functions; GPU tests compare the HIP kernels with
the oracle bit-exactly (uint8 image -> identical floats)."""
import numpy as np
import pytest
import torch

from oracle import np_ref


def test_resize_oracle_properties():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(np_ref.resize_linear_u8(img, 37, 53), img)                      # identity
    const = np.full((20, 30, 3), 77, np.uint8)
    assert np.array_equal(np_ref.resize_linear_u8(const, 64, 48), np.full((64, 48, 3), 77, np.uint8))
    ramp = np.tile(np.arange(0, 200, 4, dtype=np.uint8)[None, :, None], (8, 1, 3))        # monotone along x
    up = np_ref.resize_linear_u8(ramp, 16, 125).astype(int)
    assert (np.diff(up, axis=1) >= 0).all() and up.min() == 0 and up.max() == 196
    # exact 2x down-sampling of an even image = mean of 2x2 blocks (coefficients are exactly 1/2), rounded half up
    small = rng.integers(0, 256, (8, 8, 1), dtype=np.uint8)
    want = (small.reshape(4, 2, 4, 2, 1).astype(int).sum((1, 3)) + 2) >> 2
    assert np.abs(np_ref.resize_linear_u8(small, 4, 4).astype(int) - want).max() <= 1


def test_letterbox_geometry_matches_reference_formula():
    from custom_d_fine_amd.infer.torch_model import letterbox_geometry
    resized, tl, pads = letterbox_geometry((1100, 1000), (640, 640))
    assert resized == (640, 582) and tl == (0, 29) and pads == (0, 58)
    resized, tl, pads = letterbox_geometry((480, 640), (640, 640))
    assert resized == (480, 640) and tl == (80, 0) and pads == (160, 0)


def _ref_process_boxes(boxes, processed_sizes, orig_sizes, keep_ratio):
    """numpy float32 restatement of Torch_model.process_boxes + norm_xywh_to_abs_xyxy / scale_boxes(_ratio_kept)
    (src/infer/torch_model.py:87-102,420-480)."""
    out = np.zeros_like(boxes, dtype=np.float32)
    for i in range(boxes.shape[0]):
        ph, pw = processed_sizes[i]
        b = boxes[i].astype(np.float32)
        xc, yc, bw, bh = b[:, 0] * np.float32(pw), b[:, 1] * np.float32(ph), b[:, 2] * np.float32(pw), b[:, 3] * np.float32(ph)
        x0 = np.maximum(np.floor(xc - bw / 2), 1); y0 = np.maximum(np.floor(yc - bh / 2), 1)
        x1 = np.minimum(np.ceil(xc + bw / 2), pw - 1); y1 = np.minimum(np.ceil(yc + bh / 2), ph - 1)
        oh, ow = orig_sizes[i]
        if keep_ratio:
            gain = min(ph / oh, pw / ow)
            padw, padh = round((pw - ow * gain) / 2 - 0.1), round((ph - oh * gain) / 2 - 0.1)
            x0, x1 = (x0 - padw) / np.float32(gain), (x1 - padw) / np.float32(gain)
            y0, y1 = (y0 - padh) / np.float32(gain), (y1 - padh) / np.float32(gain)
            x0, x1, y0, y1 = np.clip(x0, 0, ow), np.clip(x1, 0, ow), np.clip(y0, 0, oh), np.clip(y1, 0, oh)
        else:
            sx, sy = np.float32(ow / pw), np.float32(oh / ph)
            x0, x1, y0, y1 = x0 * sx, x1 * sx, y0 * sy, y1 * sy
        out[i] = np.stack([x0, y0, x1, y1], 1)
    return out


@pytest.mark.parametrize("keep_ratio", [False, True])
def test_torch_model_process_boxes(keep_ratio):
    from custom_d_fine_amd.infer.torch_model import Torch_model
    rng = np.random.default_rng(1)
    boxes = np.concatenate([rng.uniform(0, 1, (3, 50, 2)), rng.uniform(0.01, 0.5, (3, 50, 2))], -1).astype(np.float32)
    ps, osz = [(640, 640)] * 3, [(480, 640), (1080, 1920), (333, 500)]
    got = Torch_model.process_boxes(torch.tensor(boxes), ps, osz, keep_ratio).numpy()
    np.testing.assert_allclose(got, _ref_process_boxes(boxes, ps, osz, keep_ratio), rtol=1e-6, atol=1e-3)


def test_torch_model_call_contract_cpu(oracle_backend):
    from custom_d_fine_amd.infer.torch_model import Torch_model
    torch.manual_seed(0)
    m = Torch_model("n", None, 5, input_width=320, input_height=320, conf_thresh=0.0, keep_ratio=True, device="cpu")
    img = np.random.default_rng(2).integers(0, 256, (200, 300, 3), dtype=np.uint8)
    out = m(img)
    assert len(out) == 1 and set(out[0]) == {"labels", "boxes", "scores"}
    r = out[0]
    assert r["labels"].dtype == torch.int64 and r["boxes"].shape == (300, 4) and r["scores"].shape == (300,)
    assert (r["boxes"][:, [0, 2]] <= 300).all() and (r["boxes"][:, [1, 3]] <= 200).all() and (r["boxes"] >= 0).all()
    batch = np.stack([img, img[::-1].copy()])
    out2 = m(batch)
    assert len(out2) == 2 and torch.equal(out2[0]["labels"], r["labels"])
    m2 = Torch_model("n", None, 5, input_width=320, input_height=320, conf_thresh=[0.5, 0.9, 0.5, 0.5, 1.1], device="cpu")
    r2 = m2(img)[0]
    assert (r2["scores"] >= torch.tensor([0.5, 0.9, 0.5, 0.5, 1.1])[r2["labels"]]).all() and not (r2["labels"] == 4).any()


@pytest.mark.gpu
@pytest.mark.parametrize("src_hw,out_hw,resized,tl", [((480, 640), (640, 640), (640, 640), (0, 0)), ((1100, 1000), (640, 640), (640, 582), (0, 29)),
                                                      ((333, 500), (320, 320), (213, 320), (53, 0)), ((640, 640), (640, 640), (640, 640), (0, 0)),
                                                      ((97, 61), (128, 96), (128, 80), (0, 8))])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_hip_preprocess_matches_oracle(cuda, src_hw, out_hw, resized, tl, dtype):
    from custom_d_fine_amd import kernels
    from oracle import torch_backend
    frames = torch.from_numpy(np.random.default_rng(sum(src_hw)).integers(0, 256, (2, *src_hw, 3), dtype=np.uint8))
    got = kernels.preprocess_frames(frames.to(cuda), out_hw, resized, tl, 114, dtype)
    want = torch_backend.preprocess_frames(frames, out_hw, resized, tl, 114, dtype)
    assert got.dtype == dtype and got.shape == (2, 3, *out_hw)
    assert torch.equal(got.cpu(), want)                    # integer image -> identical values, bit for bit


@pytest.mark.gpu
def test_torch_model_gpu_matches_cpu_oracle(cuda):
    from custom_d_fine_amd.infer.torch_model import Torch_model
    from oracle import torch_backend
    from tests import helpers
    from tests.test_model_cpu import assert_same_query_set
    img = np.random.default_rng(4).integers(0, 256, (2, 240, 360, 3), dtype=np.uint8)
    torch.manual_seed(0)
    g = Torch_model("n", None, 80, input_width=320, input_height=320, conf_thresh=0.0, keep_ratio=True, device="cuda")
    sd = helpers.seeded_state_dict(g.model.state_dict())
    g.model.load_state_dict(sd)
    og = g(img)
    torch_backend.install()
    try:
        c = Torch_model("n", None, 80, input_width=320, input_height=320, conf_thresh=0.0, keep_ratio=True, device="cpu")
        c.model.load_state_dict(sd)
        oc = c(img)
    finally:
        torch_backend.uninstall()
    for a, b in zip(og, oc):
        assert a["boxes"].is_cuda and a["labels"].dtype == torch.int64
        # the same detections as a set (score ties / 1-ulp score differences reorder neighbours)
        ka = torch.cat([a["scores"][:, None].cpu(), a["labels"][:, None].float().cpu()], 1)[None]
        kb = torch.cat([b["scores"][:, None], b["labels"][:, None].float()], 1)[None]
        assert_same_query_set(ka, a["boxes"][None].cpu() / 360, kb, b["boxes"][None] / 360, tol=2e-3, min_frac=0.97)


@pytest.mark.gpu
@pytest.mark.parametrize("half", [False, True])
def test_torch_model_hip_graph_replays_the_eager_forward(cuda, half):
    """Torch_model(hip_graph=True): the network forward of every input shape is captured once and replayed - the results must
    be the eager ones, call after call, for new frames, and for a second batch size (its own graph)."""
    from custom_d_fine_amd.infer.torch_model import Torch_model
    from tests import helpers
    rng = np.random.default_rng(9)
    torch.manual_seed(0)
    e = Torch_model("n", None, 80, input_width=320, input_height=320, conf_thresh=0.0, device="cuda", half=half)
    sd = helpers.seeded_state_dict(e.model.state_dict())
    e.model.load_state_dict(sd)
    g = Torch_model("n", None, 80, input_width=320, input_height=320, conf_thresh=0.0, device="cuda", half=half, hip_graph=True)
    g.model.load_state_dict(sd)
    g._graphs.clear()                                   # graphs of the constructor's test call were captured with other weights
    for shape in ((2, 240, 360, 3), (2, 240, 360, 3), (240, 360, 3), (2, 240, 360, 3)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        want, got = e(img), g(img)
        assert len(want) == len(got)
        for a, b in zip(want, got):
            assert torch.equal(a["labels"], b["labels"]) and torch.equal(a["boxes"], b["boxes"]) and torch.equal(a["scores"], b["scores"])
    assert len(g._graphs) == 2 and all(v is not False for v in g._graphs.values())
