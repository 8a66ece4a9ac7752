"""(f3) Device-side data path - first pieces.

What the reference does per sample on the host inside DataLoader workers (OpenCV + numpy + albumentations,
`src/dl/dataset.py`, `src/dl/utils.py`) and what of it runs here, on the GPU the batch is going to anyway:
  * `parse_yolo_label_file` (dataset.py:31-73): YOLO txt rows (`cls xc yc w h`, or `cls x1 y1 x2 y2 ...` polygons) - host.
  * `get_mosaic_coordinate` (utils.py:392-414), `get_transform_matrix` (utils.py:298-322): host geometry, a handful of
    numbers per sample; the random draws are made in the reference's order from a `random.Random`.
  * mosaic composition + random affine of the IMAGE (dataset.py:258-300, utils.py:339-341): `dfine_mosaic_place_u8` x 4 and
    `dfine_warp_affine_u8` (csrc/data.hip) on uint8 HWC device frames - the pixels never visit the host.
  * random affine of the BOXES + clipping + candidate filter (utils.py:343-377): `dfine_affine_boxes`.
  * multi-scale collate (dataset.py:667-694): bilinear resize of the stacked batch / the masks (`kernels.bilinear_resize`).
Polygons (YOLO-seg) are parsed and carried, their clipping / rasterisation (utils.py:219-275, dataset.py:355-366) is still
missing, as are the albumentations colour augmentations.  `YoloTxtDataset` + `write_synthetic_yolo_dataset` give
BASELINE configs[0] its "16 synthetic YOLO-labelled images on disk".
"""
import math
import random
from pathlib import Path
from typing import List, Tuple

import numpy as np
import torch

from .. import kernels


# ----------------------------------------------------------------------------------------------- labels on disk
def parse_yolo_label_file(path) -> Tuple[np.ndarray, List[np.ndarray]]:
    """-> (boxes_norm [N, 5] f32 = cls, xc, yc, w, h; polys_norm: list of [K, 2] f32 normalised polygons, empty per box row).
    5-column rows are boxes, rows with >= 7 columns polygons (an odd trailing value is dropped), anything else is an error
    (ref dataset.py:31-73)."""
    boxes, polys = [], []
    with open(path, "r") as f:
        for ln, raw in enumerate(f, 1):
            s = raw.strip()
            if not s or s.startswith("#"):
                continue
            parts = s.split()
            cl = float(parts[0])
            nums = [float(x) for x in parts[1:]]
            if len(nums) == 4:
                boxes.append([cl, *nums])
                polys.append(np.empty((0, 2), dtype=np.float32))
            elif len(nums) >= 6:
                if len(nums) % 2 == 1:
                    nums = nums[:-1]
                poly = np.array(nums).reshape(-1, 2)
                polys.append(poly)
                (x0, y0), (x1, y1) = poly.min(axis=0), poly.max(axis=0)
                boxes.append([cl, (x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0])
            else:
                raise ValueError(f"Invalid label line (wrong number of values) {path}:{ln}: {s}")
    if not boxes:
        return np.zeros((0, 5), dtype=np.float32), []
    return np.asarray(boxes, dtype=np.float32), polys


# ----------------------------------------------------------------------------------------------- host geometry
def get_mosaic_coordinate(mosaic_index, xc, yc, w, h, target_h, target_w):
    """((x1, y1, x2, y2) on the 2H x 2W canvas, (x1, y1, x2, y2) inside the resized frame) of quadrant `mosaic_index`
    around the mosaic centre (xc, yc) (ref utils.py:392-414)."""
    if mosaic_index == 0:
        x1, y1, x2, y2 = max(xc - w, 0), max(yc - h, 0), xc, yc
        small = w - (x2 - x1), h - (y2 - y1), w, h
    elif mosaic_index == 1:
        x1, y1, x2, y2 = xc, max(yc - h, 0), min(xc + w, target_w * 2), yc
        small = 0, h - (y2 - y1), min(w, x2 - x1), h
    elif mosaic_index == 2:
        x1, y1, x2, y2 = max(xc - w, 0), yc, xc, min(target_h * 2, yc + h)
        small = w - (x2 - x1), 0, w, min(y2 - y1, h)
    else:
        x1, y1, x2, y2 = xc, yc, min(xc + w, target_w * 2), min(target_h * 2, yc + h)
        small = 0, 0, min(w, x2 - x1), min(y2 - y1, h)
    return (x1, y1, x2, y2), small


def get_transform_matrix(img_shape, new_shape, degrees, scale, shear, translate, rng=random):
    """M = T @ S @ R @ C (3 x 3, float64) and the drawn scale; draws in the reference's order: angle, scale, shear x, shear y,
    translate x, translate y (ref utils.py:298-322; cv2.getRotationMatrix2D(center=(0, 0)) written out)."""
    new_width, new_height = new_shape
    C = np.eye(3)
    C[0, 2], C[1, 2] = -img_shape[1] / 2, -img_shape[0] / 2
    a = rng.uniform(-degrees, degrees)
    s = rng.uniform(1.0 - scale, 1.0 + scale) if isinstance(scale, float) else rng.uniform(scale[0], scale[1])
    R = np.eye(3)
    alpha, beta = s * math.cos(math.radians(a)), s * math.sin(math.radians(a))
    R[0, :2], R[1, :2] = (alpha, beta), (-beta, alpha)
    S = np.eye(3)
    S[0, 1] = math.tan(rng.uniform(-shear, shear) * math.pi / 180)
    S[1, 0] = math.tan(rng.uniform(-shear, shear) * math.pi / 180)
    T = np.eye(3)
    T[0, 2] = rng.uniform(0.5 - translate, 0.5 + translate) * new_width
    T[1, 2] = rng.uniform(0.5 - translate, 0.5 + translate) * new_height
    return T @ S @ R @ C, s


# ----------------------------------------------------------------------------------------------- device operators
def mosaic_affine(frames, labels, target_hw, rng=random, degrees=0.0, translate=0.1, scales=(0.5, 1.5), shear=0.0,
                  keep_ratio=False):
    """Four uint8 HWC device frames + their label arrays ([N_i, 5] = cls, xc, yc, w, h normalised) -> (image uint8 [H, W, 3]
    on the device, labels i64 [M], boxes f32 [M, 4] absolute xyxy in the target frame) - `_load_mosaic` + `random_affine`
    of the reference (dataset.py:258-345, utils.py:325-389) with the pixel work and the box transform on the GPU."""
    hip = kernels._hip()
    th, tw = target_hw
    dev = frames[0].device
    yc = int(rng.uniform(th * 0.6, th * 1.4))
    xc = int(rng.uniform(tw * 0.6, tw * 1.4))
    canvas = torch.full((2 * th, 2 * tw, 3), 114, dtype=torch.uint8, device=dev)
    all_boxes = []
    for i, (img, lab) in enumerate(zip(frames, labels)):
        h0, w0 = img.shape[:2]
        if keep_ratio:
            sh = sw = min(th / h0, tw / w0)
        else:
            sh, sw = th / h0, tw / w0
        w, h = int(w0 * sw), int(h0 * sh)
        (lx1, ly1, lx2, ly2), (sx1, sy1, _, _) = get_mosaic_coordinate(i, xc, yc, w, h, th, tw)
        hip.mosaic_place(img.contiguous(), canvas, (h, w), (lx1, ly1, lx2, ly2), (sx1, sy1))
        padw, padh = lx1 - sx1, ly1 - sy1
        lab = np.asarray(lab, dtype=np.float32).reshape(-1, 5)
        if lab.size:
            b = np.empty_like(lab)
            b[:, 0] = lab[:, 0]
            # normalised cxcywh of the source frame -> absolute xyxy there -> scaled and shifted onto the canvas
            b[:, 1] = sw * (lab[:, 1] - lab[:, 3] / 2) * w0 + padw
            b[:, 2] = sh * (lab[:, 2] - lab[:, 4] / 2) * h0 + padh
            b[:, 3] = sw * (lab[:, 1] + lab[:, 3] / 2) * w0 + padw
            b[:, 4] = sh * (lab[:, 2] + lab[:, 4] / 2) * h0 + padh
            all_boxes.append(b)
    tgt = np.concatenate(all_boxes, 0) if all_boxes else np.zeros((0, 5), dtype=np.float32)
    if len(tgt):
        np.clip(tgt[:, 1], 0, 2 * tw, out=tgt[:, 1]); np.clip(tgt[:, 2], 0, 2 * th, out=tgt[:, 2])
        np.clip(tgt[:, 3], 0, 2 * tw, out=tgt[:, 3]); np.clip(tgt[:, 4], 0, 2 * th, out=tgt[:, 4])
    M, s = get_transform_matrix((2 * th, 2 * tw), (tw, th), degrees, scales, shear, translate, rng)
    image = hip.warp_affine(canvas, M[:2], (th, tw), 114)
    if len(tgt):
        boxes, keep = hip.affine_boxes(torch.from_numpy(tgt[:, 1:5].copy()).to(dev), M[:2], s, (tw, th), 0.1)
        wh = boxes[:, 2:] - boxes[:, :2]
        keep = keep.bool() & (wh.min(1).values > 1)                       # "remove tiny boxes after affine" (dataset.py:338-342)
        cls = torch.from_numpy(tgt[:, 0]).to(dev).long()
        return image, cls[keep], boxes[keep]
    return image, torch.zeros(0, dtype=torch.int64, device=dev), torch.zeros(0, 4, device=dev)


def multiscale_collate(images, targets, offset):
    """The reference's `train_collate_fn` resize (dataset.py:667-694) for a drawn `offset` (one of +-32, +-64): images [B, 3, H, W]
    -> [B, 3, H + offset, W + offset] bilinear; boxes are normalised and stay; masks are resized and re-binarised at 0.5."""
    new_hw = (images.shape[2] + offset, images.shape[3] + offset)
    images = kernels.bilinear_resize(images.contiguous(), new_hw)
    for t in targets:
        m = t.get("masks")
        if m is None or m.numel() == 0:
            continue
        r = kernels.bilinear_resize(m.unsqueeze(1).float().contiguous(), new_hw).squeeze(1)
        t["masks"] = (r > 0.5).to(torch.uint8)
    return images, targets


# ----------------------------------------------------------------------------------------------- a dataset on disk
def write_synthetic_yolo_dataset(root, n_images=16, size=(240, 320), num_classes=3, seed=0):
    """`root/images/*.png` + `root/labels/*.txt` (YOLO rows): coloured rectangles on noise, 1-4 boxes per image."""
    from PIL import Image
    root = Path(root)
    (root / "images").mkdir(parents=True, exist_ok=True)
    (root / "labels").mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(seed)
    h, w = size
    for i in range(n_images):
        img = rng.integers(0, 60, (h, w, 3), dtype=np.uint8)
        rows = []
        for _ in range(int(rng.integers(1, 5))):
            c = int(rng.integers(0, num_classes))
            bw, bh = rng.uniform(0.1, 0.4), rng.uniform(0.1, 0.4)
            cx, cy = rng.uniform(bw / 2 + 0.02, 1 - bw / 2 - 0.02), rng.uniform(bh / 2 + 0.02, 1 - bh / 2 - 0.02)
            x0, x1, y0, y1 = int((cx - bw / 2) * w), int((cx + bw / 2) * w), int((cy - bh / 2) * h), int((cy + bh / 2) * h)
            img[y0:y1, x0:x1] = np.array([60 + 60 * c, 200 - 50 * c, 90 + 40 * c], dtype=np.uint8)
            rows.append(f"{c} {cx:.6f} {cy:.6f} {bw:.6f} {bh:.6f}")
        Image.fromarray(img).save(root / "images" / f"img_{i:03d}.png")
        (root / "labels" / f"img_{i:03d}.txt").write_text("\n".join(rows) + "\n")
    return root


class YoloTxtDataset:
    """Images under `root/images`, labels under `root/labels/<stem>.txt` (missing file = background image).  `batch(indices,
    device)` -> the batch contract of the hot path: images f32 [B, 3, H, W] in [0, 1], targets = list of
    {"labels" i64 [T], "boxes" f32 [T, 4] normalised cxcywh, "orig_size" i64 [2]}.  The uint8 frames are uploaded as they are;
    resize + HWC -> CHW + / 255 is the inference runtime's pre-processing kernel (CUDA) or a PIL resize (CPU plumbing runs)."""

    def __init__(self, root, img_size=(640, 640)):
        self.root, self.img_size = Path(root), tuple(img_size)
        self.paths = sorted(p for p in (self.root / "images").iterdir() if p.suffix.lower() in (".png", ".jpg", ".jpeg", ".bmp"))
        if not self.paths:
            raise FileNotFoundError(f"no images under {self.root / 'images'}")

    def __len__(self):
        return len(self.paths)

    def _load(self, i):
        from PIL import Image
        img = np.asarray(Image.open(self.paths[i]).convert("RGB"))
        lab = self.root / "labels" / (self.paths[i].stem + ".txt")
        boxes, _ = parse_yolo_label_file(lab) if lab.exists() else (np.zeros((0, 5), dtype=np.float32), [])
        return img, boxes

    def batch(self, indices, device, mosaic_prob=0.0, rng=random):
        """mosaic_prob: with that probability a sample becomes the 2 x 2 mosaic of itself and three other images of the folder,
        randomly scaled / translated into the target frame (the reference's `_load_mosaic` + `random_affine`, dataset.py:258-345,
        drawn per sample like dataset.py:386-392) - pixel work and box transform on the device (csrc/data.hip)."""
        device = torch.device(device)
        th, tw = self.img_size
        images, targets = [], []
        for i in indices:
            img, boxes = self._load(i)
            h0, w0 = img.shape[:2]
            if device.type == "cuda" and mosaic_prob > 0 and rng.random() < mosaic_prob:
                picks = [i] + [rng.randrange(len(self)) for _ in range(3)]
                loaded = [(img, boxes)] + [self._load(j) for j in picks[1:]]
                frames = [torch.from_numpy(np.ascontiguousarray(f[..., ::-1])).to(device) for f, _ in loaded]
                canvas, cls, xyxy = mosaic_affine(frames, [b for _, b in loaded], (th, tw), rng)
                images.append(kernels.preprocess_frames(canvas[None], (th, tw), (th, tw))[0])
                scale = torch.tensor([tw, th, tw, th], dtype=torch.float32, device=device)
                cxcywh = torch.cat([(xyxy[:, :2] + xyxy[:, 2:]) / 2, xyxy[:, 2:] - xyxy[:, :2]], 1) / scale
                targets.append({"labels": cls, "boxes": cxcywh, "orig_size": torch.tensor([th, tw], dtype=torch.int64, device=device)})
                continue
            if device.type == "cuda":
                frame = torch.from_numpy(np.ascontiguousarray(img[..., ::-1])).to(device)[None]     # the kernel takes BGR frames
                x = kernels.preprocess_frames(frame, (th, tw), (th, tw))[0]
            else:
                from PIL import Image
                x = torch.from_numpy(np.asarray(Image.fromarray(img).resize((tw, th), Image.BILINEAR))).permute(2, 0, 1).float() / 255
            images.append(x)
            targets.append({"labels": torch.from_numpy(boxes[:, 0]).long().to(device),
                            "boxes": torch.from_numpy(boxes[:, 1:5].copy()).float().to(device),
                            "orig_size": torch.tensor([h0, w0], dtype=torch.int64, device=device)})
        return torch.stack(images), targets
