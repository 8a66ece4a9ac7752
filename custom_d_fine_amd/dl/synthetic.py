"""Synthetic COCO-shaped batches with the collate contract of the reference's loader
(`src/dl/dataset.py:639-662`, consumed at `src/dl/train.py:550-565`):
    images  f32 [B, 3, H, W] in [0, 1]
    targets list of B dicts: labels i64 [T_i], boxes f32 [T_i, 4] normalised cxcywh
            (+ masks u8 [T_i, H, W] for the segment task), orig_size i64 [2]
Definition from BASELINE.md section 3: T_i ~ clamp(Poisson(7.3), 1, 100), labels ~ U{0..C-1},
cx,cy ~ U(0.2,0.8), w,h ~ U(0.05,0.35) (boxes always inside the image), seed 42 + rank.
"""
import torch


def make_batch(batch_size, img_size, num_classes=80, seed=42, device="cpu", with_masks=False):
    g = torch.Generator().manual_seed(seed)
    h, w = (img_size, img_size) if isinstance(img_size, int) else img_size
    images = torch.rand(batch_size, 3, h, w, generator=g)
    counts = torch.poisson(torch.full((batch_size,), 7.3), generator=g).clamp(1, 100).long()
    targets = []
    for n in counts.tolist():
        cxcy = torch.rand(n, 2, generator=g) * 0.6 + 0.2
        wh = torch.rand(n, 2, generator=g) * 0.3 + 0.05
        t = {"labels": torch.randint(0, num_classes, (n,), generator=g),
             "boxes": torch.cat([cxcy, wh], 1),
             "orig_size": torch.tensor([h, w])}
        if with_masks:
            ys = torch.arange(h)[None, :, None]
            xs = torch.arange(w)[None, None, :]
            x1, y1 = ((cxcy - wh / 2) * torch.tensor([w, h])).unbind(1)
            x2, y2 = ((cxcy + wh / 2) * torch.tensor([w, h])).unbind(1)
            t["masks"] = ((xs >= x1[:, None, None]) & (xs < x2[:, None, None]) &
                          (ys >= y1[:, None, None]) & (ys < y2[:, None, None])).to(torch.uint8)
        targets.append(t)
    images = images.to(device)
    targets = [{k: v.to(device) for k, v in t.items()} for t in targets]
    return images, targets
