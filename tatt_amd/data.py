"""Synthetic TextZoom-shaped batches (BASELINE.json: the benchmark uses synthetic 16x64 -> 32x128 data).
Shapes and the mask rule follow the reference's collate (dataset/dataset.py:282-325, :677-693): 4 channels =
RGB + binary mask (gray <= mean(gray)); text prior = softmax over K of recogniser-like peaky logits."""
import torch


def synthetic_batch(B, K=37, seed=123, device="cuda", H=16, W=64):
    g = torch.Generator().manual_seed(seed)
    yy = torch.arange(2 * H).view(1, 1, 2 * H, 1).float()
    xx = torch.arange(2 * W).view(1, 1, 1, 2 * W).float()
    f1 = 0.05 + 0.55 * torch.rand(B, 3, 1, 1, generator=g)
    f2 = 0.05 + 0.85 * torch.rand(B, 3, 1, 1, generator=g)
    ph = 6.28 * torch.rand(B, 3, 1, 1, generator=g)
    rgb = (0.5 + 0.35 * torch.sin(f1 * xx + ph) * torch.cos(f2 * yy) + 0.1 * torch.randn(B, 3, 2 * H, 2 * W, generator=g)).clamp(0, 1)

    def with_mask(img):
        gray = 0.299 * img[:, 0] + 0.587 * img[:, 1] + 0.114 * img[:, 2]
        mask = (gray <= gray.mean((1, 2), keepdim=True)).float()
        return torch.cat([img, mask.unsqueeze(1)], 1)

    hr = with_mask(rgb)
    lr = with_mask(rgb.view(B, 3, H, 2, W, 2).mean((3, 5)))
    logits = 3.0 * torch.randn(26, B, K, generator=g)
    tp = torch.softmax(logits, -1).permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)     # [B,K,1,26] view, as the trainer builds it
    return hr.to(device), lr.to(device), tp.to(device)
