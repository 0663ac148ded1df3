"""Oracle: landmark patch gather (closed form of the reference's n sequential grid_sample calls)
and the landmark head's min-max scaling -- fp32 CPU.  Test infrastructure only."""
import math

import torch


def _bilinear_zero_pad(img, px, py):
    """img [B,C,H,W]; px,py [B, ...] pixel coordinates (x = width axis).  Zeros outside."""
    B, C, H, W = img.shape
    x0 = torch.floor(px); y0 = torch.floor(py)
    fx = px - x0; fy = py - y0
    out = 0
    flat = img.reshape(B, C, H * W)
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xi = (x0 + dx).long(); yi = (y0 + dy).long()
            ok = ((xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)).to(img.dtype)
            idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).reshape(B, 1, -1).expand(B, C, -1)
            val = torch.gather(flat, 2, idx).reshape(B, C, *px.shape[1:])
            out = out + val * (wx * wy * ok).unsqueeze(1)
    return out


def extract_patches(imgs, landmarks, patch=8):
    """extract_patches_pytorch_gridsample (face_pre_pro/ViT_face.py:1615-1656) in closed form.

    imgs [B,C,S,S], landmarks [B,n,2] in pixels with landmarks[...,0] = x (width axis).
    patch_k[c,i,j] = bilinear(img[c], x = th_k[0] + (i - p/2) - 0.5, y = th_k[1] + (j - p/2) - 0.5):
    output ROW i walks along x (the patch comes out transposed) and there is a -0.5 px shift
    because the reference normalises by S/2 under align_corners=False (:1645-1646).
    Patch k goes to mosaic block (k // r, k % r), r = sqrt(n) (:1649-1654).  Differentiable
    with respect to ``landmarks`` and ``imgs``.
    """
    B, C, S, _ = imgs.shape
    n = landmarks.shape[1]
    r = int(math.isqrt(n))
    assert r * r == n
    off = torch.arange(patch, dtype=imgs.dtype) - patch / 2 - 0.5
    px = landmarks[:, :, 0, None, None] + off[None, None, :, None]          # [B,n,i,1]
    py = landmarks[:, :, 1, None, None] + off[None, None, None, :]          # [B,n,1,j]
    px, py = torch.broadcast_tensors(px, py)
    patches = _bilinear_zero_pad(imgs, px, py)                              # [B,C,n,i,j]
    patches = patches.reshape(B, C, r, r, patch, patch).permute(0, 1, 2, 4, 3, 5)
    return patches.reshape(B, C, r * patch, r * patch)


def landmarks_from_head(theta_raw, n):
    """Per-sample min-max scaling to [0, 111] pixels and reshape to [B, n, 2]
    (face_pre_pro/ViT_face.py:694-705, 1347-1355)."""
    tmax = theta_raw.max(dim=1, keepdim=True)[0]
    tmin = theta_raw.min(dim=1, keepdim=True)[0]
    return ((theta_raw - tmin) / (tmax - tmin) * 111).view(-1, n, 2)
