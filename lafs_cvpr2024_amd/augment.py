"""Device-side view augmentation of the LAFS loader: DataAugmentation_LAFS (reference
lafs_train.py:790-886) -- RandomResizedCrop(112, bicubic) + flip, then per view ColorJitter / RandomGrayscale / GaussianBlur /
Solarization / normalisation -- from a uint8 batch resident in HBM to the 2*(2+n_local) float views the landmark front-end
consumes, in ONE kernel launch per batch.

Why: at 30 k crops/s per GPU the step consumes 3000 images/s, i.e. 60 000 PIL view constructions per second per GPU; the
reference's 6 loader workers deliver a few thousand.  The arithmetic is Pillow's (what torchvision's PIL transforms call),
restated with its fixed-point / float conventions so that the kernel reproduces Pillow bit for bit
(oracle/augment.py is pinned against Pillow, the kernel against the oracle).  Random PARAMETERS are drawn on the host with
the same distributions as torchvision's get_params (different random stream).
"""
import math

import numpy as np
import torch

from . import _lib
from .ops import _p, call

PRECISION_BITS = 32 - 8 - 2
OUT = 112
MAX_TAPS = 6
P_WORDS = 20                                        # int32 words per (image, crop) parameter record


# ------------------------------------------------------------------------------------------------ Pillow resampling table
def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def coeff_table(max_in=OUT, out_size=OUT):
    """Pillow's precompute_coeffs (Resample.c, bicubic, 8-bit fixed point) for every source extent 1..max_in -> out_size.
    int32 [max_in + 1, out_size, 2 + MAX_TAPS]: (first source index, number of taps, coefficients)."""
    tab = np.zeros((max_in + 1, out_size, 2 + MAX_TAPS), np.int32)
    for n in range(1, max_in + 1):
        scale = n / out_size
        filterscale = max(scale, 1.0)
        support = 2.0 * filterscale
        for xx in range(out_size):
            center = (xx + 0.5) * scale
            xmin = max(int(center - support + 0.5), 0)
            xmax = min(int(center + support + 0.5), n) - xmin
            k = [_bicubic((x + xmin - center + 0.5) / filterscale) for x in range(xmax)]
            ww = sum(k)
            k = [v / ww if ww != 0.0 else v for v in k]
            assert xmax <= MAX_TAPS
            tab[n, xx, 0], tab[n, xx, 1] = xmin, xmax
            for t, v in enumerate(k):
                tab[n, xx, 2 + t] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
    return tab


def gaussian_box(radius, passes=3):
    """(int radius, ww, fw) of Pillow's box approximation of GaussianBlur(radius) (BoxBlur.c), or (0, 0, 0) for no blur."""
    f = np.float32
    sigma2 = f(radius) * f(radius) / f(passes)
    L = f(math.sqrt(f(12.0) * sigma2 + f(1.0)))
    l = f(math.floor((L - f(1.0)) / f(2.0)))
    a = (f(2) * l + f(1)) * (l * (l + f(1)) - f(3) * sigma2)
    a = a / (f(6) * (sigma2 - (l + f(1)) * (l + f(1))))
    fr = float(l + a)
    if not fr > 0:
        return 0, 0, 0
    r = int(fr)
    ww = int(np.uint32((1 << 24) / (np.float32(fr) * np.float32(2) + np.float32(1))))
    fw = ((1 << 24) - (r * 2 + 1) * ww) // 2
    return r, ww, fw


# ------------------------------------------------------------------------------------------------ parameter sampling
def _resized_crop_params(rng, H, W, scale, ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """torchvision RandomResizedCrop.get_params."""
    area = H * W
    for _ in range(10):
        target = area * rng.uniform(scale[0], scale[1])
        aspect = math.exp(rng.uniform(math.log(ratio[0]), math.log(ratio[1])))
        w = int(round(math.sqrt(target * aspect)))
        h = int(round(math.sqrt(target / aspect)))
        if 0 < w <= W and 0 < h <= H:
            return rng.randint(0, H - h + 1), rng.randint(0, W - w + 1), h, w
    in_ratio = W / H
    if in_ratio < ratio[0]:
        w, h = W, int(round(W / ratio[0]))
    elif in_ratio > ratio[1]:
        h, w = H, int(round(H * ratio[1]))
    else:
        w, h = W, H
    return (H - h) // 2, (W - w) // 2, h, w


def sample_view_params(rng, n_local=8, size=OUT, crops_scale=(0.4, 1.0)):
    """Parameter dicts of the 2 + n_local crops of ONE image, in the loader's order (global 1, global 2, locals).
    Probabilities and ranges: lafs_train.py:792-840 (ColorJitter(0.4, 0.4, 0.2, 0.1) with p 0.8, RandomGrayscale 0.2,
    GaussianBlur p = 1.0 / 0.1 / 0.5 with radius U(0.1, 2), Solarization 0.2 on the second global view)."""
    out = []
    for k in range(2 + n_local):
        i, j, h, w = _resized_crop_params(rng, size, size, crops_scale)
        p = dict(i=i, j=j, h=h, w=w, flip=bool(rng.rand() < 0.5))
        p["jitter"] = bool(rng.rand() < 0.8)
        p["order"] = [int(v) for v in rng.permutation(4)]
        p["factors"] = [rng.uniform(0.6, 1.4), rng.uniform(0.6, 1.4), rng.uniform(0.8, 1.2), rng.uniform(-0.1, 0.1)]
        p["gray"] = bool(rng.rand() < 0.2)
        blur_p = 1.0 if k == 0 else (0.1 if k == 1 else 0.5)
        do_blur = bool(rng.rand() <= blur_p)
        radius = rng.uniform(0.1, 2.0)
        p["blur_radius"] = radius if do_blur else 0.0
        p["solarize"] = bool(k == 1 and rng.rand() < 0.2)
        out.append(p)
    return out


def pack_params(per_image):
    """list (images) of lists (crops) of parameter dicts -> int32 [B, n_crops, P_WORDS] record array for the kernel."""
    B, K = len(per_image), len(per_image[0])
    rec = np.zeros((B, K, P_WORDS), np.int32)
    fview = rec.view(np.float32)
    for b, crops in enumerate(per_image):
        for k, p in enumerate(crops):
            r = rec[b, k]
            r[0], r[1], r[2], r[3] = p["i"], p["j"], p["h"], p["w"]
            r[4] = (1 if p["flip"] else 0) | (2 if p["jitter"] else 0) | (4 if p["gray"] else 0) | (8 if p["solarize"] else 0)
            o = p["order"]
            r[5] = o[0] | (o[1] << 2) | (o[2] << 4) | (o[3] << 6)
            fview[b, k, 6], fview[b, k, 7], fview[b, k, 8] = p["factors"][0], p["factors"][1], p["factors"][2]
            r[9] = int(p["factors"][3] * 255) & 0xFF                        # np.uint8(hue_factor * 255) of torchvision
            br, ww, fw = gaussian_box(p["blur_radius"]) if p["blur_radius"] > 0 else (0, 0, 0)
            r[10], r[11], r[12] = br, ww, fw
            r[13] = 1 if (p["blur_radius"] > 0 and ww > 0) else 0
    return rec


def sample_packed(rng, B, n_local=8, size=OUT, crops_scale=(0.4, 1.0), return_dicts=False):
    """Vectorised sample_view_params + pack_params for a whole batch (same distributions, one numpy call per quantity instead
    of ~15 Python-level draws per crop: 640 crops per step would otherwise cost ~10 ms of host time)."""
    K = 2 + n_local
    n = B * K
    area = float(size * size)
    lr0, lr1 = math.log(3.0 / 4.0), math.log(4.0 / 3.0)
    target = area * rng.uniform(crops_scale[0], crops_scale[1], (n, 10))
    aspect = np.exp(rng.uniform(lr0, lr1, (n, 10)))
    w = np.rint(np.sqrt(target * aspect)).astype(np.int64)
    h = np.rint(np.sqrt(target / aspect)).astype(np.int64)
    ok = (w > 0) & (w <= size) & (h > 0) & (h <= size)
    first = np.where(ok.any(1), ok.argmax(1), 0)
    rows = np.arange(n)
    w, h = w[rows, first], h[rows, first]
    none = ~ok.any(1)                                      # fallback of get_params: the whole (square) image
    w[none], h[none] = size, size
    i = np.floor(rng.uniform(0, 1, n) * (size - h + 1)).astype(np.int64)
    j = np.floor(rng.uniform(0, 1, n) * (size - w + 1)).astype(np.int64)
    i[none], j[none] = 0, 0
    kidx = np.tile(np.arange(K), B)
    flip = rng.rand(n) < 0.5
    jitter = rng.rand(n) < 0.8
    order = np.argsort(rng.rand(n, 4), axis=1)             # uniform random permutations
    fb, fc, fs = rng.uniform(0.6, 1.4, n), rng.uniform(0.6, 1.4, n), rng.uniform(0.8, 1.2, n)
    fh = rng.uniform(-0.1, 0.1, n)
    gray = rng.rand(n) < 0.2
    blur_p = np.where(kidx == 0, 1.0, np.where(kidx == 1, 0.1, 0.5))
    do_blur = rng.rand(n) <= blur_p
    radius = rng.uniform(0.1, 2.0, n)
    sol = (kidx == 1) & (rng.rand(n) < 0.2)
    rec = np.zeros((n, P_WORDS), np.int32)
    fv = rec.view(np.float32)
    rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3] = i, j, h, w
    rec[:, 4] = flip * 1 + jitter * 2 + gray * 4 + sol * 8
    rec[:, 5] = order[:, 0] | (order[:, 1] << 2) | (order[:, 2] << 4) | (order[:, 3] << 6)
    fv[:, 6], fv[:, 7], fv[:, 8] = fb, fc, fs
    rec[:, 9] = (fh * 255).astype(np.int64) & 0xFF       # truncation toward zero, then the uint8 wrap
    # the box parameters of Pillow's GaussianBlur (gaussian_box above, float32 arithmetic), vectorised
    f = np.float32
    sigma2 = radius.astype(f) * radius.astype(f) / f(3)
    L = np.sqrt((f(12.0) * sigma2 + f(1.0)).astype(np.float64)).astype(f)
    l = np.floor((L - f(1.0)) / f(2.0)).astype(f)
    a = (f(2) * l + f(1)) * (l * (l + f(1)) - f(3) * sigma2)
    a = a / (f(6) * (sigma2 - (l + f(1)) * (l + f(1))))
    fr = (l + a).astype(f)
    br = fr.astype(np.int64)
    ww = (np.float32(1 << 24) / (fr * f(2) + f(1))).astype(np.int64)
    fw = ((1 << 24) - (br * 2 + 1) * ww) // 2
    on = do_blur & (fr > 0)
    rec[:, 10], rec[:, 11], rec[:, 12], rec[:, 13] = np.where(on, br, 0), np.where(on, ww, 0), np.where(on, fw, 0), on
    rec = rec.reshape(B, K, P_WORDS)
    if not return_dicts:
        return rec
    dicts = [[dict(i=int(i[t]), j=int(j[t]), h=int(h[t]), w=int(w[t]), flip=bool(flip[t]), jitter=bool(jitter[t]),
                   order=[int(v) for v in order[t]], factors=[float(f(fb[t])), float(f(fc[t])), float(f(fs[t])), float(fh[t])],
                   gray=bool(gray[t]), blur_radius=float(radius[t]) if do_blur[t] else 0.0, solarize=bool(sol[t]))
              for t in range(bi * K, (bi + 1) * K)] for bi in range(B)]
    return rec, dicts


class DeviceAugmenter:
    """uint8 NCHW batch [B,3,112,112] on the device -> views f32 [2*(2+n_local), B, 3, 112, 112] (clean / augmented pairs in the
    loader's order), one launch."""

    def __init__(self, batch_size, n_local=8, device=None, seed=0):
        self.device = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
        self.B, self.K = batch_size, 2 + n_local
        self.rng = np.random.RandomState(seed)
        self.table = torch.from_numpy(coeff_table()).to(self.device)
        self.views = torch.empty(2 * self.K, batch_size, 3, OUT, OUT, device=self.device, dtype=torch.float32)
        self.params_dev = torch.empty(batch_size, self.K, P_WORDS, device=self.device, dtype=torch.int32)
        from .utils import PinnedRing
        self.params_ring = PinnedRing((batch_size, self.K, P_WORDS), torch.int32)      # see PinnedRing: the host runs ahead

    def sample(self):
        return [sample_view_params(self.rng, self.K - 2) for _ in range(self.B)]

    def __call__(self, images_u8, params=None):
        """images_u8: uint8 [B,3,112,112] device tensor.  params: optional explicit per-image parameter lists (tests)."""
        if images_u8.dtype != torch.uint8 or not images_u8.is_cuda or tuple(images_u8.shape) != (self.B, 3, OUT, OUT):
            raise _lib.LafsHipError("DeviceAugmenter expects a uint8 device tensor [B,3,112,112]")
        rec = pack_params(params) if params is not None else sample_packed(self.rng, self.B, self.K - 2)
        self.params_ring.upload(self.params_dev, lambda buf: buf.copy_(torch.from_numpy(rec)))
        call("lafs_augment_views", _p(images_u8.contiguous()), _p(self.params_dev), _p(self.table), self.B, self.K, _p(self.views))
        return self.views
