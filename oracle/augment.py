"""Oracle: the per-view image augmentations of DataAugmentation_LAFS (reference lafs_train.py:790-886) restated on uint8
numpy arrays.  Test infrastructure only.

The reference composes torchvision transforms over PIL images; torchvision is a thin wrapper there, the arithmetic is
Pillow's (third-party, not vendored by the reference; Pillow 12.2 is installed in the build container, so every function
below is PINNED against the real Pillow call by tests/test_augment_oracle.py):

  resized_crop        torchvision F.resized_crop = img.crop(box).resize((S, S), BICUBIC)        lafs_train.py:806,816,833
  hflip               img.transpose(FLIP_LEFT_RIGHT)                                              :807,817,834
  brightness/contrast/saturation   ImageEnhance.{Brightness,Contrast,Color}(img).enhance(f)      :794 (ColorJitter)
  hue                 HSV round trip with the uint8 hue channel shifted by uint8(f*255)            :794
  grayscale           img.convert('L') replicated to 3 channels (RandomGrayscale)                  :797
  gaussian_blur       img.filter(ImageFilter.GaussianBlur(radius))   (utils.GaussianBlur)          :812,822,838
  solarize            ImageOps.solarize(img)  (utils.Solarization)                                 :823
  normalize           ToTensor + Normalize(0.5, 0.5)                                               :799-802

All images here are uint8 arrays [H, W, 3] (RGB).  Integer arithmetic follows Pillow's C code (Resample.c, Blend.c, Convert.c,
BoxBlur.c) so the results are bit-identical, not merely close.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


# ------------------------------------------------------------------------------------------------ resize (bicubic)
def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size, out_size):
    """Pillow precompute_coeffs for a full-range box (Resample.c): per output index (xmin, integer coefficients)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(k)
        k = [v / ww if ww != 0.0 else v for v in k]
        ki = [int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS)) for v in k]
        out.append((xmin, ki))
    return out


def _clip8(v):
    return 0 if v < 0 else (255 if v > 255 else v)


def resize_bicubic(img, out_size):
    """img.resize((out_size, out_size), Image.BICUBIC) for an [H, W, 3] uint8 image: horizontal pass, then vertical pass, 8-bit
    intermediate, fixed-point coefficients (ImagingResample, Resample.c)."""
    H, W, C = img.shape
    cur = img.astype(np.int64)
    if W != out_size:
        co = resample_coeffs(W, out_size)
        tmp = np.empty((H, out_size, C), np.int64)
        for xx, (xmin, k) in enumerate(co):
            acc = np.full((H, C), 1 << (PRECISION_BITS - 1), np.int64)
            for t, kv in enumerate(k):
                acc += cur[:, xmin + t, :] * kv
            tmp[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
        cur = tmp
    if H != out_size:
        co = resample_coeffs(H, out_size)
        tmp = np.empty((out_size, cur.shape[1], C), np.int64)
        for yy, (ymin, k) in enumerate(co):
            acc = np.full((cur.shape[1], C), 1 << (PRECISION_BITS - 1), np.int64)
            for t, kv in enumerate(k):
                acc += cur[ymin + t, :, :] * kv
            tmp[yy, :, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
        cur = tmp
    return cur.astype(np.uint8)


def resized_crop(img, i, j, h, w, size):
    """torchvision F.resized_crop(img, top=i, left=j, height=h, width=w, size, BICUBIC)."""
    return resize_bicubic(img[i:i + h, j:j + w], size)


def hflip(img):
    return img[:, ::-1].copy()


# ------------------------------------------------------------------------------------------------ colour
def to_luma(img):
    """convert('L'): ITU-R 601-2 luma in 16.16 fixed point (Convert.c rgb2l / L24 macro)."""
    r, g, b = (img[..., k].astype(np.int64) for k in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def _blend(degenerate, image, alpha):
    """Image.blend(degenerate, image, alpha) (Blend.c): C float arithmetic, truncation toward zero, clipping only when
    extrapolating."""
    d = degenerate.astype(np.int32)
    v = image.astype(np.int32)
    a = np.float32(alpha)
    t = d.astype(np.float32) + a * (v - d).astype(np.float32)
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.int32).astype(np.uint8)                 # (UINT8) cast of an in-range float: truncation
    out = np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32)))
    return out.astype(np.uint8)


def adjust_brightness(img, f):
    return _blend(np.zeros_like(img), img, f)


def adjust_contrast(img, f):
    mean = int(to_luma(img).astype(np.float64).mean() + 0.5)        # ImageStat.Stat(L).mean[0] + 0.5 truncated
    return _blend(np.full_like(img, mean), img, f)


def adjust_saturation(img, f):
    return _blend(np.repeat(to_luma(img)[..., None], 3, axis=2), img, f)


def rgb_to_hsv(img):
    """Convert.c rgb2hsv_row."""
    r, g, b = (img[..., k].astype(np.int32) for k in range(3))
    maxc = np.maximum(r, np.maximum(g, b))
    minc = np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(np.float32)
    safe = np.where(cr == 0, np.float32(1), cr)
    s = cr / np.where(maxc == 0, 1, maxc).astype(np.float32)
    rc = (maxc - r).astype(np.float32) / safe
    gc = (maxc - g).astype(np.float32) / safe
    bc = (maxc - b).astype(np.float32) / safe
    # `float h`: "bc - gc" is float arithmetic, "2.0 + rc - bc" is evaluated in double (the literal) and rounded on assignment
    rc64, gc64, bc64 = rc.astype(np.float64), gc.astype(np.float64), bc.astype(np.float64)
    h = np.where(r == maxc, (bc - gc).astype(np.float64), np.where(g == maxc, 2.0 + rc64 - bc64, 4.0 + gc64 - rc64)).astype(np.float32)
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)      # the double result is rounded back to float
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    gray = maxc == minc
    uh = np.where(gray, 0, uh)
    us = np.where(gray, 0, us)
    return np.stack([uh, us, maxc], axis=-1).astype(np.uint8)


def hsv_to_rgb(hsv):
    """Convert.c hsv2rgb."""
    h, s, v = (hsv[..., k].astype(np.int32) for k in range(3))
    fh = h.astype(np.float64) * 6.0 / 255.0
    i = np.floor(fh).astype(np.int32)
    f = (fh - i).astype(np.float32)
    fs = s.astype(np.float32) / np.float32(255.0)
    vf = v.astype(np.float32)
    rnd = lambda x: np.clip(np.round(x.astype(np.float64)).astype(np.int32), 0, 255)   # C round(): halves away from zero (x >= 0 here)
    p = rnd(vf * (np.float32(1.0) - fs))
    q = rnd(vf * (np.float32(1.0) - fs * f))
    t = rnd(vf * (np.float32(1.0) - fs * (np.float32(1.0) - f)))
    k = i % 6
    r = np.choose(k, [v, q, p, p, t, v])
    g = np.choose(k, [t, v, v, q, p, p])
    b = np.choose(k, [p, p, t, v, v, q])
    gray = s == 0
    out = np.stack([np.where(gray, v, r), np.where(gray, v, g), np.where(gray, v, b)], axis=-1)
    return out.astype(np.uint8)


def hue_shift_u8(hue_factor):
    """np.uint8(hue_factor * 255): truncation toward zero, then wrap to 0..255 (torchvision F_pil.adjust_hue)."""
    return int(hue_factor * 255) & 0xFF


def adjust_hue(img, hue_factor):
    hsv = rgb_to_hsv(img)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + hue_shift_u8(hue_factor)).astype(np.uint8)        # uint8 wrap-around
    return hsv_to_rgb(hsv)


def to_grayscale3(img):
    return np.repeat(to_luma(img)[..., None], 3, axis=2)


def solarize(img, threshold=128):
    return np.where(img < threshold, img, 255 - img).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ Gaussian blur (box approximation)
def gaussian_box_radius(radius, passes=3):
    """BoxBlur.c _gaussian_blur_radius (C float arithmetic)."""
    f = np.float32
    sigma2 = f(radius) * f(radius) / f(passes)
    L = f(math.sqrt(f(12.0) * sigma2 + f(1.0)))
    l = f(math.floor((L - f(1.0)) / f(2.0)))
    a = (f(2) * l + f(1)) * (l * (l + f(1)) - f(3) * sigma2)
    a = a / (f(6) * (sigma2 - (l + f(1)) * (l + f(1))))
    return float(l + a)


def _box_blur_lines(arr, float_radius):
    """One ImagingHorizontalBoxBlur pass over the LAST-BUT-ONE axis being lines: arr [lines, n, C] uint8 -> same."""
    n = arr.shape[1]
    radius = int(float_radius)
    ww = int(np.uint32((1 << 24) / (np.float32(float_radius) * np.float32(2) + np.float32(1))))
    fw = ((1 << 24) - (radius * 2 + 1) * ww) // 2
    src = arr.astype(np.int64)
    idx = np.arange(n)
    acc = np.zeros_like(src)
    for k in range(-radius, radius + 1):
        acc += src[:, np.clip(idx + k, 0, n - 1), :]
    far = src[:, np.clip(idx - radius - 1, 0, n - 1), :] + src[:, np.clip(idx + radius + 1, 0, n - 1), :]
    bulk = acc * ww + far * fw
    return ((bulk + (1 << 23)) >> 24).astype(np.uint8)


def gaussian_blur(img, radius, passes=3):
    """img.filter(ImageFilter.GaussianBlur(radius)): `passes` horizontal box blurs, then `passes` vertical ones, each rounded
    to 8 bits (ImagingGaussianBlur -> ImagingBoxBlur, BoxBlur.c)."""
    fr = gaussian_box_radius(radius, passes)
    out = img
    if fr > 0:
        for _ in range(passes):
            out = _box_blur_lines(out, fr)
        t = out.transpose(1, 0, 2)
        for _ in range(passes):
            t = _box_blur_lines(t, fr)
        out = t.transpose(1, 0, 2)
    return np.ascontiguousarray(out)


# ------------------------------------------------------------------------------------------------ composition
def normalize(img):
    """ToTensor + Normalize((.5,.5,.5), (.5,.5,.5)): uint8 HWC -> float32 CHW in [-1, 1]."""
    x = img.astype(np.float32) / np.float32(255.0)
    return ((x - np.float32(0.5)) / np.float32(0.5)).transpose(2, 0, 1).copy()


JITTER_OPS = (adjust_brightness, adjust_contrast, adjust_saturation, adjust_hue)      # ColorJitter's fn_idx numbering


def color_pipeline(img, p):
    """The 'con2' branch of one view: ColorJitter (if p['jitter']) in p['order'], RandomGrayscale, GaussianBlur, Solarization,
    each switched by the sampled parameters (lafs_train.py:792-798, 809-840)."""
    out = img
    if p["jitter"]:
        for fn_id in p["order"]:
            out = JITTER_OPS[fn_id](out, p["factors"][fn_id])
    if p["gray"]:
        out = to_grayscale3(out)
    if p["blur_radius"] > 0:
        out = gaussian_blur(out, p["blur_radius"])
    if p["solarize"]:
        out = solarize(out)
    return out


def make_views(img, crops, size=112):
    """One source image -> the list [clean_0, aug_0, clean_1, aug_1, ...] of float32 CHW views (DataAugmentation_LAFS.__call__,
    lafs_train.py:842-856).  crops: list of parameter dicts (see augment_params.sample_view_params)."""
    views = []
    for p in crops:
        base = resized_crop(img, p["i"], p["j"], p["h"], p["w"], size)
        if p["flip"]:
            base = hflip(base)
        views.append(normalize(base))
        views.append(normalize(color_pipeline(base, p)))
    return views
