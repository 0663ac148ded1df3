"""Oracle: the RandAugment of the fine-tune loader (reference util/rand_aa_face.py, built by FaceDataset with
config 'rand-m1-mstd0.5-inc1' and hparams {'translate_const': 117}: face_pre_pro/dataloader_web.py:240-243,
train_largescale.py:506; applied per decoded sample at dataloader_web.py:342-346 / image_iter.py:324-329) restated on uint8 numpy
arrays.  Test infrastructure only.

Two halves:
  * `sample_record`: the reference's random decisions for ONE image, drawn from a `random.Random` and a `numpy RandomState` in
    exactly the order the reference draws them from the global `random` / `np.random` modules (RandAugment.__call__ :619-625,
    AugmentOp.__call__ :333-345, the level functions :168-246, _interpolation :40-45), so that equal seeds give equal records;
  * `apply_record`: the thirteen PIL operations of `_RAND_INCREASING_TRANSFORMS` (:560-576, the solarize entries are commented
    out in the reference's copy) on an [H, W, 3] uint8 image.  The arithmetic is Pillow's (third party, not vendored by the
    reference; Pillow 12.2 is installed in the build container and every function below is PINNED against the real Pillow
    call by tests/test_randaug_oracle.py): ImageOps.autocontrast / equalize / invert / posterize (LUTs from the per-band
    histogram), ImageEnhance.Color / Contrast / Brightness / Sharpness (Image.blend with a degenerate image: Blend.c float
    arithmetic; Sharpness' degenerate is ImageFilter.SMOOTH, Filter.c 3x3 float32), Image.transform(AFFINE) with BILINEAR or
    BICUBIC resampling and a (128, 128, 128) fill (Geometry.c: double arithmetic, truncation) for Rotate / ShearX / ShearY /
    TranslateXRel / TranslateYRel.
"""
import math

import numpy as np

from .augment import _blend, adjust_brightness, adjust_contrast, adjust_saturation

# op ids = positions in the reference's _RAND_INCREASING_TRANSFORMS (rand_aa_face.py:560-576)
OPS = ("AutoContrast", "Equalize", "Invert", "Rotate", "PosterizeIncreasing", "ColorIncreasing", "ContrastIncreasing",
       "BrightnessIncreasing", "SharpnessIncreasing", "ShearX", "ShearY", "TranslateXRel", "TranslateYRel")
AUTOCONTRAST, EQUALIZE, INVERT, ROTATE, POSTERIZE, COLOR, CONTRAST, BRIGHTNESS, SHARPNESS, SHEAR_X, SHEAR_Y, TRANS_X, TRANS_Y = range(13)
GEOMETRIC = (ROTATE, SHEAR_X, SHEAR_Y, TRANS_X, TRANS_Y)
BILINEAR, BICUBIC = 2, 3                    # PIL.Image.BILINEAR / BICUBIC
FILL = (128, 128, 128)                      # _FILL :25 (hparams carry no img_mean: dataloader_web.py:242)
MAX_LEVEL = 10.0


def parse_config(config_str):
    """rand_augment_transform's parsing of 'rand-m1-mstd0.5-inc1' (:643-669): (magnitude, num_layers, magnitude_std, increasing)."""
    import re
    magnitude, num_layers, mstd, inc = int(MAX_LEVEL), 2, 0.0, False
    parts = config_str.split("-")
    assert parts[0] == "rand"
    for c in parts[1:]:
        cs = re.split(r"(\d.*)", c)
        if len(cs) < 2:
            continue
        key, val = cs[:2]
        if key == "mstd":
            mstd = float(val)
        elif key == "inc":
            inc = bool(val)
        elif key == "m":
            magnitude = int(val)
        elif key == "n":
            num_layers = int(val)
        else:
            raise AssertionError("Unknown RandAugment config section")
    return magnitude, num_layers, mstd, inc


def sample_record(rnd, nprnd, magnitude=1, num_layers=2, mstd=0.5, prob=0.5, translate_pct=0.45):
    """The decisions of one RandAugment call: a list of num_layers entries (op, applied, arg, resample).
    `arg` is the Python value the reference hands to the PIL wrapper (degrees, factor, bits, shear factor, translate fraction)."""
    ops = nprnd.choice(len(OPS), num_layers)                  # np.random.choice(self.ops, n, replace=True, p=None)  :621-622
    out = []
    for op in (int(o) for o in ops):
        if prob < 1.0 and rnd.random() > prob:                # AugmentOp.__call__ :334
            out.append((op, False, 0.0, 0))
            continue
        m = float(magnitude)
        if mstd:
            m = rnd.uniform(0, m) if mstd == float("inf") else rnd.gauss(m, mstd)             # :338-342
        m = min(MAX_LEVEL, max(0, m))
        neg = lambda v: -v if rnd.random() > 0.5 else v       # _randomly_negate :162-165
        arg, resample = 0.0, 0
        if op == ROTATE:
            arg = neg((m / MAX_LEVEL) * 30.0)
        elif op == POSTERIZE:
            arg = 4 - int((m / MAX_LEVEL) * 4)                # _posterize_increasing_level_to_arg :217-221
        elif op in (COLOR, CONTRAST, BRIGHTNESS, SHARPNESS):
            arg = 1.0 + neg((m / MAX_LEVEL) * 0.9)            # _enhance_increasing_level_to_arg :178-183
        elif op in (SHEAR_X, SHEAR_Y):
            arg = neg((m / MAX_LEVEL) * 0.3)
        elif op in (TRANS_X, TRANS_Y):
            arg = neg((m / MAX_LEVEL) * translate_pct)        # _translate_rel_level_to_arg :200-205
        if op in GEOMETRIC:
            resample = rnd.choice((BILINEAR, BICUBIC))        # _check_args_tf -> _interpolation :40-51
        out.append((op, True, arg, resample))
    return out


# ------------------------------------------------------------------------------------------------ LUT operations (ImageOps)
def _histogram(img):
    return [np.bincount(img[..., c].ravel(), minlength=256).astype(np.int64) for c in range(3)]


def autocontrast_lut(h):
    """ImageOps.autocontrast, cutoff 0, one band: the identity when the band is constant."""
    nz = np.nonzero(h)[0]
    lo, hi = int(nz[0]), int(nz[-1])
    if hi <= lo:
        return list(range(256))
    scale = 255.0 / (hi - lo)
    offset = -lo * scale
    return [min(255, max(0, int(ix * scale + offset))) for ix in range(256)]


def equalize_lut(h):
    """ImageOps.equalize, one band."""
    histo = [int(f) for f in h if f]
    if len(histo) <= 1:
        return list(range(256))
    step = (sum(histo) - histo[-1]) // 255
    if not step:
        return list(range(256))
    n = step // 2
    lut = []
    for i in range(256):
        lut.append(n // step)
        n += int(h[i])
    return lut


def _apply_luts(img, luts):
    out = np.empty_like(img)
    for c in range(3):
        out[..., c] = np.asarray(luts[c], np.int64).clip(0, 255).astype(np.uint8)[img[..., c]]
    return out


def autocontrast(img):
    return _apply_luts(img, [autocontrast_lut(h) for h in _histogram(img)])


def equalize(img):
    return _apply_luts(img, [equalize_lut(h) for h in _histogram(img)])


def invert(img):
    return (255 - img).astype(np.uint8)


def posterize(img, bits):
    if bits >= 8:                                             # posterize :147-150
        return img.copy()
    mask = ~(2 ** (8 - bits) - 1) & 0xFF
    return (img & np.uint8(mask)).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ ImageFilter.SMOOTH / Sharpness
def smooth(img):
    """img.filter(ImageFilter.SMOOTH): 3x3 kernel (1 1 1 / 1 5 1 / 1 1 1) / 13 in float32 (Filter.c ImagingFilter3x3): per band
    ss = 0.5 + row(y+1) + row(y) + row(y-1), each row (p[x-1] k0 + p[x] k1) + p[x+1] k2, clip8 by truncation; the one-pixel
    border is copied."""
    f = np.float32
    k1, k5 = f(1.0) / f(13.0), f(5.0) / f(13.0)
    p = img.astype(np.float32)
    H, W, _ = img.shape

    def row(a, kl, kc, kr):
        return (a[:, :-2] * kl + a[:, 1:-1] * kc) + a[:, 2:] * kr

    ss = np.full((H - 2, W - 2, 3), f(0.5), np.float32)
    ss = ss + row(p[2:], k1, k1, k1)
    ss = ss + row(p[1:-1], k1, k5, k1)
    ss = ss + row(p[:-2], k1, k1, k1)
    out = img.copy()
    out[1:-1, 1:-1] = np.where(ss <= 0.0, 0, np.where(ss >= 255.0, 255, ss.astype(np.int32))).astype(np.uint8)
    return out


def adjust_sharpness(img, f):
    return _blend(smooth(img), img, f)


# ------------------------------------------------------------------------------------------------ Image.transform(AFFINE)
def rotate_matrix(degrees, w, h):
    """Image.rotate's matrix (Image.py; the same arithmetic as rand_aa_face.rotate's own branch :89-111)."""
    angle = -math.radians(degrees % 360.0)
    m = [round(math.cos(angle), 15), round(math.sin(angle), 15), 0.0, round(-math.sin(angle), 15), round(math.cos(angle), 15), 0.0]
    cx, cy = w / 2.0, h / 2.0
    m[2] = m[0] * -cx + m[1] * -cy + m[2] + cx
    m[5] = m[3] * -cx + m[4] * -cy + m[5] + cy
    return m


def affine_matrix(op, arg, w, h):
    if op == ROTATE:
        return rotate_matrix(arg, w, h)
    if op == SHEAR_X:
        return [1.0, float(arg), 0.0, 0.0, 1.0, 0.0]          # shear_x :54-56
    if op == SHEAR_Y:
        return [1.0, 0.0, 0.0, float(arg), 1.0, 0.0]
    if op == TRANS_X:
        return [1.0, 0.0, float(arg * w), 0.0, 1.0, 0.0]      # translate_x_rel :64-67
    if op == TRANS_Y:
        return [1.0, 0.0, 0.0, 0.0, 1.0, float(arg * h)]
    raise ValueError(op)


def rotate_special(degrees, w, h):
    """Image.rotate returns a copy / a transpose instead of resampling for 0, 180 and (square images) 90, 270 degrees."""
    a = degrees % 360.0
    if a == 0:
        return "copy"
    if a == 180:
        return "rot180"
    if a in (90, 270) and w == h:
        return "rot90" if a == 90 else "rot270"
    return None


def _floor(v):
    return np.where(v >= 0.0, v.astype(np.int64), np.floor(v).astype(np.int64))


def affine(img, m, resample, fill=FILL):
    """img.transform(img.size, AFFINE, m, resample, fillcolor=fill) for BILINEAR / BICUBIC (Geometry.c affine_transform +
    bilinear_filter32RGB / bicubic_filter32RGB): source position of output pixel (x, y) is the matrix applied to (x + 0.5, y + 0.5)
    in double; outside [0, W) x [0, H) the fill colour stays; neighbours are clamped to the image."""
    H, W, _ = img.shape
    ys, xs = np.mgrid[0:H, 0:W]
    xin = xs + 0.5
    yin = ys + 0.5
    xo = m[0] * xin + m[1] * yin + m[2]
    yo = m[3] * xin + m[4] * yin + m[5]
    inside = (xo >= 0.0) & (xo < W) & (yo >= 0.0) & (yo < H)
    xo = xo - 0.5
    yo = yo - 0.5
    x = _floor(xo)
    y = _floor(yo)
    dx = xo - x
    dy = yo - y
    src = img.astype(np.float64)
    cx = lambda v: np.clip(v, 0, W - 1)
    cy = lambda v: np.clip(v, 0, H - 1)
    out = np.empty_like(img)
    if resample == BILINEAR:
        for c in range(3):
            s = src[..., c]
            r0, r1 = cy(y), cy(y + 1)
            a0 = s[r0, cx(x)]; a1 = s[r0, cx(x + 1)]
            v1 = a0 + (a1 - a0) * dx
            b0 = s[r1, cx(x)]; b1 = s[r1, cx(x + 1)]
            v2 = b0 + (b1 - b0) * dx
            v = v1 + (v2 - v1) * dy
            out[..., c] = v.astype(np.int64).astype(np.uint8)                       # (UINT8) v: truncation
    elif resample == BICUBIC:
        def cubic(v1, v2, v3, v4, d):
            p1 = v2
            p2 = -v1 + v3
            p3 = 2 * (v1 - v2) + v3 - v4
            p4 = -v1 + v2 - v3 + v4
            return p1 + d * (p2 + d * (p3 + d * p4))
        x0 = x - 1
        y0 = y - 1
        for c in range(3):
            s = src[..., c]
            rows = []
            for k in range(4):
                r = cy(y0 + k)
                rows.append(cubic(s[r, cx(x0)], s[r, cx(x0 + 1)], s[r, cx(x0 + 2)], s[r, cx(x0 + 3)], dx))
            v = cubic(rows[0], rows[1], rows[2], rows[3], dy)
            out[..., c] = np.where(v <= 0.0, 0, np.where(v >= 255.0, 255, v.astype(np.int64))).astype(np.uint8)
    else:
        raise ValueError("resample must be BILINEAR or BICUBIC")
    for c in range(3):
        out[..., c] = np.where(inside, out[..., c], fill[c])
    return out


def geometric(img, op, arg, resample):
    H, W, _ = img.shape
    if op == ROTATE:
        sp = rotate_special(arg, W, H)
        if sp == "copy":
            return img.copy()
        if sp == "rot180":
            return img[::-1, ::-1].copy()
        if sp == "rot90":
            return np.rot90(img, 1).copy()                     # Image.ROTATE_90: counter-clockwise
        if sp == "rot270":
            return np.rot90(img, 3).copy()
    return affine(img, affine_matrix(op, arg, W, H), resample)


# ------------------------------------------------------------------------------------------------ the composed transform
def apply_op(img, op, arg, resample):
    if op == AUTOCONTRAST:
        return autocontrast(img)
    if op == EQUALIZE:
        return equalize(img)
    if op == INVERT:
        return invert(img)
    if op == POSTERIZE:
        return posterize(img, int(arg))
    if op == COLOR:
        return adjust_saturation(img, arg)
    if op == CONTRAST:
        return adjust_contrast(img, arg)
    if op == BRIGHTNESS:
        return adjust_brightness(img, arg)
    if op == SHARPNESS:
        return adjust_sharpness(img, arg)
    return geometric(img, op, arg, resample)


def apply_record(img, record):
    for op, applied, arg, resample in record:
        if applied:
            img = apply_op(img, op, arg, resample)
    return img
