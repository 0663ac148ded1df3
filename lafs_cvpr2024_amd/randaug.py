"""RandAugment of the fine-tune loader on the device (reference util/rand_aa_face.py:606-672 as FaceDataset builds it --
face_pre_pro/dataloader_web.py:240-243 with config 'rand-m1-mstd0.5-inc1' (train_largescale.py:506) and hparams
{'translate_const': 117} -- applied to every decoded sample at dataloader_web.py:342-346 / image_iter.py:324-329).

The reference runs thirteen PIL operations on one CPU worker per sample.  Here the host only draws the DECISIONS -- from a
`random.Random` and a numpy `RandomState`, in exactly the order the reference draws them from the global `random` / `np.random`
modules (np.random.choice of the layer ops, then per op: random() against prob, gauss(magnitude, std), random() for the sign,
choice of the resampling filter) -- and packs them into one 64-byte record per image and layer; lafs_randaug_apply (csrc/randaug.hip)
applies them with one workgroup per image, the picture resident in LDS, bit-identical to the PIL calls."""
import math
import random as _random
import re

import numpy as np
import torch

from . import _lib
from .ops import _p, call

OPS = ("AutoContrast", "Equalize", "Invert", "Rotate", "PosterizeIncreasing", "ColorIncreasing", "ContrastIncreasing",
       "BrightnessIncreasing", "SharpnessIncreasing", "ShearX", "ShearY", "TranslateXRel", "TranslateYRel")     # :560-576
(AUTOCONTRAST, EQUALIZE, INVERT, ROTATE, POSTERIZE, COLOR, CONTRAST, BRIGHTNESS, SHARPNESS, SHEAR_X, SHEAR_Y, TRANS_X, TRANS_Y,
 ROT180, ROT90, ROT270) = range(16)
BILINEAR, BICUBIC = 2, 3
_MAX_LEVEL = 10.0
RECORD = np.dtype([("op", "<i4"), ("resample", "<i4"), ("iarg", "<i4"), ("farg", "<f4"), ("m", "<f8", (6,))])
assert RECORD.itemsize == 64


def parse_config(config_str):
    """'rand-m1-mstd0.5-inc1' -> (magnitude, num_layers, magnitude_std)   (rand_augment_transform :643-669; only the increasing op
    list without choice weights is built: what the reference's loaders ask for)."""
    magnitude, num_layers, mstd, inc = int(_MAX_LEVEL), 2, 0.0, False
    parts = config_str.split("-")
    if parts[0] != "rand":
        raise ValueError("RandAugment config must start with 'rand'")
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
        elif key == "w":
            raise NotImplementedError("choice weights ('w') are not used by the reference's loaders")
        else:
            raise ValueError("Unknown RandAugment config section")
    if not inc:
        raise NotImplementedError("only the 'inc1' op list is built (the reference's loaders use it)")
    return magnitude, num_layers, mstd


def _rotate_matrix(degrees, w, h):
    """Image.rotate's matrix (the arithmetic rand_aa_face.rotate spells out at :89-111)."""
    angle = -math.radians(degrees)
    m = [round(math.cos(angle), 15), round(math.sin(angle), 15), 0.0, round(-math.sin(angle), 15), round(math.cos(angle), 15), 0.0]
    cx, cy = w / 2.0, h / 2.0
    m[2] = m[0] * -cx + m[1] * -cy + m[2] + cx
    m[5] = m[3] * -cx + m[4] * -cy + m[5] + cy
    return m


class DeviceRandAugment:
    """aug = DeviceRandAugment('rand-m1-mstd0.5-inc1', {'translate_const': 117}); out = aug(images_u8)"""

    def __init__(self, config_str="rand-m1-mstd0.5-inc1", hparams=None, seed=None, prob=0.5):
        self.magnitude, self.num_layers, self.mstd = parse_config(config_str)
        hparams = dict(hparams or {})
        self.translate_pct = hparams.get("translate_pct", 0.45)
        if tuple(hparams.get("img_mean", (128, 128, 128))) != (128, 128, 128):
            raise NotImplementedError("fill colour other than the reference's (128, 128, 128)")
        self.prob = prob
        self.rnd = _random.Random(seed)
        self.nprnd = np.random.RandomState(seed)

    def seed(self, seed):
        """The state random.seed(seed); np.random.seed(seed) leaves the reference's generators in."""
        self.rnd.seed(seed)
        self.nprnd.seed(seed)

    # ---- the reference's decisions for one image, in its own random order
    def sample_one(self, H, W, out):
        rnd = self.rnd
        ops = self.nprnd.choice(len(OPS), self.num_layers)                       # RandAugment.__call__ :621-622
        for l, op in enumerate(int(o) for o in ops):
            rec = out[l]
            rec["op"] = -1
            if self.prob < 1.0 and rnd.random() > self.prob:                     # AugmentOp.__call__ :334
                continue
            m = float(self.magnitude)
            if self.mstd:
                m = rnd.uniform(0, m) if self.mstd == float("inf") else rnd.gauss(m, self.mstd)
            m = min(_MAX_LEVEL, max(0, m))
            level = m / _MAX_LEVEL
            neg = lambda v: -v if rnd.random() > 0.5 else v                      # _randomly_negate :162-165
            if op in (AUTOCONTRAST, EQUALIZE, INVERT):
                rec["op"] = op
            elif op == POSTERIZE:
                bits = 4 - int(level * 4)                                        # :217-221
                if bits < 8:
                    rec["op"] = op
                    rec["iarg"] = ~(2 ** (8 - bits) - 1) & 0xFF
            elif op in (COLOR, CONTRAST, BRIGHTNESS, SHARPNESS):
                rec["op"] = op
                rec["farg"] = 1.0 + neg(level * 0.9)                             # :178-183
            else:
                if op == ROTATE:
                    deg = neg(level * 30.0)
                elif op in (SHEAR_X, SHEAR_Y):
                    f = neg(level * 0.3)
                else:
                    f = neg(level * self.translate_pct)
                rec["resample"] = rnd.choice((BILINEAR, BICUBIC))                # _check_args_tf -> _interpolation :40-51
                rec["op"] = op
                if op == ROTATE:
                    a = deg % 360.0                                              # Image.rotate: transposes instead of resampling
                    if a == 0:
                        rec["op"] = -1
                    elif a == 180:
                        rec["op"] = ROT180
                    elif a in (90, 270) and W == H:
                        rec["op"] = ROT90 if a == 90 else ROT270
                    else:
                        rec["m"] = _rotate_matrix(a, W, H)
                elif op == SHEAR_X:
                    rec["m"] = (1.0, f, 0.0, 0.0, 1.0, 0.0)
                elif op == SHEAR_Y:
                    rec["m"] = (1.0, 0.0, 0.0, f, 1.0, 0.0)
                elif op == TRANS_X:
                    rec["m"] = (1.0, 0.0, f * W, 0.0, 1.0, 0.0)
                else:
                    rec["m"] = (1.0, 0.0, 0.0, 0.0, 1.0, f * H)

    def sample(self, B, H=112, W=112):
        """Records [B, num_layers] for B images, drawn image after image like B consecutive calls of the reference's transform."""
        recs = np.zeros((B, self.num_layers), RECORD)
        for b in range(B):
            self.sample_one(H, W, recs[b])
        return recs

    def __call__(self, images, records=None, out=None):
        """images: uint8 CUDA tensor [B, H, W, 3] or [B, 3, H, W] (told apart by the position of the 3)."""
        if images.dtype != torch.uint8 or not images.is_cuda or images.dim() != 4:
            raise ValueError("expected a uint8 CUDA tensor [B,H,W,3] or [B,3,H,W]")
        chw = images.shape[1] == 3 and images.shape[3] != 3
        if not chw and images.shape[3] != 3:
            raise ValueError("expected 3 colour channels")
        B = images.shape[0]
        H, W = (images.shape[2], images.shape[3]) if chw else (images.shape[1], images.shape[2])
        if records is None:
            records = self.sample(B, H, W)
        if records.dtype != RECORD or records.shape[0] != B:
            raise ValueError("records must come from sample()")
        images = images.contiguous()
        layers = int(records.shape[1])
        rec = torch.from_numpy(np.ascontiguousarray(records).view(np.uint8).reshape(B, layers * RECORD.itemsize)).to(images.device)
        if out is None:
            out = torch.empty_like(images)
        call("lafs_randaug_apply", _p(images), _p(out), _p(rec), B, H, W, layers, 1 if chw else 0)
        return out
