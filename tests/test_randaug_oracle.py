"""CPU: the RandAugment oracle (oracle/randaug.py) against Pillow itself, against the reference's transform (golden F19:
util/rand_aa_face.py run here with fixed seeds, tools/make_golden.py) and against the product's host-side sampler."""
import os
import random

import numpy as np
import pytest

PIL = pytest.importorskip("PIL")
from PIL import Image, ImageEnhance, ImageFilter, ImageOps  # noqa: E402

from oracle import randaug as R  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "f19_randaugment.npz")
FILL = (128, 128, 128)


def _imgs():
    rng = np.random.RandomState(0)
    yy, xx = np.mgrid[0:112, 0:112]
    out = [rng.randint(0, 256, (112, 112, 3)), rng.randint(60, 200, (112, 112, 3)),
           np.stack([(xx * 2) % 256, (yy * 2 + xx) % 256, (xx * yy) % 256], -1), np.full((112, 112, 3), 77),
           (rng.rand(112, 112, 3) ** 3) * 255, rng.randint(0, 256, (96, 112, 3)), rng.randint(0, 256, (40, 25, 3))]
    return [np.asarray(a).astype(np.uint8) for a in out]


def _same(got, ref):
    return np.array_equal(got, np.asarray(ref))


@pytest.mark.parametrize("k", range(7))
def test_lut_and_enhance_operations_are_pillows(k):
    a = _imgs()[k]
    im = Image.fromarray(a)
    assert _same(R.autocontrast(a), ImageOps.autocontrast(im))
    assert _same(R.equalize(a), ImageOps.equalize(im))
    assert _same(R.invert(a), ImageOps.invert(im))
    for bits in range(1, 8):
        assert _same(R.posterize(a, bits), ImageOps.posterize(im, bits))
    assert _same(R.posterize(a, 8), im)
    assert _same(R.smooth(a), im.filter(ImageFilter.SMOOTH))
    for f in (0.1, 0.55, 0.91, 1.0, 1.09, 1.45, 1.9):
        assert _same(R.adjust_saturation(a, f), ImageEnhance.Color(im).enhance(f))
        assert _same(R.adjust_contrast(a, f), ImageEnhance.Contrast(im).enhance(f))
        assert _same(R.adjust_brightness(a, f), ImageEnhance.Brightness(im).enhance(f))
        assert _same(R.adjust_sharpness(a, f), ImageEnhance.Sharpness(im).enhance(f))


@pytest.mark.parametrize("k", [0, 2, 5, 6])
@pytest.mark.parametrize("rs", [Image.BILINEAR, Image.BICUBIC])
def test_geometric_operations_are_pillows(k, rs):
    a = _imgs()[k]
    im = Image.fromarray(a)
    for deg in (3.0, -2.7, 29.5, -30.0, 0.0, 90.0, 180.0, 270.0, 45.0, 0.3, -90.0, 359.2):
        assert _same(R.geometric(a, R.ROTATE, deg, rs), im.rotate(deg, resample=rs, fillcolor=FILL)), deg
    for s in (0.03, -0.03, 0.3, -0.21):
        assert _same(R.geometric(a, R.SHEAR_X, s, rs), im.transform(im.size, Image.AFFINE, (1, s, 0, 0, 1, 0), resample=rs, fillcolor=FILL))
        assert _same(R.geometric(a, R.SHEAR_Y, s, rs), im.transform(im.size, Image.AFFINE, (1, 0, 0, s, 1, 0), resample=rs, fillcolor=FILL))
    for t in (0.045, -0.045, 0.45, -0.33, 0.0131):
        assert _same(R.geometric(a, R.TRANS_X, t, rs),
                     im.transform(im.size, Image.AFFINE, (1, 0, t * im.size[0], 0, 1, 0), resample=rs, fillcolor=FILL))
        assert _same(R.geometric(a, R.TRANS_Y, t, rs),
                     im.transform(im.size, Image.AFFINE, (1, 0, 0, 0, 1, t * im.size[1]), resample=rs, fillcolor=FILL))


@pytest.mark.parametrize("tag", ["m1", "m9n3"])
def test_f19_reference_transform_with_the_reference_random_stream(tag):
    """The reference's rand_augment_transform (both configurations), image i under random.seed(s+i); np.random.seed(s+i): the oracle
    draws the same decisions from generators in the same state and produces the same bytes."""
    g = np.load(GOLD)
    m, n, sd, inc = R.parse_config(str(g["cfg_" + tag]))
    assert inc
    used = set()
    for i, a in enumerate(g["images"]):
        s = int(g["seed0"]) + i
        rec = R.sample_record(random.Random(s), np.random.RandomState(s), m, n, sd)
        used |= {r[0] for r in rec if r[1]}
        assert np.array_equal(R.apply_record(a, rec), g["out_" + tag][i]), (i, rec)
    if tag == "m9n3":
        assert used == set(range(13))                      # every operation of the list occurs in the fixture


def test_product_sampler_draws_the_oracle_records():
    """lafs_cvpr2024_amd.randaug.DeviceRandAugment.sample (host logic of the product) against the oracle's restatement of the
    reference's decision order, including Image.rotate's transpose shortcuts and the Posterize mask."""
    from lafs_cvpr2024_amd import randaug as P
    for cfg in ("rand-m1-mstd0.5-inc1", "rand-m9-n3-mstd0.5-inc1", "rand-m10-n2-inc1"):
        m, n, sd, _ = R.parse_config(cfg)
        aug = P.DeviceRandAugment(cfg, {"translate_const": 117})
        for s in range(60):
            aug.seed(s)
            got = aug.sample(3, 112, 112)
            rnd, nprnd = random.Random(s), np.random.RandomState(s)
            for b in range(3):
                rec = R.sample_record(rnd, nprnd, m, n, sd)
                for l, (op, applied, arg, resample) in enumerate(rec):
                    r = got[b, l]
                    if not applied or (op == R.POSTERIZE and arg >= 8):
                        assert r["op"] == -1
                        continue
                    if op == R.ROTATE and R.rotate_special(arg, 112, 112):
                        assert r["op"] == {"copy": -1, "rot180": P.ROT180, "rot90": P.ROT90, "rot270": P.ROT270}[R.rotate_special(arg, 112, 112)]
                        continue
                    assert r["op"] == op
                    if op in R.GEOMETRIC:
                        assert r["resample"] == resample
                        assert list(r["m"]) == R.affine_matrix(op, arg, 112, 112)
                    elif op == R.POSTERIZE:
                        assert r["iarg"] == (~(2 ** (8 - arg) - 1) & 0xFF)
                    elif op in (R.COLOR, R.CONTRAST, R.BRIGHTNESS, R.SHARPNESS):
                        assert r["farg"] == np.float32(arg)
    with pytest.raises(NotImplementedError):
        P.DeviceRandAugment("rand-m1-mstd0.5")              # the non-increasing list is not what the reference's loaders build
