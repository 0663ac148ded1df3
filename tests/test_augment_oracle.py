"""CPU: the augmentation oracle (oracle/augment.py) against Pillow itself -- the arithmetic torchvision's PIL transforms
delegate to (DataAugmentation_LAFS, reference lafs_train.py:790-886).  Bit-exact unless stated."""
import numpy as np
import pytest

PIL = pytest.importorskip("PIL")
from PIL import Image, ImageEnhance, ImageFilter, ImageOps  # noqa: E402

from oracle import augment as A  # noqa: E402


def _img(seed, h=112, w=112, smooth=False):
    rng = np.random.RandomState(seed)
    if smooth:                                    # low-frequency content + noise: closer to a photograph than white noise
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([127 + 100 * np.sin(xx / 9.0 + c) * np.cos(yy / 7.0 - c) for c in range(3)], -1)
        return np.clip(base + rng.randn(h, w, 3) * 12, 0, 255).astype(np.uint8)
    return rng.randint(0, 256, (h, w, 3)).astype(np.uint8)


def _diff(a, b):
    return np.abs(a.astype(np.int32) - b.astype(np.int32))


@pytest.mark.parametrize("box", [(0, 0, 112, 112), (5, 9, 71, 83), (20, 3, 90, 64), (0, 30, 112, 76), (17, 17, 40, 45)])
def test_resized_crop_bicubic(box):
    i, j, h, w = box
    img = _img(1, smooth=True)
    ref = np.asarray(Image.fromarray(img).crop((j, i, j + w, i + h)).resize((112, 112), Image.BICUBIC))
    got = A.resized_crop(img, i, j, h, w, 112)
    assert _diff(got, ref).max() == 0
    ref48 = np.asarray(Image.fromarray(img).crop((j, i, j + w, i + h)).resize((48, 48), Image.BICUBIC))   # a down-scaling case
    assert _diff(A.resized_crop(img, i, j, h, w, 48), ref48).max() == 0


def test_luma_grayscale_solarize_flip():
    img = _img(2)
    pil = Image.fromarray(img)
    assert _diff(A.to_luma(img), np.asarray(pil.convert("L"))).max() == 0
    assert _diff(A.to_grayscale3(img), np.asarray(Image.merge("RGB", [pil.convert("L")] * 3))).max() == 0
    assert _diff(A.solarize(img), np.asarray(ImageOps.solarize(pil, 128))).max() == 0
    assert _diff(A.hflip(img), np.asarray(pil.transpose(Image.FLIP_LEFT_RIGHT))).max() == 0


@pytest.mark.parametrize("f", [0.6, 0.83, 1.0, 1.17, 1.4])
def test_brightness_contrast_saturation(f):
    for seed, smooth in ((3, False), (4, True)):
        img = _img(seed, smooth=smooth)
        pil = Image.fromarray(img)
        assert _diff(A.adjust_brightness(img, f), np.asarray(ImageEnhance.Brightness(pil).enhance(f))).max() == 0
        assert _diff(A.adjust_contrast(img, f), np.asarray(ImageEnhance.Contrast(pil).enhance(f))).max() == 0
        assert _diff(A.adjust_saturation(img, 0.8 + (f - 0.6) * 0.5), np.asarray(ImageEnhance.Color(pil).enhance(0.8 + (f - 0.6) * 0.5))).max() == 0


def _pil_hue(pil, hue_factor):
    """torchvision F_pil.adjust_hue."""
    h, s, v = pil.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    with np.errstate(over="ignore"):
        np_h += np.array(int(hue_factor * 255)).astype(np.uint8)
    return Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")


@pytest.mark.parametrize("hf", [-0.1, -0.037, 0.0, 0.05, 0.1])
def test_hsv_round_trip_and_hue(hf):
    img = _img(5)
    pil = Image.fromarray(img)
    assert _diff(A.rgb_to_hsv(img), np.asarray(pil.convert("HSV"))).max() == 0
    hsv = np.asarray(pil.convert("HSV"))
    assert _diff(A.hsv_to_rgb(hsv), np.asarray(Image.fromarray(hsv, "HSV").convert("RGB"))).max() == 0
    assert _diff(A.adjust_hue(img, hf), np.asarray(_pil_hue(pil, hf))).max() == 0


@pytest.mark.parametrize("radius", [0.1, 0.45, 1.0, 1.37, 2.0])
def test_gaussian_blur(radius):
    for seed, smooth in ((6, False), (7, True)):
        img = _img(seed, smooth=smooth)
        ref = np.asarray(Image.fromarray(img).filter(ImageFilter.GaussianBlur(radius)))
        assert _diff(A.gaussian_blur(img, radius), ref).max() == 0


def test_full_view_pipeline_against_pil_composition():
    """A whole (clean, augmented) pair as DataAugmentation_LAFS builds it, composed from the Pillow calls."""
    img = _img(8, smooth=True)
    p = dict(i=7, j=11, h=88, w=79, flip=True, jitter=True, order=[2, 0, 3, 1], factors=[1.21, 0.77, 0.93, -0.06], gray=False,
             blur_radius=1.3, solarize=True)
    clean, aug = A.make_views(img, [p])
    pil = Image.fromarray(img).crop((11, 7, 11 + 79, 7 + 88)).resize((112, 112), Image.BICUBIC).transpose(Image.FLIP_LEFT_RIGHT)
    ref_clean = (np.asarray(pil).astype(np.float32) / 255 - 0.5) / 0.5
    q = ImageEnhance.Color(pil).enhance(0.93)
    q = ImageEnhance.Brightness(q).enhance(1.21)
    q = _pil_hue(q, -0.06)
    q = ImageEnhance.Contrast(q).enhance(0.77)
    q = q.filter(ImageFilter.GaussianBlur(1.3))
    q = ImageOps.solarize(q, 128)
    ref_aug = (np.asarray(q).astype(np.float32) / 255 - 0.5) / 0.5
    np.testing.assert_allclose(clean, ref_clean.transpose(2, 0, 1), atol=1e-6)
    np.testing.assert_allclose(aug, ref_aug.transpose(2, 0, 1), atol=1e-6)
