"""GPU: the one-launch view augmentation (lafs_augment_views) against the Pillow-pinned oracle, bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
pytestmark = pytest.mark.gpu


def _images(B, seed):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:112, 0:112]
    out = []
    for b in range(B):
        base = np.stack([127 + 90 * np.sin(xx / (6.0 + b) + c) * np.cos(yy / (5.0 + c) - b) for c in range(3)], -1)
        out.append(np.clip(base + rng.randn(112, 112, 3) * 20, 0, 255).astype(np.uint8))
    return out


def test_device_augmenter_matches_oracle_bit_for_bit():
    from lafs_cvpr2024_amd import augment as aug
    from oracle import augment as A
    B, nl = 3, 3
    imgs = _images(B, 0)
    da = aug.DeviceAugmenter(B, n_local=nl, device="cuda", seed=5)
    params = da.sample()
    # make sure every branch is exercised at least once
    params[0][0].update(i=0, j=0, h=112, w=112, flip=False, jitter=True, order=[3, 1, 0, 2], gray=False, blur_radius=2.0, solarize=True)
    params[0][1].update(i=3, j=0, h=100, w=112, flip=True, jitter=True, order=[0, 1, 2, 3], gray=True, blur_radius=0.1, solarize=False)
    params[1][2].update(i=0, j=7, h=112, w=90, flip=True, jitter=False, gray=False, blur_radius=0.0, solarize=False)
    params[2][4].update(jitter=True, order=[1, 3, 2, 0], factors=[0.61, 1.39, 0.8, -0.1], gray=False, blur_radius=1.21, solarize=False)
    x = torch.from_numpy(np.stack(imgs).transpose(0, 3, 1, 2).copy()).cuda()
    views = da(x, params).cpu().numpy()
    assert views.shape == (2 * (2 + nl), B, 3, 112, 112)
    worst = 0.0
    for b in range(B):
        ref = A.make_views(imgs[b], params[b])
        for v, r in enumerate(ref):
            d = np.abs(views[v, b] - r).max()
            worst = max(worst, float(d))
            assert d == 0.0, (b, v, float(d) * 127.5, params[b][v // 2])
    assert worst == 0.0


def test_device_augmenter_sampled_parameters_bit_exact():
    """A full loader batch with SAMPLED parameters (8 images x 10 crops = 160 views): still zero differing bytes."""
    from lafs_cvpr2024_amd import augment as aug
    from oracle import augment as A
    B, nl = 8, 8
    rng = np.random.RandomState(3)
    imgs = _images(B - 2, 1) + [rng.randint(0, 256, (112, 112, 3)).astype(np.uint8), np.full((112, 112, 3), 255, np.uint8)]
    da = aug.DeviceAugmenter(B, n_local=nl, device="cuda", seed=11)
    params = da.sample()
    x = torch.from_numpy(np.stack(imgs).transpose(0, 3, 1, 2).copy()).cuda()
    views = da(x, params).cpu().numpy()
    bad = 0
    for b in range(B):
        for v, r in enumerate(A.make_views(imgs[b], params[b])):
            bad += int((views[v, b] != r).sum())
    assert bad == 0


def test_vectorised_sampler_records_bit_exact_and_distributed():
    """sample_packed (the per-step host path): its records drive the kernel to the same bytes as the oracle fed with the decoded
    parameters, and the draws follow the loader's distributions."""
    from lafs_cvpr2024_amd import augment as aug
    from lafs_cvpr2024_amd.ops import _p, call
    from oracle import augment as A
    B, nl = 4, 8
    rng = np.random.RandomState(21)
    rec, dicts = aug.sample_packed(rng, B, nl, return_dicts=True)
    imgs = _images(B, 2)
    da = aug.DeviceAugmenter(B, n_local=nl, device="cuda", seed=0)
    x = torch.from_numpy(np.stack(imgs).transpose(0, 3, 1, 2).copy()).cuda()
    da.params_dev.copy_(torch.from_numpy(rec))
    call("lafs_augment_views", _p(x), _p(da.params_dev), _p(da.table), B, 2 + nl, _p(da.views))
    views = da.views.cpu().numpy()
    bad = sum(int((views[v, b] != r).sum()) for b in range(B) for v, r in enumerate(A.make_views(imgs[b], dicts[b])))
    assert bad == 0
    r = np.concatenate([aug.sample_packed(rng, 64, 8).reshape(-1, aug.P_WORDS) for _ in range(10)])
    assert ((r[:, 0] + r[:, 2]) <= 112).all() and ((r[:, 1] + r[:, 3]) <= 112).all() and (r[:, 2:4] > 0).all()
    area = r[:, 2] * r[:, 3] / 112.0 ** 2
    assert 0.38 < area.min() and 0.6 < area.mean() < 0.8
    assert abs((r[:, 4] & 1).mean() - 0.5) < 0.03 and abs(((r[:, 4] >> 1) & 1).mean() - 0.8) < 0.03 and abs(((r[:, 4] >> 2) & 1).mean() - 0.2) < 0.03
    k = np.tile(np.arange(10), len(r) // 10)
    assert r[k == 0, 13].mean() > 0.99 and abs(r[k == 1, 13].mean() - 0.1) < 0.05 and abs(r[k >= 2, 13].mean() - 0.5) < 0.03
    assert abs(((r[k == 1, 4] >> 3) & 1).mean() - 0.2) < 0.06 and ((r[k != 1, 4] >> 3) & 1).sum() == 0


def test_resampling_table_matches_oracle():
    from lafs_cvpr2024_amd import augment as aug
    from oracle import augment as A
    tab = aug.coeff_table()
    for n in (1, 2, 5, 40, 71, 99, 111, 112):
        for xx, (xmin, k) in enumerate(A.resample_coeffs(n, 112)):
            assert tab[n, xx, 0] == xmin and tab[n, xx, 1] == len(k) and list(tab[n, xx, 2:2 + len(k)]) == k
    for r in (0.1, 0.5, 1.0, 1.7, 2.0):
        br, ww, fw = aug.gaussian_box(r)
        fr = A.gaussian_box_radius(r)
        assert br == int(fr) and ww == int(np.uint32((1 << 24) / (np.float32(fr) * np.float32(2) + np.float32(1))))


def test_sampled_parameters_follow_the_loader_distributions():
    from lafs_cvpr2024_amd import augment as aug
    rng = np.random.RandomState(0)
    ps = [aug.sample_view_params(rng, 8) for _ in range(400)]
    flat = [p for crops in ps for p in crops]
    area = np.array([p["h"] * p["w"] / 112.0 ** 2 for p in flat])
    assert 0.38 < area.min() and area.max() <= 1.0 and 0.6 < area.mean() < 0.8
    assert all(0 <= p["i"] <= 112 - p["h"] and 0 <= p["j"] <= 112 - p["w"] for p in flat)
    frac = lambda key, sel=lambda p: True: np.mean([bool(p[key]) for p in flat if sel(p)])
    assert abs(frac("flip") - 0.5) < 0.05 and abs(frac("jitter") - 0.8) < 0.04 and abs(frac("gray") - 0.2) < 0.04
    g1 = [c[0] for c in ps]; g2 = [c[1] for c in ps]; loc = [p for c in ps for p in c[2:]]
    assert all(p["blur_radius"] > 0 for p in g1) and abs(np.mean([p["blur_radius"] > 0 for p in g2]) - 0.1) < 0.05
    assert abs(np.mean([p["blur_radius"] > 0 for p in loc]) - 0.5) < 0.05
    assert abs(np.mean([p["solarize"] for p in g2]) - 0.2) < 0.06 and not any(p["solarize"] for p in g1 + loc)
