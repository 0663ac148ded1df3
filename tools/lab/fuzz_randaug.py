"""GPU box: 768 random images x three RandAugment configurations (incl. uniform magnitudes, 4 layers): device against the oracle."""
import sys, random, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lafs_cvpr2024_amd import randaug as P
from oracle import randaug as R
rng = np.random.RandomState(123)
bad = 0; n = 0
for cfg, seed in (("rand-m9-n3-mstd0.5-inc1", 1), ("rand-m10-n2-mstdinf-inc1", 2), ("rand-m5-n4-mstd1-inc1", 3)):
    B = 256
    imgs = np.stack([(rng.randint(0, 256, (112, 112, 3)) if k % 2 else np.clip(128 + 70 * rng.randn(14, 14, 3), 0, 255).repeat(8, 0).repeat(8, 1)).astype(np.uint8) for k in range(B)])
    aug = P.DeviceRandAugment(cfg, seed=seed)
    recs = aug.sample(B)
    got = aug(torch.from_numpy(imgs).cuda(), records=recs).cpu().numpy()
    m, nl, sd, _ = R.parse_config(cfg)
    rnd, nprnd = random.Random(seed), np.random.RandomState(seed)
    for b in range(B):
        ref = R.apply_record(imgs[b], R.sample_record(rnd, nprnd, m, nl, sd))
        n += 1
        if not np.array_equal(ref, got[b]): bad += 1
print("images", n, "mismatching", bad)
