"""GPU box: time of the device RandAugment on a fine-tune batch (128 x 112 x 112 x 3) against Pillow on one host core."""
import random
import time

import numpy as np
import torch

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lafs_cvpr2024_amd import randaug as P

B = 128
x = torch.randint(0, 256, (B, 3, 112, 112), dtype=torch.uint8, device="cuda")
for cfg in ("rand-m1-mstd0.5-inc1", "rand-m9-n3-mstd0.5-inc1"):
    aug = P.DeviceRandAugment(cfg, seed=0)
    recs = aug.sample(B)
    for _ in range(5):
        aug(x, records=recs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50):
        aug(x, records=recs)
    e1.record(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        aug.sample(B)
    ts = (time.time() - t0) / 20
    print(f"{cfg}: device {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per batch of {B} (launch + record upload), host sampling {ts * 1e3:.2f} ms per batch")
try:
    import sys
    sys.path.insert(0, ".")
    from oracle import randaug as R  # noqa: F401  (timing yardstick only)
    from PIL import Image
    imgs = x.permute(0, 2, 3, 1).cpu().numpy()
    aug = P.DeviceRandAugment("rand-m1-mstd0.5-inc1", seed=0)
    recs = aug.sample(B)
    t0 = time.time()
    for b in range(B):
        im = Image.fromarray(imgs[b])
        for r in recs[b]:
            if r["op"] in (3, 9, 10, 11, 12):
                im = im.transform(im.size, Image.AFFINE, tuple(r["m"]), resample=int(r["resample"]), fillcolor=(128, 128, 128))
    print(f"Pillow, geometric ops of the same records only, one core: {(time.time() - t0) * 1e3:.1f} ms per batch")
except Exception as e:
    print("pillow timing skipped:", e)
