#!/usr/bin/env python3
"""Write a small synthetic face dataset in the InsightFace RecordIO layout (train.rec / train.idx) for end-to-end runs of
`lafs_train.py --data recordio`.  usage: tools/make_synthetic_rec.py OUT_DIR [n_identities] [images_per_identity]"""
import io
import os
import sys

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lafs_cvpr2024_amd import recordio as R

out = sys.argv[1]
n_ids = int(sys.argv[2]) if len(sys.argv) > 2 else 32
per = int(sys.argv[3]) if len(sys.argv) > 3 else 16
os.makedirs(out, exist_ok=True)
rng = np.random.RandomState(0)
w = R.IndexedRecordWriter(os.path.join(out, "train.idx"), os.path.join(out, "train.rec"))
n_img = n_ids * per
w.write_idx(0, R.pack(R.IRHeader(2, [n_img + 1, n_img + 1 + n_ids], 0, 0), b""))
yy, xx = np.mgrid[0:112, 0:112]
key = 1
for ident in range(n_ids):
    base = np.stack([127 + 90 * np.sin(xx / (5.0 + ident % 7) + c) * np.cos(yy / (6.0 + c) - ident) for c in range(3)], -1)
    for _ in range(per):
        img = np.clip(base + rng.randn(112, 112, 3) * 25, 0, 255).astype(np.uint8)
        b = io.BytesIO(); Image.fromarray(img).save(b, format="JPEG", quality=95)
        w.write_idx(key, R.pack(R.IRHeader(0, float(ident), key, 0), b.getvalue()))
        key += 1
for ident in range(n_ids):
    w.write_idx(n_img + 1 + ident, R.pack(R.IRHeader(2, [1 + ident * per, 1 + (ident + 1) * per], n_img + 1 + ident, 0), b""))
w.close()
print(f"wrote {n_img} images of {n_ids} identities to {out}/train.rec")
