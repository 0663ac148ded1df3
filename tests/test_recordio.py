"""CPU: the MXNet RecordIO restatement (lafs_cvpr2024_amd/recordio.py; parity unpinned -- MXNet is not installed -- so these are
round trips through the published format: InsightFace-style train.rec / train.idx, records containing the magic word)."""
import io
import os
import struct

import numpy as np
import pytest

from lafs_cvpr2024_amd import recordio as R


def _png(arr):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(arr).save(b, format="PNG")
    return b.getvalue()


def _build(tmp_path, n_ids=3, per_id=(2, 3, 1)):
    """InsightFace layout: key 0 = header with [first identity key, last identity key), image keys 1..n, then identity keys."""
    rng = np.random.RandomState(0)
    rec, idx = str(tmp_path / "train.rec"), str(tmp_path / "train.idx")
    w = R.IndexedRecordWriter(idx, rec)
    n_img = sum(per_id)
    imgs, labels = [], []
    w_key = 1
    ranges = []
    for ident, cnt in enumerate(per_id):
        ranges.append((w_key, w_key + cnt))
        for _ in range(cnt):
            a = rng.randint(0, 256, (112, 112, 3)).astype(np.uint8)
            imgs.append(a); labels.append(ident)
            w_key += 1
    w.write_idx(0, R.pack(R.IRHeader(2, [n_img + 1, n_img + 1 + n_ids], 0, 0), b""))
    for k, (a, lab) in enumerate(zip(imgs, labels)):
        w.write_idx(k + 1, R.pack(R.IRHeader(0, float(lab), k + 1, 0), _png(a)))
    for ident, (a, b) in enumerate(ranges):
        w.write_idx(n_img + 1 + ident, R.pack(R.IRHeader(2, [a, b], n_img + 1 + ident, 0), b""))
    w.close()
    return rec, imgs, labels


def test_face_record_dataset_round_trip(tmp_path):
    rec, imgs, labels = _build(tmp_path)
    ds = R.FaceRecordDataset(rec)
    assert len(ds) == len(imgs) and ds.header0 == (7, 10) and ds.id2range[7] == (1, 3)
    for i in range(len(ds)):
        arr, lab = ds[i]
        assert arr.dtype == np.uint8 and arr.shape == (112, 112, 3) and lab == labels[i]
        assert np.array_equal(arr, imgs[i])
    assert len(R.FaceRecordDataset(rec, partition=0.5)) == 3


def test_payload_containing_the_magic_word_is_split_and_rejoined(tmp_path):
    rec, idx = str(tmp_path / "x.rec"), str(tmp_path / "x.idx")
    magic = struct.pack("<I", R.MAGIC)
    payloads = [b"abc" + magic + b"defgh" + magic, magic + b"z", b"plain-payload", b""]
    w = R.IndexedRecordWriter(idx, rec)
    for k, p in enumerate(payloads):
        w.write_idx(k, R.pack(R.IRHeader(0, float(k), k, 0), p))
    w.close()
    r = R.IndexedRecordIO(idx, rec)
    assert r.keys == [0, 1, 2, 3]
    for k, p in enumerate(payloads):
        h, body = R.unpack(r.read_idx(k))
        assert body == p and h.flag == 0 and h.label == float(k) and h.id == k
    assert os.path.getsize(rec) % 4 == 0
    with open(rec, "rb") as f:                              # every stored part starts with the magic word
        assert struct.unpack("<I", f.read(4))[0] == R.MAGIC


def test_forked_loader_workers_do_not_share_the_file_offset(tmp_path):
    """The parent reads the identity headers before the DataLoader forks its workers: each process must get its own handle."""
    torch = pytest.importorskip("torch")
    rec, imgs, labels = _build(tmp_path, n_ids=4, per_id=(6, 6, 6, 6))
    ds = R.FaceRecordDataset(rec)
    seen = 0
    for _ in range(3):
        for x, y in R.device_batches(ds, 4, "cpu", num_workers=3, shuffle=False):
            for b in range(x.shape[0]):
                assert np.array_equal(x[b].permute(1, 2, 0).numpy(), imgs[seen % len(imgs)]) and int(y[b]) == labels[seen % len(imgs)]
                seen += 1
    assert seen == 3 * len(imgs)


def test_device_batches_on_cpu(tmp_path):
    torch = pytest.importorskip("torch")
    rec, imgs, labels = _build(tmp_path)
    ds = R.FaceRecordDataset(rec)
    got = list(R.device_batches(ds, 2, "cpu", num_workers=0, shuffle=False))
    assert len(got) == 3 and got[0][0].shape == (2, 3, 112, 112) and got[0][0].dtype == torch.uint8
    assert np.array_equal(got[0][0][1].permute(1, 2, 0).numpy(), imgs[1]) and got[1][1].tolist() == labels[2:4]
