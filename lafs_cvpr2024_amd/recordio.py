"""MXNet RecordIO reader for the face datasets the reference trains on (image_iter.py:187-300 `FaceDataset`: `train.rec` /
`train.idx` in the InsightFace layout), without MXNet.

PARITY UNPINNED: `mxnet` is a third-party dependency of the reference that is not installed in this image, so the format is
restated from its published definition (dmlc-core `recordio.h`, `python/mxnet/recordio.py`) and checked by write -> read round
trips only (tests/test_recordio.py):

    record   := u32 magic 0xced7230a | u32 (cflag << 29 | length) | payload | zero padding to a multiple of 4 bytes
                cflag 0 = whole record; 1 / 2 / 3 = first / middle / last part of a payload that contained the magic word
                (the writer splits there and drops the 4 magic bytes; the reader re-inserts them between parts)
    payload  := IRHeader '<IfQQ' (flag, label, id, id2) | [flag x f32 labels when flag > 0] | encoded image bytes
    .idx     := text lines "key\\toffset"

`FaceRecordDataset` reproduces FaceDataset's index logic (record 0 holds the [first, last) range of the identity headers, each
of which holds the [first, last) range of its images) and yields (uint8 HWC RGB image, integer label); JPEG decoding is Pillow's
(the reference uses mx.image.imdecode, i.e. OpenCV).  `device_batches` turns it into uint8 NCHW device batches for
augment.DeviceAugmenter -- the rest of the reference's loader (PIL transforms in 6 workers) runs on the GPU here.
"""
import io
import os
import struct
from collections import namedtuple

import numpy as np

MAGIC = 0xCED7230A
_MAGIC_BYTES = struct.pack("<I", MAGIC)
IRHeader = namedtuple("IRHeader", ["flag", "label", "id", "id2"])
_IR_FORMAT = "<IfQQ"
_IR_SIZE = struct.calcsize(_IR_FORMAT)


def pack(header, payload):
    """mxnet.recordio.pack: header (flag, label, id, id2) + bytes -> record payload."""
    label = header.label
    if isinstance(label, (int, float)):
        head = struct.pack(_IR_FORMAT, 0, float(label), header.id, header.id2)
    else:
        lab = np.asarray(label, dtype=np.float32)
        head = struct.pack(_IR_FORMAT, lab.size, 0.0, header.id, header.id2) + lab.tobytes()
    return head + payload


def unpack(s):
    """mxnet.recordio.unpack: record payload -> (IRHeader, image bytes)."""
    flag, label, id_, id2 = struct.unpack(_IR_FORMAT, s[:_IR_SIZE])
    s = s[_IR_SIZE:]
    if flag > 0:
        label = np.frombuffer(s[:flag * 4], dtype=np.float32).copy()
        s = s[flag * 4:]
    return IRHeader(flag, label, id_, id2), s


class IndexedRecordWriter:
    """Writes `.rec` + `.idx` (used by the tests and to build synthetic datasets)."""

    def __init__(self, idx_path, rec_path):
        self.f = open(rec_path, "wb")
        self.fidx = open(idx_path, "w")

    def write_idx(self, key, payload):
        self.fidx.write(f"{key}\t{self.f.tell()}\n")
        parts = payload.split(_MAGIC_BYTES)                   # dmlc: the magic word never appears inside a stored part
        for n, part in enumerate(parts):
            if len(parts) == 1:
                cflag = 0
            elif n == 0:
                cflag = 1
            elif n == len(parts) - 1:
                cflag = 3
            else:
                cflag = 2
            self.f.write(struct.pack("<II", MAGIC, (cflag << 29) | len(part)))
            self.f.write(part)
            self.f.write(b"\x00" * ((4 - len(part) % 4) % 4))

    def close(self):
        self.f.close(); self.fidx.close()


class IndexedRecordIO:
    """Random-access reader (mxnet.recordio.MXIndexedRecordIO, read mode)."""

    def __init__(self, idx_path, rec_path):
        self.rec_path = rec_path
        self.idx = {}
        self.keys = []
        with open(idx_path) as f:
            for line in f:
                k, off = line.strip().split("\t")
                self.idx[int(k)] = int(off)
                self.keys.append(int(k))
        self._f, self._pid = None, None

    def _file(self):
        # One handle per PROCESS: DataLoader workers are forked after the parent has already read the headers, and a file
        # description shared across processes would interleave their seek/read pairs.
        if self._f is None or self._pid != os.getpid():
            self._f, self._pid = open(self.rec_path, "rb"), os.getpid()
        return self._f

    def __getstate__(self):
        d = dict(self.__dict__); d["_f"], d["_pid"] = None, None
        return d

    def read_idx(self, key):
        f = self._file()
        f.seek(self.idx[key])
        out = []
        while True:
            head = f.read(8)
            if len(head) < 8:
                raise EOFError("truncated record")
            magic, lrec = struct.unpack("<II", head)
            if magic != MAGIC:
                raise ValueError(f"bad RecordIO magic {magic:#x} at key {key}")
            cflag, length = lrec >> 29, lrec & ((1 << 29) - 1)
            data = f.read(length)
            f.read((4 - length % 4) % 4)
            out.append(data)
            if cflag in (0, 3):
                break
        return _MAGIC_BYTES.join(out)


class FaceRecordDataset:
    """The index logic of the reference's FaceDataset (image_iter.py:262-296) + Pillow decoding."""

    def __init__(self, path_imgrec, partition=1):
        self.rec = IndexedRecordIO(path_imgrec[:-4] + ".idx", path_imgrec)
        header, _ = unpack(self.rec.read_idx(0))
        if header.flag > 0:
            self.header0 = (int(header.label[0]), int(header.label[1]))
            self.imgidx, self.id2range = [], {}
            for identity in range(self.header0[0], self.header0[1]):
                h, _ = unpack(self.rec.read_idx(identity))
                a, b = int(h.label[0]), int(h.label[1])
                self.id2range[identity] = (a, b)
                self.imgidx += range(a, b)
        else:
            self.imgidx = list(self.rec.keys)
        self.seq = self.imgidx[: int(len(self.imgidx) * partition)] if partition else self.imgidx

    def __len__(self):
        return len(self.seq)

    def __getitem__(self, index):
        from PIL import Image
        header, img = unpack(self.rec.read_idx(self.seq[index]))
        label = header.label
        if not isinstance(label, (int, float)):
            label = label[0]
        arr = np.asarray(Image.open(io.BytesIO(img)).convert("RGB"))
        return arr, int(label)


def device_batches(dataset, batch_size, device, num_workers=6, shuffle=True, seed=0, drop_last=True):
    """uint8 NCHW device batches [B,3,H,W] + int64 labels from a FaceRecordDataset (decode on CPU workers, everything after
    it on the device)."""
    import torch

    def collate(items):
        x = torch.from_numpy(np.stack([it[0] for it in items])).permute(0, 3, 1, 2).contiguous()
        return x, torch.tensor([it[1] for it in items], dtype=torch.int64)

    g = torch.Generator().manual_seed(seed)
    loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers, drop_last=drop_last,
                                         collate_fn=collate, pin_memory=(torch.device(device).type == "cuda"), generator=g,
                                         persistent_workers=num_workers > 0)
    for x, y in loader:
        yield x.to(device, non_blocking=True), y.to(device, non_blocking=True)
