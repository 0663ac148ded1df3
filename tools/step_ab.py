#!/usr/bin/env python3
"""A/B timing of the graph-replayed C2 step under kernel-SELECTION debug flags (equally correct code paths; lafs_hip.h).
usage: python tools/step_ab.py 0 65536 2 4 ...   -> ms/step for each flag value, interleaved rounds"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lafs_cvpr2024_amd import _lib, vision_transformer as vits
from lafs_cvpr2024_amd.dino_loss import DINOLoss
from lafs_cvpr2024_amd.engine import LafsPretrainEngine
from lafs_cvpr2024_amd.utils import MultiCropWrapper

flags = [int(a, 0) for a in sys.argv[1:]] or [0]
B, K, nl = 64, 100000, 8
engines = {}
for f in flags:
    _lib.lib().lafs_debug_set(f)
    torch.manual_seed(0)
    student = MultiCropWrapper(vits.vit_small(patch_size=8, drop_path_rate=0.1), vits.DINOHead(384, K))
    teacher = MultiCropWrapper(vits.vit_small(patch_size=8), vits.DINOHead(384, K))
    teacher.load_state_dict(student.state_dict())
    eng = LafsPretrainEngine(student, teacher, DINOLoss(K, 2 + nl, 0.07, 0.04, 30, 41), B, n_local=nl, device="cuda")
    eng.in_global_all.normal_().clamp_(-1, 1); eng.in_local_all.normal_().clamp_(-1, 1)
    for _ in range(3):
        eng.step(lr=1e-4, wd=0.04, momentum=0.996, teacher_temp=0.04, epoch=1)      # captures the graphs under this flag
    engines[f] = eng
_lib.lib().lafs_debug_set(0)
res = {f: [] for f in flags}
for r in range(4):
    for f, eng in engines.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            eng.step(lr=1e-4, wd=0.04, momentum=0.996, teacher_temp=0.04, epoch=1)
        torch.cuda.synchronize()
        res[f].append((time.perf_counter() - t0) / 10 * 1e3)
for f in flags:
    print(f"flag {f:6d}: {min(res[f]):7.3f} ms/step (min of 4 rounds; {', '.join(f'{v:.2f}' for v in res[f])})")
