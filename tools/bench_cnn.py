#!/usr/bin/env python3
"""Landmark CNN (MobileNetV3-large trunk, 3x112x112 -> 160x4x4) timing on the GPU box: layouts / dtypes, inference at the
LAFS front-end batch (640) and training at the fine-tune batch (128)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lafs_cvpr2024_amd.face_pre_pro.mobilenet import MobileNetV3_backbone

dev = "cuda"
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

HIP_ONLY = "--hip-only" in sys.argv
torch.manual_seed(0)
if HIP_ONLY:
    from lafs_cvpr2024_amd.face_pre_pro.ViT_face import face_landmark_4simmin_glo_loc
    from lafs_cvpr2024_amd.landmark_cnn import HipLandmarkCNN
    lc = face_landmark_4simmin_glo_loc(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8, dim=64, depth=1,
                                       heads=1, mlp_dim=64).to(dev).eval()
    hip = HipLandmarkCNN(lc, dev)
    x = torch.randn(640, 3, 112, 112, device=dev)
    print(f"infer N=640 HIP plan (bf16 NHWC): {timeit(lambda: hip(x), n=20):8.2f} ms")
    sys.exit(0)
net = MobileNetV3_backbone(mode="large").to(dev).eval()
x = torch.randn(640, 3, 112, 112, device=dev)
with torch.no_grad():
    print(f"infer N=640 fp32 NCHW          : {timeit(lambda: net(x)):8.2f} ms")
    netc = net.to(memory_format=torch.channels_last); xc = x.contiguous(memory_format=torch.channels_last)
    print(f"infer N=640 fp32 channels_last : {timeit(lambda: netc(xc)):8.2f} ms")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        print(f"infer N=640 bf16 autocast CL   : {timeit(lambda: netc(xc)):8.2f} ms")
        print(f"infer N=640 bf16 autocast NCHW : {timeit(lambda: netc(x)):8.2f} ms")
    nb = MobileNetV3_backbone(mode="large").to(dev).eval().to(torch.bfloat16).to(memory_format=torch.channels_last)
    xb = xc.to(torch.bfloat16)
    print(f"infer N=640 pure bf16 CL       : {timeit(lambda: nb(xb)):8.2f} ms")
    nb2 = MobileNetV3_backbone(mode="large").to(dev).eval().to(torch.bfloat16)
    xb2 = x.to(torch.bfloat16)
    print(f"infer N=640 pure bf16 NCHW     : {timeit(lambda: nb2(xb2)):8.2f} ms")
from lafs_cvpr2024_amd.face_pre_pro.ViT_face import face_landmark_4simmin_glo_loc
from lafs_cvpr2024_amd.landmark_cnn import HipLandmarkCNN
lc = face_landmark_4simmin_glo_loc(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8, dim=64, depth=1, heads=1,
                                   mlp_dim=64).to(dev).eval()
hip = HipLandmarkCNN(lc, dev)
print(f"infer N=640 HIP plan (bf16 NHWC): {timeit(lambda: hip(x)):8.2f} ms   (includes the theta head)")
net.train()
x2 = torch.randn(128, 3, 112, 112, device=dev)
def fb(model, inp, ac=False):
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac):
        y = model(inp)
    y.float().mean().backward()
print(f"train N=128 fp32 NCHW          : {timeit(lambda: fb(net, x2)):8.2f} ms")
x2c = x2.contiguous(memory_format=torch.channels_last)
print(f"train N=128 fp32 CL            : {timeit(lambda: fb(netc, x2c)):8.2f} ms")
print(f"train N=128 bf16 autocast CL   : {timeit(lambda: fb(netc, x2c, True)):8.2f} ms")
print(f"train N=128 bf16 autocast NCHW : {timeit(lambda: fb(net, x2, True)):8.2f} ms")
