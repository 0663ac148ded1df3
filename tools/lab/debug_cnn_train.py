#!/usr/bin/env python3
"""GPU box: the HIP training plan of the landmark CNN, stage by stage, against its torch statement with bf16 roundings at the same
points (fp32 torch ops on the GPU) -- locates the first stage where they part."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import torch.nn.functional as F
from conftest import det_fill_random
from lafs_cvpr2024_amd.face_pre_pro.ViT_face import ViT_face_landmark_patch8
from lafs_cvpr2024_amd.landmark_train import HipLandmarkTrainer
from lafs_cvpr2024_amd.vision_transformer import attach_arena

DEV = "cuda"
torch.manual_seed(0)
B = 8
m = ViT_face_landmark_patch8(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8, dim=64, depth=1, heads=1,
                             mlp_dim=64, dropout=0.0, emb_dropout=0.0, with_land=True)
det_fill_random(m.stn); det_fill_random(m.output_layer)
arena = attach_arena(m, DEV)
m.train()
tr = HipLandmarkTrainer(m, arena, B, 112, device=DEV)
x = torch.randn(B, 3, 112, 112, device=DEV).clamp(-1, 1)
keep = (torch.rand(B, 160, device=DEV) >= 0.5).float() / 0.5
tr.fixed_drop = keep
theta = tr.forward(x)
torch.cuda.synchronize()
rnd = lambda t: t.to(torch.bfloat16).float()
def bn(x, b):
    mu = x.mean(dim=(0, 2, 3), keepdim=True); var = x.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    return (x - mu) / torch.sqrt(var + b.eps) * b.weight.view(1, -1, 1, 1) + b.bias.view(1, -1, 1, 1)
def nhwc(buf, N, H, C):
    return buf.view(N, H, H, -1)[..., :C].permute(0, 3, 1, 2).float()
def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))
with torch.no_grad():
    feats = m.stn.features
    s_raw = rnd(F.conv2d(rnd(x), rnd(feats[0][0].weight), None, 2, 1))
    print("stem_raw", rel(nhwc(tr.B["s_raw"], B, 56, 16), s_raw), " exact-equal frac", float((nhwc(tr.B["s_raw"], B, 56, 16) == s_raw).float().mean()))
    cur = rnd(feats[0][2](bn(s_raw, feats[0][1])))
    print("x0", rel(nhwc(tr.B["x0"], B, 56, 16), cur))
    for i, (blk, L, D) in enumerate(zip(feats[1:], tr.blocks, tr.B["layers"]), start=1):
        c = blk.conv
        # feed the PLAN's own input to each stage so that errors do not accumulate: isolates the stage
        cur_p = nhwc(D["x_in"], B, D["H"], L["cin"])
        e_raw = rnd(F.conv2d(cur_p, rnd(c[0].weight)))
        r1 = rel(nhwc(D["e_raw"], B, D["H"], L["cexp"]), e_raw)
        e = rnd(c[2](bn(nhwc(D["e_raw"], B, D["H"], L["cexp"]), c[1])))
        r2 = rel(nhwc(D["e"], B, D["H"], L["cexp"]), e)
        d_raw = rnd(F.conv2d(nhwc(D["e"], B, D["H"], L["cexp"]), c[3].weight, None, c[3].stride, c[3].padding, 1, c[3].groups))
        r3 = rel(nhwc(D["d_raw"], B, D["Ho"], L["cexp"]), d_raw)
        dr = nhwc(D["d_raw"], B, D["Ho"], L["cexp"])
        if L["se"] is None:
            d = rnd(c[6](bn(dr, c[4])))
            r4 = rel(nhwc(D["d"], B, D["Ho"], L["cexp"]), d); rse = ""
        else:
            zb = rnd(bn(dr, c[4]))
            rz = rel(nhwc(D["zb"], B, D["Ho"], L["cexp"]), zb)
            zbp = nhwc(D["zb"], B, D["Ho"], L["cexp"])
            pool = rnd(zbp.mean(dim=(2, 3)))
            rp = rel(D["pool"][:, :L["cexp"]].float(), pool)
            hid = rnd(F.relu(F.linear(D["pool"][:, :L["cexp"]].float(), rnd(c[5].fc[0].weight))))
            rh = rel(D["hid"][:, :L["se"]["h"]].float(), hid)
            gate = rnd(F.hardsigmoid(F.linear(D["hid"][:, :L["se"]["h"]].float(), rnd(c[5].fc[2].weight))))
            rg = rel(D["gate"][:, :L["cexp"]].float(), gate)
            d = rnd(c[6](zbp * D["gate"][:, :L["cexp"]].float()[:, :, None, None]))
            r4 = rel(nhwc(D["d"], B, D["Ho"], L["cexp"]), d)
            rse = f" zb {rz:.1e} pool {rp:.1e} hid {rh:.1e} gate {rg:.1e}"
        y_raw = rnd(F.conv2d(nhwc(D["d"], B, D["Ho"], L["cexp"]), rnd(c[7].weight)))
        r5 = rel(nhwc(D["y_raw"], B, D["Ho"], L["cout"]), y_raw)
        y = bn(nhwc(D["y_raw"], B, D["Ho"], L["cout"]), c[8])
        y = rnd(y + cur_p if blk.residual else y)
        r6 = rel(nhwc(D["y"], B, D["Ho"], L["cout"]), y)
        print(f"b{i:2d} e_raw {r1:.1e} e {r2:.1e} d_raw {r3:.1e} d {r4:.1e} y_raw {r5:.1e} y {r6:.1e}{rse}")
# ---- chained: the statement run end to end on its own intermediates
with torch.no_grad():
    cur = rnd(feats[0][2](bn(rnd(F.conv2d(rnd(x), rnd(feats[0][0].weight), None, 2, 1)), feats[0][1])))
    for i, (blk, L, D) in enumerate(zip(feats[1:], tr.blocks, tr.B["layers"]), start=1):
        c = blk.conv
        e = rnd(c[2](bn(rnd(F.conv2d(cur, rnd(c[0].weight))), c[1])))
        d_raw = rnd(F.conv2d(e, c[3].weight, None, c[3].stride, c[3].padding, 1, c[3].groups))
        if L["se"] is None:
            d = rnd(c[6](bn(d_raw, c[4])))
        else:
            zb = rnd(bn(d_raw, c[4]))
            hid = rnd(F.relu(F.linear(rnd(zb.mean(dim=(2, 3))), rnd(c[5].fc[0].weight))))
            gate = rnd(F.hardsigmoid(F.linear(hid, rnd(c[5].fc[2].weight))))
            d = rnd(c[6](zb * gate[:, :, None, None]))
        y = bn(rnd(F.conv2d(d, rnd(c[7].weight))), c[8])
        cur = rnd(y + cur if blk.residual else y)
        print(f"chained b{i:2d} y {rel(nhwc(D['y'], B, D['Ho'], L['cout']), cur):.2e}")
    feat = rnd(rnd(cur.mean(dim=(2, 3))) * keep)
    t = F.linear(feat, rnd(m.output_layer[1].weight), m.output_layer[1].bias)
    print("chained feat", rel(tr.B["featd"][:, :160].float(), feat), "t", rel(tr.B["t"], t))
