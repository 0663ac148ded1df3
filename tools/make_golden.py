#!/usr/bin/env python3
"""Generate the golden parity fixtures under tests/golden/ by running the REFERENCE itself.

Runs only in the build container, where the read-only reference checkout is mounted at
/root/reference.  It imports the reference's own modules on CPU (fp32), feeds them small seeded
inputs and stores inputs + parameters + outputs as .npz files.  The fixtures are data only; no
reference source travels with the repo.  Import recipe follows SURVEY.md section 8c:

  * `utils`, `vision_transformer`, `util.mixup_my` import unmodified;
  * `lafs_train` needs empty `torchvision{,.datasets,.transforms,.models}` stubs;
  * `face_pre_pro.ViT_face` additionally needs `IPython.embed` and
    `timm.models.layers.{DropPath, trunc_normal_}` (mapped to vision_transformer.DropPath and
    torch.nn.init.trunc_normal_);
  * hard-coded `.cuda()` calls are neutralised with `torch.Tensor.cuda = identity`;
  * DINOLoss needs a process group -> gloo, world_size 1.

Usage:  python tools/make_golden.py   (rewrites tests/golden/*.npz deterministically)
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("LAFS_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    for name in ("torchvision", "torchvision.datasets", "torchvision.transforms", "torchvision.models"):
        sys.modules.setdefault(name, types.ModuleType(name))
    tv = sys.modules["torchvision"]
    tv.datasets, tv.transforms, tv.models = (sys.modules["torchvision.datasets"],
                                             sys.modules["torchvision.transforms"],
                                             sys.modules["torchvision.models"])
    ip = types.ModuleType("IPython"); ip.embed = lambda *a, **k: None
    sys.modules.setdefault("IPython", ip)
    import utils as ref_utils                      # noqa: E402  (reference module)
    import vision_transformer as ref_vit           # noqa: E402
    timm = types.ModuleType("timm"); tm = types.ModuleType("timm.models"); tl = types.ModuleType("timm.models.layers")
    tl.DropPath = ref_vit.DropPath
    tl.trunc_normal_ = torch.nn.init.trunc_normal_
    timm.models, tm.layers = tm, tl
    for n, m in (("timm", timm), ("timm.models", tm), ("timm.models.layers", tl)):
        sys.modules.setdefault(n, m)
    torch.Tensor.cuda = lambda self, *a, **k: self
    import lafs_train as ref_lafs                  # noqa: E402
    import face_pre_pro.ViT_face as ref_face       # noqa: E402
    import util.mixup_my as ref_mix                # noqa: E402
    return ref_utils, ref_vit, ref_lafs, ref_face, ref_mix


def npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


ONLY = os.environ.get("LAFS_GOLDEN_ONLY", "")       # e.g. "f13": rewrite only the fixtures whose name starts with it


def save(name, **arrays):
    if ONLY and not name.startswith(ONLY):
        return
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrays.items()})
    print(f"  wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB, {len(arrays)} arrays)")


def sd(module, prefix="p."):
    return {prefix + k: v for k, v in module.state_dict().items()}


def grads(module, prefix="g."):
    return {prefix + k: p.grad for k, p in module.named_parameters() if p.grad is not None}


def main():
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("gloo", rank=0, world_size=1)
    ref_utils, ref_vit, ref_lafs, ref_face, ref_mix = _import_reference()
    torch.set_num_threads(4)

    # ---------------------------------------------------------------- F19 RandAugment of the fine-tune loader
    # util/rand_aa_face.py as FaceDataset builds it (dataloader_web.py:240-243, train_largescale.py:506): 'rand-m1-mstd0.5-inc1',
    # hparams {'translate_const': 117}; plus a strong 3-layer configuration so that every op is far from the identity.  Image i is
    # transformed with random.seed(seed_i); np.random.seed(seed_i) set right before the call -- the decisions are reproducible.
    print("F19 fine-tune RandAugment")
    import random as _random
    from PIL import Image as _Image
    from util import rand_aa_face as ref_ra
    g19 = np.random.RandomState(19)
    yy, xx = np.mgrid[0:112, 0:112]
    imgs19 = []
    for k in range(40):
        kind = k % 5
        if kind == 0:
            a = g19.randint(0, 256, (112, 112, 3))
        elif kind == 1:
            a = np.stack([(xx * (k + 1)) % 256, (yy * 2 + xx * k) % 256, (xx * yy // (k + 1)) % 256], -1)
        elif kind == 2:
            a = 40 + 150 * (0.5 + 0.5 * np.sin(xx[..., None] * 0.07 * (1 + np.arange(3)) + yy[..., None] * 0.05 + k))
        elif kind == 3:
            a = (g19.rand(112, 112, 3) ** 3) * 255
        else:
            a = np.clip(128 + 60 * g19.randn(14, 14, 3), 0, 255).repeat(8, 0).repeat(8, 1)
        imgs19.append(np.asarray(a).astype(np.uint8))
    imgs19 = np.stack(imgs19)
    out19 = {}
    for tag, cfgs in (("m1", "rand-m1-mstd0.5-inc1"), ("m9n3", "rand-m9-n3-mstd0.5-inc1")):
        tr = ref_ra.rand_augment_transform(cfgs, {"translate_const": 117})
        res = []
        for i in range(len(imgs19)):
            _random.seed(1900 + i); np.random.seed(1900 + i)
            res.append(np.asarray(tr(_Image.fromarray(imgs19[i]))))
        out19["out_" + tag] = np.stack(res)
        out19["cfg_" + tag] = np.array(cfgs)
    save("f19_randaugment", images=imgs19, seed0=np.int64(1900), **out19)
    if ONLY == "f19":
        return

    # ---------------------------------------------------------------- F1 VisionTransformer fwd/bwd
    # tiny geometry that the HIP kernels also accept (head_dim 64): D=128, 2 heads, depth 2, p=8,
    # pos table 28x28 (img_size 224) resampled to 14x14 and 6x6.
    print("F1 vit")
    torch.manual_seed(1)
    vitm = ref_vit.VisionTransformer(img_size=[224], patch_size=8, embed_dim=128, depth=2, num_heads=2,
                                     mlp_ratio=4, qkv_bias=True,
                                     norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6), drop_path_rate=0.0)
    with torch.no_grad():                           # non-trivial affine / biases so that they are exercised
        for n_, p_ in vitm.named_parameters():
            if n_.endswith("bias") or "norm" in n_:
                p_.add_(0.1 * torch.randn_like(p_))
    xg = torch.randn(2, 3, 112, 112).clamp(-1, 1)
    xl = torch.randn(3, 3, 48, 48).clamp(-1, 1)
    wg, wl = torch.randn(2, 128), torch.randn(3, 128)
    og = vitm(xg); ol = vitm(xl)
    ((og * wg).sum() + (ol * wl).sum()).backward()
    pos14 = vitm.interpolate_pos_encoding(torch.zeros(1, 197, 128), 112, 112)
    pos6 = vitm.interpolate_pos_encoding(torch.zeros(1, 37, 128), 48, 48)
    save("f1_vit", xg=xg, xl=xl, wg=wg, wl=wl, og=og, ol=ol, pos14=pos14, pos6=pos6, **sd(vitm), **grads(vitm))

    # ---------------------------------------------------------------- F2 DINOHead fwd/bwd
    print("F2 dino head")
    torch.manual_seed(2)
    head = ref_vit.DINOHead(128, 1000, use_bn=False, norm_last_layer=True, nlayers=3, hidden_dim=256,
                            bottleneck_dim=64)
    xh = torch.randn(6, 128, requires_grad=True)
    wh = torch.randn(6, 1000)
    oh = head(xh)
    (oh * wh).sum().backward()
    save("f2_head", x=xh, w=wh, out=oh, gx=xh.grad, **sd(head), **grads(head))
    head2 = ref_vit.DINOHead(128, 1000, use_bn=False, norm_last_layer=False, nlayers=3, hidden_dim=256,
                             bottleneck_dim=64)
    with torch.no_grad():
        head2.last_layer.weight_g.mul_(1 + 0.2 * torch.randn_like(head2.last_layer.weight_g))
    xh2 = torch.randn(6, 128, requires_grad=True)
    oh2 = head2(xh2)
    (oh2 * wh).sum().backward()
    save("f2_head_freeg", x=xh2, w=wh, out=oh2, gx=xh2.grad, **sd(head2), **grads(head2))

    # ---------------------------------------------------------------- F3 MultiCropWrapper
    print("F3 multicrop")
    torch.manual_seed(3)
    vit3 = ref_vit.VisionTransformer(img_size=[112], patch_size=8, embed_dim=64, depth=1, num_heads=1,
                                     qkv_bias=True, norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6))
    head3 = ref_vit.DINOHead(64, 200, hidden_dim=128, bottleneck_dim=64)
    mc = ref_utils.MultiCropWrapper(vit3, head3)
    crops = [torch.randn(2, 3, 112, 112).clamp(-1, 1) for _ in range(2)] + \
            [torch.randn(2, 3, 48, 48).clamp(-1, 1) for _ in range(3)]
    out3 = mc(crops)
    save("f3_multicrop", out=out3, **{f"crop{i}": c for i, c in enumerate(crops)}, **sd(mc))

    # ---------------------------------------------------------------- F4 DINOLoss
    print("F4 dino loss")
    for ncrops in (4, 10):
        torch.manual_seed(40 + ncrops)
        K, B = 1000, 3
        crit = ref_lafs.DINOLoss(K, ncrops, 0.07, 0.04, 6, 10)
        crit.center.copy_(0.05 * torch.randn(1, K))
        res = {}
        for epoch in (0, 7):
            s = (0.5 * torch.randn(ncrops * B, K)).requires_grad_(True)
            t = 0.5 * torch.randn(2 * B, K)
            c0 = crit.center.clone()
            loss = crit(s, t, epoch)
            loss.backward()
            res.update({f"e{epoch}_student": s, f"e{epoch}_teacher": t, f"e{epoch}_center_before": c0,
                        f"e{epoch}_loss": loss, f"e{epoch}_grad": s.grad, f"e{epoch}_center_after": crit.center,
                        f"e{epoch}_temp": np.float64(crit.teacher_temp_schedule[epoch])})
        save(f"f4_dinoloss_nc{ncrops}", schedule=crit.teacher_temp_schedule, **res)

    # ---------------------------------------------------------------- F5 full LAFS step x2
    # Re-assembled from the reference's own components in the order of lafs_train.py:577-613.
    print("F5 lafs step")
    torch.manual_seed(5)
    K, B, ncrops = 512, 2, 5
    mk = lambda dpr: ref_vit.VisionTransformer(img_size=[112], patch_size=8, embed_dim=64, depth=2, num_heads=1,
                                               qkv_bias=True, drop_path_rate=dpr,
                                               norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6))
    student = ref_utils.MultiCropWrapper(mk(0.0), ref_vit.DINOHead(64, K, hidden_dim=128, bottleneck_dim=64,
                                                                   norm_last_layer=True))
    teacher = ref_utils.MultiCropWrapper(mk(0.0), ref_vit.DINOHead(64, K, hidden_dim=128, bottleneck_dim=64))
    teacher.load_state_dict(student.state_dict())
    for p in teacher.parameters():
        p.requires_grad = False
    crit = ref_lafs.DINOLoss(K, ncrops, 0.07, 0.04, 3, 10)
    opt = torch.optim.AdamW(ref_utils.get_params_groups(student))
    groups = ref_utils.get_params_groups(student)
    reg_ids = {id(p) for p in groups[0]["params"]}
    membership = {n: (id(p) in reg_ids) for n, p in student.named_parameters() if p.requires_grad}
    fx = {"init." + k: v.clone() for k, v in student.state_dict().items()}
    lrs, wds, moms = [5e-4, 4e-4], [0.04, 0.05], [0.9, 0.95]           # large (1-m) so EMA is visible
    for step in range(2):
        epoch = step                                 # step 0 -> epoch 0 (last layer frozen), step 1 -> epoch 1
        imgs = [torch.randn(B, 3, 112, 112).clamp(-1, 1) for _ in range(2)] + \
               [torch.randn(B, 3, 48, 48).clamp(-1, 1) for _ in range(ncrops - 2)]
        for i, g in enumerate(opt.param_groups):
            g["lr"] = lrs[step]
            if i == 0:
                g["weight_decay"] = wds[step]
        t_out = teacher(imgs[:2]); s_out = student(imgs)
        loss = crit(s_out, t_out, epoch)
        opt.zero_grad()
        loss.backward()
        norms = ref_utils.clip_gradients(student, 3.0)
        post = {n: p.grad.clone() for n, p in student.named_parameters() if p.grad is not None}
        ref_utils.cancel_gradients_last_layer(epoch, student, 1)
        opt.step()
        with torch.no_grad():
            for pq, pk in zip(student.parameters(), teacher.parameters()):
                pk.data.mul_(moms[step]).add_((1 - moms[step]) * pq.detach().data)
        fx.update({f"s{step}.crop{i}": im for i, im in enumerate(imgs)})
        fx.update({f"s{step}.loss": loss, f"s{step}.center": crit.center, f"s{step}.t_out": t_out,
                   f"s{step}.s_out": s_out, f"s{step}.norms": np.array(norms)})
        fx.update({f"s{step}.grad_post.{n}": g for n, g in post.items()})
        fx.update({f"s{step}.student.{k}": v.clone() for k, v in student.state_dict().items()})
        fx.update({f"s{step}.teacher.{k}": v.clone() for k, v in teacher.state_dict().items()})
    fx["hyper"] = np.array([lrs, wds, moms])
    fx["norm_names"] = np.array([n for n, p in student.named_parameters() if p.requires_grad])
    fx["membership_names"] = np.array(list(membership.keys()))
    fx["membership_reg"] = np.array(list(membership.values()))
    save("f5_lafs_step", **fx)

    # ---------------------------------------------------------------- F16 LAFS step x2 on the reference's real pair:
    # ViT_face_landmark_patch8 student / teacher (lafs_train.py:300-335) fed [B, n, 192] patch tokens (:538-569), rates 0
    print("F16 lafs step, part-fvit backbones")
    torch.manual_seed(16)
    K, B, ncrops = 256, 2, 4
    def mk_pf():
        m = ref_face.ViT_face_landmark_patch8(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8, dim=64,
                                              depth=2, heads=2, num_patches=196, mlp_dim=128, dropout=0.0, emb_dropout=0.0,
                                              with_land=False, use_standcoord=False, Random_prob=False, shuffle=False)
        for q in m.modules():
            if isinstance(q, ref_vit.DropPath):
                q.drop_prob = 0.0
        return m
    student = ref_utils.MultiCropWrapper(mk_pf(), ref_vit.DINOHead(64, K, hidden_dim=64, bottleneck_dim=32, norm_last_layer=True))
    teacher = ref_utils.MultiCropWrapper(mk_pf(), ref_vit.DINOHead(64, K, hidden_dim=64, bottleneck_dim=32))
    teacher.load_state_dict(student.state_dict())
    for p in teacher.parameters():
        p.requires_grad = False
    crit = ref_lafs.DINOLoss(K, ncrops, 0.07, 0.04, 3, 10)
    opt = torch.optim.AdamW(ref_utils.get_params_groups(student))
    fx = {"init." + k: v.clone() for k, v in student.state_dict().items()}
    lrs, wds, moms = [5e-4, 4e-4], [0.04, 0.05], [0.9, 0.95]
    for step in range(2):
        epoch = step
        toks = [torch.randn(B, 196, 192).clamp(-1, 1) for _ in range(2)] + [torch.randn(B, 36, 192).clamp(-1, 1) for _ in range(ncrops - 2)]
        for i, g in enumerate(opt.param_groups):
            g["lr"] = lrs[step]
            if i == 0:
                g["weight_decay"] = wds[step]
        t_out = teacher(toks[:2]); s_out = student(toks)
        loss = crit(s_out, t_out, epoch)
        opt.zero_grad()
        loss.backward()
        norms = ref_utils.clip_gradients(student, 3.0)
        post = {n: p.grad.clone() for n, p in student.named_parameters() if p.grad is not None}
        ref_utils.cancel_gradients_last_layer(epoch, student, 1)
        opt.step()
        with torch.no_grad():
            for pq, pk in zip(student.parameters(), teacher.parameters()):
                pk.data.mul_(moms[step]).add_((1 - moms[step]) * pq.detach().data)
        fx.update({f"s{step}.crop{i}": im for i, im in enumerate(toks)})
        fx.update({f"s{step}.loss": loss, f"s{step}.center": crit.center, f"s{step}.t_out": t_out, f"s{step}.s_out": s_out,
                   f"s{step}.norms": np.array(norms)})
        fx.update({f"s{step}.grad_post.{n}": g for n, g in post.items()})
        fx.update({f"s{step}.student.{k}": v.clone() for k, v in student.state_dict().items()})
        fx.update({f"s{step}.teacher.{k}": v.clone() for k, v in teacher.state_dict().items()})
    fx["hyper"] = np.array([lrs, wds, moms])
    fx["norm_names"] = np.array([n for n, p in student.named_parameters() if p.requires_grad])
    # what the stage-3 script loads from this checkpoint: ckpt['teacher'] keys with 'backbone.' stripped (train_largescale.py:639-657)
    fx["teacher_backbone_keys"] = np.array([k[len("backbone."):] for k in teacher.state_dict() if k.startswith("backbone.")])
    save("f16_lafs_step_partfvit", **fx)

    # ---------------------------------------------------------------- F17 one LAFS step at K = 8192 with DropPath LIVE in the
    # student (rates linspace(0, 0.3, 3), vision_transformer.py:150): the reference's own masks are recorded by observing the
    # torch.rand draws of drop_path (:30) during the student pass, so the HIP engine can be fed the same masks.  K >= 8192 puts
    # the loss at ~ln K, the regime of the benchmark's K = 100 000, and holds the 1e-3 loss bar against reference output.
    print("F17 lafs step, K = 8192, live DropPath")
    torch.manual_seed(17)
    K, B, ncrops, depth = 8192, 2, 5, 3
    mk = lambda dpr: ref_vit.VisionTransformer(img_size=[112], patch_size=8, embed_dim=64, depth=depth, num_heads=1, qkv_bias=True,
                                               drop_path_rate=dpr, norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6))
    student = ref_utils.MultiCropWrapper(mk(0.3), ref_vit.DINOHead(64, K, hidden_dim=128, bottleneck_dim=32, norm_last_layer=True))
    teacher = ref_utils.MultiCropWrapper(mk(0.0), ref_vit.DINOHead(64, K, hidden_dim=128, bottleneck_dim=32))
    teacher.load_state_dict(student.state_dict())
    for p in teacher.parameters():
        p.requires_grad = False
    crit = ref_lafs.DINOLoss(K, ncrops, 0.07, 0.04, 3, 10)
    crit.center.copy_(0.05 * torch.randn(1, K))                       # a non-trivial center, as in a run that is under way
    opt = torch.optim.AdamW(ref_utils.get_params_groups(student))
    fx = {"init." + k: v.clone() for k, v in student.state_dict().items()}
    fx["center0"] = crit.center.clone()
    lr, wd, mom, epoch = 5e-4, 0.04, 0.9, 1
    imgs = [torch.randn(B, 3, 112, 112).clamp(-1, 1) for _ in range(2)] + [torch.randn(B, 3, 48, 48).clamp(-1, 1) for _ in range(ncrops - 2)]
    for i, g in enumerate(opt.param_groups):
        g["lr"] = lr
        if i == 0:
            g["weight_decay"] = wd
    t_out = teacher(imgs[:2])
    draws, real_rand = [], torch.rand
    def observing_rand(*a, **k):
        r = real_rand(*a, **k)
        draws.append(r.clone())
        return r
    torch.rand = observing_rand
    try:
        s_out = student(imgs)
    finally:
        torch.rand = real_rand
    # draws arrive per backbone pass (global group: 2B sequences, then local group: 3B), per block with rate > 0, attention branch
    # then MLP branch (vision_transformer.py:107-113); block 0 has rate 0 and draws nothing
    rates = [x.item() for x in torch.linspace(0, 0.3, depth)]
    scales = []
    it = iter(draws)
    for n_seq in (2 * B, (ncrops - 2) * B):
        sc = torch.ones(depth, 2, n_seq)
        for l in range(depth):
            if rates[l] == 0.0:
                continue
            for br in range(2):
                r = next(it).view(-1)
                assert r.numel() == n_seq
                sc[l, br] = torch.floor(1 - rates[l] + r) / (1 - rates[l])
        scales.append(sc)
    assert next(it, None) is None
    assert any(float((sc == 0).sum()) > 0 for sc in scales), "seed drew no dropped path"
    loss = crit(s_out, t_out, epoch)
    opt.zero_grad()
    loss.backward()
    norms = ref_utils.clip_gradients(student, 3.0)
    post = {n: p.grad.clone() for n, p in student.named_parameters() if p.grad is not None}
    ref_utils.cancel_gradients_last_layer(epoch, student, 1)
    opt.step()
    with torch.no_grad():
        for pq, pk in zip(student.parameters(), teacher.parameters()):
            pk.data.mul_(mom).add_((1 - mom) * pq.detach().data)
    fx.update({f"crop{i}": im for i, im in enumerate(imgs)})
    fx.update({"loss": loss, "center": crit.center, "t_out": t_out, "s_out": s_out, "norms": np.array(norms),
               "scales_global": scales[0], "scales_local": scales[1], "rates": np.array(rates)})
    fx.update({f"grad_post.{n}": g for n, g in post.items()})
    fx.update({f"teacher.{k}": v.clone() for k, v in teacher.state_dict().items() if "last_layer" not in k})
    fx["hyper"] = np.array([lr, wd, mom])
    fx["norm_names"] = np.array([n for n, p in student.named_parameters() if p.requires_grad])
    save("f17_lafs_step_k8192_droppath", **fx)

    # ---------------------------------------------------------------- F12 param_groups_lrd (train_largescale.py:122-196)
    # train_largescale.py is a script with top-level side effects (argparse, NCCL init) and cannot be imported; its two grouping
    # functions are pure, so their source lines are exec'ed straight out of the reference file here (nothing of it is stored).
    print("F12 param_groups_lrd")
    src = open(os.path.join(REF, "train_largescale.py")).read().split("\n")
    def grab(fn):
        i = next(k for k, l in enumerate(src) if l.startswith(f"def {fn}("))
        j = next(k for k in range(i + 1, len(src)) if src[k].startswith(("def ", "class ")) or (src[k] and not src[k][0].isspace() and not src[k].startswith("#")))
        return "\n".join(src[i:j])
    ns = {}
    exec(grab("get_layer_id_for_vit"), ns); exec(grab("param_groups_lrd"), ns)
    torch.manual_seed(12)
    ft = ref_face.ViT_face_landmark_patch8(loss_type="CosFace", GPU_ID=None, num_class=50, image_size=112, patch_size=8, dim=64, depth=2,
                                           heads=2, mlp_dim=128, dropout=0.1, emb_dropout=0.1, with_land=True)
    groups = ns["param_groups_lrd"](ft, 1e-1, no_weight_decay_list=[], layer_decay=0.58)          # the call at :619-621
    by_id = {id(p): n for n, p in ft.named_parameters()}
    names, wds, lrs = [], [], []
    for gp in groups:
        for p in gp["params"]:
            names.append(by_id[id(p)]); wds.append(gp["weight_decay"]); lrs.append(gp["lr_scale"])
    assert sorted(names) == sorted(n for n, p in ft.named_parameters() if p.requires_grad)
    save("f12_param_groups_lrd", names=np.array(names), weight_decay=np.array(wds), lr_scale=np.array(lrs),
         ndim=np.array([dict(ft.named_parameters())[n].dim() for n in names]))

    # ---------------------------------------------------------------- F6 schedules
    print("F6 schedules")
    save("f6_schedules",
         lr=ref_utils.cosine_scheduler(5e-4 * 64 / 256, 1e-6, 6, 11, warmup_epochs=2),
         wd=ref_utils.cosine_scheduler(0.04, 0.4, 6, 11),
         mom=ref_utils.cosine_scheduler(0.996, 1, 6, 11))

    # ---------------------------------------------------------------- F7 Part-fViT
    print("F7 part-fvit")
    torch.manual_seed(7)
    pv = ref_face.ViT_face_landmark_patch8(loss_type="None", GPU_ID=None, num_class=10, image_size=112,
                                           patch_size=8, dim=128, depth=2, heads=3, mlp_dim=256, dropout=0.0,
                                           emb_dropout=0.0, with_land=False)
    for m in pv.modules():                          # Residual_droppath hard-codes rate 0.1 -> parity mode = 0
        if isinstance(m, ref_vit.DropPath):
            m.drop_prob = 0.0
    ximg = torch.randn(2, 3, 112, 112).clamp(-1, 1)
    xpat = torch.randn(3, 36, 192)                  # 3-D patch input (local views, 36 landmarks)
    w1, w2 = torch.randn(2, 128), torch.randn(3, 128)
    e1, e2 = pv(ximg), pv(xpat)
    ((e1 * w1).sum() + (e2 * w2).sum()).backward()
    save("f7_partfvit", ximg=ximg, xpat=xpat, w1=w1, w2=w2, e1=e1, e2=e2, **sd(pv), **grads(pv))

    # ---------------------------------------------------------------- F20 Part-fViT forward options: use_standcoord (+ Random_prob, shuffle), save_token
    print("F20 part-fvit standcoord / save_token")
    torch.manual_seed(20)
    kw = dict(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8, dim=128, depth=2, heads=3, mlp_dim=256,
              dropout=0.0, emb_dropout=0.0, with_land=False, use_standcoord=True)
    ps = ref_face.ViT_face_landmark_patch8(**kw)
    for m in ps.modules():
        if isinstance(m, ref_vit.DropPath):
            m.drop_prob = 0.0
    with torch.no_grad():                           # (N(0, 1) position / cls tables would drown the patches: keep the output sensitive to WHERE they are gathered)
        ps.pos_embedding.mul_(0.05); ps.cls_token.mul_(0.05)
    ps.eval()
    xs = torch.randn(2, 3, 112, 112).clamp(-1, 1)
    with torch.no_grad():
        e_plain, tok_plain, _ = ps(xs, save_token=True)
        ps.Random_prob, ps.shuffle = True, True
        torch.manual_seed(2020)                     # the forward draws torch.randn(theta.shape) and then torch.randint(0, c, (b, c, 1))
        e_rand = ps(xs)
        torch.manual_seed(2020)
        noise = torch.randn(2, 196, 2); ids = torch.randint(0, 196, (2, 196, 1))
    save("f20_partfvit_standcoord", x=xs, e_plain=e_plain, tok_plain=tok_plain, e_rand=e_rand, noise=noise, ids=ids, **sd(ps))

    # ---------------------------------------------------------------- F8 landmark patch gather
    print("F8 gather")
    torch.manual_seed(8)
    for n in (196, 36):
        img = torch.randn(2, 3, 112, 112, requires_grad=True)
        th = (torch.rand(2, n, 2) * 130 - 10).requires_grad_(True)      # includes out-of-image landmarks
        out = ref_face.extract_patches_pytorch_gridsample(img, th, patch_shape=torch.tensor([8, 8]), num_landm=n)
        w = torch.randn_like(out)
        (out * w).sum().backward()
        save(f"f8_gather_n{n}", img=img, theta=th, w=w, out=out, gtheta=th.grad, gimg=img.grad)

    # ---------------------------------------------------------------- F9 landmark CNN wrapper (frozen, eval)
    # 2.8 M parameters are too many to ship: both sides fill every tensor of the (identical) state_dict with the same
    # closed-form pattern (tests/conftest.det_fill), so only inputs and outputs are stored.
    print("F9 landmark cnn")
    sys.path.insert(0, os.path.join(os.path.dirname(OUT)))
    from conftest import det_fill
    torch.manual_seed(9)
    lc = ref_face.face_landmark_4simmin_glo_loc(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8,
                                                dim=64, depth=1, heads=1, mlp_dim=64)
    det_fill(lc)
    lc.eval()
    x9 = torch.randn(2, 3, 112, 112).clamp(-1, 1); xa9 = torch.randn(2, 3, 112, 112).clamp(-1, 1)
    with torch.no_grad():
        th_a, mo_a = lc(x9, x_Aug=xa9, Random_prob=False)
        torch.manual_seed(123); th_b, mo_b = lc(x9, x_Aug=xa9, Random_prob=True, return_prob=True, random_coor=False)
        torch.manual_seed(124); th_c, mo_c = lc(x9, x_Aug=xa9, Random_prob=True, ran_sample=True, random_coor=False)
    save("f9_landmark_cnn", x=x9, x_aug=xa9, theta_plain=th_a, mosaic_plain=mo_a, theta_jitter=th_b, mosaic_jitter=mo_b,
         theta_local=th_c, mosaic_local=mo_c, keys=np.array(sorted(lc.state_dict().keys())))

    # ---------------------------------------------------------------- F10 CosFace
    print("F10 cosface")
    torch.manual_seed(10)
    cf = ref_face.CosFace(64, 300, None, s=64.0, m=0.4)
    x = torch.randn(4, 64, requires_grad=True)
    y = torch.tensor([3, 299, 0, 17])
    o_hard = cf(x, y)
    wcf = torch.randn(4, 300)
    (o_hard * wcf).sum().backward()
    gx_h, gw_h = x.grad.clone(), cf.weight.grad.clone()
    x.grad = None; cf.weight.grad = None
    ysoft = 0.3 * torch.nn.functional.one_hot(y, 300).float() + 0.7 * torch.nn.functional.one_hot(y.flip(0), 300).float()
    o_soft = cf(x, ysoft)
    ce = torch.sum(-ysoft * torch.nn.functional.log_softmax(o_soft, dim=-1), dim=-1).mean()
    ce.backward()
    save("f10_cosface", x=x, y=y, weight=cf.weight, w=wcf, out_hard=o_hard, gx_hard=gx_h, gw_hard=gw_h,
         ysoft=ysoft, out_soft=o_soft, ce_soft=ce, gx_soft=x.grad, gw_soft=cf.weight.grad)

    # ---------------------------------------------------------------- F11 Mixup (batch mode)
    print("F11 mixup")
    mix = ref_mix.Mixup(mixup_alpha=0.2, cutmix_alpha=0.0, cutmix_minmax=None, prob=1.0, switch_prob=0.5,
                        mode="batch", label_smoothing=0.0, num_classes=50)
    np.random.seed(11)
    xm = torch.randn(4, 3, 16, 16)
    ym = torch.tensor([1, 7, 7, 49])
    x_in = xm.clone()
    xo, yo = mix(xm, ym, device="cpu")
    np.random.seed(11)
    np.random.rand(); lam = float(np.random.beta(0.2, 0.2))
    save("f11_mixup", x_in=x_in, y=ym, x_out=xo, target=yo, lam=np.float64(lam))
    # ---------------------------------------------------------------- F13 Part-fViT with the trainable landmark branch
    # (train_largescale.py:432,556: with_land=True).  eval mode (BatchNorm running statistics, Dropout/DropPath off) so the
    # pass is deterministic; gradients reach the CNN through theta.
    print("F13 part-fvit with_land")
    torch.manual_seed(13)
    pl = ref_face.ViT_face_landmark_patch8(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8,
                                           dim=128, depth=2, heads=3, mlp_dim=256, dropout=0.0, emb_dropout=0.0,
                                           with_land=True)
    det_fill(pl.stn); det_fill(pl.output_layer)     # the 2.8 M-parameter CNN: closed form; the trunk: seeded init, stored
    pl.eval()
    x13 = torch.randn(2, 3, 112, 112).clamp(-1, 1)
    w13 = torch.randn(2, 128)
    e13 = pl(x13)
    (e13 * w13).sum().backward()
    g13 = {k: p.grad for k, p in pl.named_parameters() if p.grad is not None}
    keep = ["output_layer.1.weight", "output_layer.1.bias", "stn.features.0.0.weight", "stn.features.15.conv.7.weight",
            "patch_to_embedding.weight", "pos_embedding", "transformer.layers.0.0.fn.fn.to_qkv.weight"]
    save("f13_partfvit_land", x=x13, w=w13, e=e13, theta=pl.theta,
         gnorm_keys=np.array(sorted(g13.keys())), gnorms=np.array([float(g13[k].norm()) for k in sorted(g13.keys())]),
         **{"g." + k: g13[k] for k in keep},
         **{"p." + k: v for k, v in pl.state_dict().items() if not k.startswith(("stn.", "output_layer."))})
    # ---------------------------------------------------------------- F18 the landmark branch in TRAIN mode (what train_largescale.py
    # really runs, :432 + model.train()): MobileNetV3 trunk with BatchNorm batch statistics, Dropout(0.5) in front of the regressor
    # (mask recorded), min-max scaling -- exactly the lines ViT_face.py:679-706 -- and the backward from a given d(loss)/d(theta).
    # Pins nn.BatchNorm2d's training semantics (biased variance in the normalisation, momentum update of the running statistics
    # with the unbiased one) and the gradient paths through min / max for the HIP training plan of the CNN.  Weights: det_fill_random
    # (pseudo-random by key: det_fill's sinusoids make the convolutions cancel, which batch-statistics BatchNorm turns into noise).
    print("F18 landmark branch, train mode")
    torch.manual_seed(18)
    pt = ref_face.ViT_face_landmark_patch8(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8,
                                           dim=64, depth=1, heads=1, mlp_dim=64, dropout=0.0, emb_dropout=0.0, with_land=True)
    from conftest import det_fill_random
    det_fill_random(pt.stn); det_fill_random(pt.output_layer)
    pt.train()
    x18 = torch.randn(4, 3, 112, 112).clamp(-1, 1)
    rec18 = []
    def recording_dropout18(inp, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return inp
        keep = torch.rand_like(inp) >= p
        rec18.append(keep)
        return inp * keep / (1.0 - p)
    orig_dropout18 = torch.nn.functional.dropout
    torch.nn.functional.dropout = recording_dropout18
    try:
        t18 = pt.stn(x18).mean(dim=(-2, -1))
        t18 = pt.output_layer(t18)
    finally:
        torch.nn.functional.dropout = orig_dropout18
    assert len(rec18) == 1 and rec18[0].shape == (4, 160)
    tmax = torch.max(t18, 1)[0].unsqueeze(1).repeat(1, 392)
    tmin = torch.min(t18, 1)[0].unsqueeze(1).repeat(1, 392)
    th18 = ((t18 - tmin) / (tmax - tmin) * 111).view(-1, 196, 2)
    dth18 = torch.randn(4, 196, 2) * 0.05
    (th18 * dth18).sum().backward()
    g18 = {k: p.grad for k, p in pt.named_parameters() if p.grad is not None and k.startswith(("stn.", "output_layer."))}
    keep18 = ["output_layer.1.weight", "output_layer.1.bias", "stn.features.0.0.weight", "stn.features.0.1.weight", "stn.features.1.conv.3.weight",
              "stn.features.4.conv.5.fc.0.weight", "stn.features.4.conv.5.fc.2.weight", "stn.features.4.conv.4.bias", "stn.features.7.conv.0.weight",
              "stn.features.13.conv.3.weight", "stn.features.15.conv.7.weight", "stn.features.15.conv.8.weight"]
    bn18 = ["stn.features.0.1", "stn.features.4.conv.4", "stn.features.15.conv.8"]
    sd18 = pt.state_dict()
    save("f18_landmark_train", x=x18, drop_keep=rec18[0], t=t18, theta=th18, dtheta=dth18,
         gnorm_keys=np.array(sorted(g18.keys())), gnorms=np.array([float(g18[k].norm()) for k in sorted(g18.keys())]),
         **{"g." + k: g18[k] for k in keep18},
         **{"rm." + k: sd18[k + ".running_mean"] for k in bn18}, **{"rv." + k: sd18[k + ".running_var"] for k in bn18},
         nbt=sd18["stn.features.0.1.num_batches_tracked"])
    # ---------------------------------------------------------------- F14 Part-fViT element dropout (train mode)
    # nn.Dropout sites of the reference (emb :614,768; to_out :150-153; after GELU and after fc2 :126-133) with the masks
    # captured: F.dropout is replaced by a recording implementation for this one forward, DropPath forced to 0.
    print("F14 part-fvit dropout sites")
    torch.manual_seed(14)
    pd_ = ref_face.ViT_face_landmark_patch8(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8,
                                            dim=128, depth=2, heads=3, mlp_dim=256, dropout=0.1, emb_dropout=0.1,
                                            with_land=False)
    for m in pd_.modules():
        if isinstance(m, ref_vit.DropPath):
            m.drop_prob = 0.0
    pd_.train()
    rec = []
    orig_dropout = torch.nn.functional.dropout

    def recording_dropout(inp, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return inp
        keep = torch.rand_like(inp) >= p
        rec.append(keep)
        return inp * keep / (1.0 - p)

    torch.nn.functional.dropout = recording_dropout
    try:
        x14 = torch.randn(2, 3, 112, 112).clamp(-1, 1)
        w14 = torch.randn(2, 128)
        e14 = pd_(x14)
        (e14 * w14).sum().backward()
    finally:
        torch.nn.functional.dropout = orig_dropout
    assert len(rec) == 1 + 3 * 2, len(rec)
    save("f14_partfvit_dropout", x=x14, w=w14, e=e14, p=np.float32(0.1),
         **{f"keep{i}": k.to(torch.uint8) for i, k in enumerate(rec)}, **sd(pd_), **grads(pd_))
    # ---------------------------------------------------------------- F15 checkpoint layout at full scale (keys + shapes only)
    # What lafs_train.py saves (:451-460) and train_largescale.py loads (:639-661): the state_dict manifests of the real
    # configurations, so that the drop-in's checkpoint compatibility is pinned on names AND shapes.
    print("F15 checkpoint manifests")
    torch.manual_seed(15)
    man = lambda m: {k: list(v.shape) for k, v in m.state_dict().items()}
    vs = ref_vit.__dict__["vit_small"](patch_size=8, drop_path_rate=0.1)
    stu = ref_utils.MultiCropWrapper(vs, ref_vit.DINOHead(384, 100000, use_bn=False, norm_last_layer=True))
    tea = ref_utils.MultiCropWrapper(ref_vit.__dict__["vit_small"](patch_size=8), ref_vit.DINOHead(384, 100000, False))
    dl = ref_lafs.DINOLoss(100000, 10, 0.07, 0.04, 30, 41)
    ft = ref_face.ViT_face_landmark_patch8(loss_type="CosFace", GPU_ID=None, num_class=1000, image_size=112, patch_size=8, dim=768,
                                           depth=12, heads=11, mlp_dim=2048, dropout=0.1, emb_dropout=0.1, with_land=True)
    lm = ref_face.face_landmark_4simmin_glo_loc(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8, dim=768,
                                                depth=12, heads=11, mlp_dim=2048)
    import json as _json
    save("f15_checkpoint_manifests", manifest=np.array(_json.dumps({
        "student": man(stu), "teacher": man(tea), "dino_loss": man(dl), "finetune_backbone": man(ft), "landmark_cnn": man(lm)})))
    print("done")


if __name__ == "__main__":
    main()
