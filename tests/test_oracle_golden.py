"""Pin the CPU oracle to golden vectors produced by the reference itself (tools/make_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub
from oracle import dino, gather, margin, optim, partfvit, step, vit

RT, AT = 2e-5, 2e-6


def close(a, b, rtol=RT, atol=AT):
    """Tensor-scale comparison: max|a-b| <= rtol * max|b| + atol (fp32 summation-order noise
    is proportional to the tensor's scale, not to each element)."""
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a.double() - b.double()).abs().max().item()
    tol = rtol * b.double().abs().max().item() + atol
    assert err <= tol, f"max abs err {err:.3e} > tol {tol:.3e}"


def close_adam(a, b, lr, frac=0.34):   # a whole key-bias third of qkv.bias may be ill-conditioned
    err = (a.double() - b.double()).abs()
    tol = 1e-4 * b.double().abs().max().item() + 2e-6
    assert err.max().item() <= 2.5 * lr, err.max().item()
    assert (err > tol).double().mean().item() <= frac, (err > tol).double().mean().item()


def test_f1_vit_forward_backward_and_pos_interp():
    fx = load_golden("f1_vit")
    cfg = vit.ViTConfig(patch_size=8, embed_dim=128, depth=2, num_heads=2, img_size=224)
    P = {k: v.clone().requires_grad_(True) for k, v in sub(fx, "p.").items()}
    close(vit.interp_pos_embed(P["pos_embed"], 196, 112, 112, 8).detach(), fx["pos14"])
    close(vit.interp_pos_embed(P["pos_embed"], 36, 48, 48, 8).detach(), fx["pos6"])
    og, ol = vit.vit_forward(P, fx["xg"], cfg), vit.vit_forward(P, fx["xl"], cfg)
    close(og.detach(), fx["og"], 1e-4, 1e-5)
    close(ol.detach(), fx["ol"], 1e-4, 1e-5)
    ((og * fx["wg"]).sum() + (ol * fx["wl"]).sum()).backward()
    G = sub(fx, "g.")
    assert set(G) == set(P)
    for k, g in G.items():
        close(P[k].grad, g, 1e-3, 1e-5)


@pytest.mark.parametrize("name", ["f2_head", "f2_head_freeg"])
def test_f2_dino_head(name):
    fx = load_golden(name)
    frozen_g = name == "f2_head"
    P = {k: v.clone().requires_grad_(not (frozen_g and k.endswith("weight_g"))) for k, v in sub(fx, "p.").items()}
    x = fx["x"].clone().requires_grad_(True)
    out = vit.dino_head_forward(P, x)
    close(out.detach(), fx["out"], 1e-4, 1e-6)
    (out * fx["w"]).sum().backward()
    close(x.grad, fx["gx"], 1e-3, 1e-6)
    G = sub(fx, "g.")
    assert set(G) == {k for k, v in P.items() if v.requires_grad}
    for k, g in G.items():
        close(P[k].grad, g, 1e-3, 1e-6)


def test_f3_multicrop_grouping_and_order():
    fx = load_golden("f3_multicrop")
    crops = [fx[f"crop{i}"] for i in range(5)]
    assert vit.crop_groups(crops) == [2, 5]
    cfg = vit.ViTConfig(patch_size=8, embed_dim=64, depth=1, num_heads=1, img_size=112)
    out = vit.multicrop_forward(sub(fx, "p.backbone."), sub(fx, "p.head."), crops, cfg)
    close(out, fx["out"], 1e-4, 1e-6)


@pytest.mark.parametrize("ncrops", [4, 10])
def test_f4_dino_loss(ncrops):
    fx = load_golden(f"f4_dinoloss_nc{ncrops}")
    sched = dino.teacher_temp_schedule(0.07, 0.04, 6, 10)
    np.testing.assert_allclose(sched, fx["schedule"].numpy(), rtol=0, atol=0)
    for e in (0, 7):
        s = fx[f"e{e}_student"].clone().requires_grad_(True)
        t, c0 = fx[f"e{e}_teacher"], fx[f"e{e}_center_before"]
        assert float(fx[f"e{e}_temp"]) == sched[e]
        loss = dino.dino_loss(s, t, c0, ncrops, sched[e])
        loss.backward()
        close(loss.detach(), fx[f"e{e}_loss"], 1e-5, 1e-6)
        close(s.grad, fx[f"e{e}_grad"], 1e-4, 1e-9)
        l2, g2 = dino.dino_loss_closed_form(s.detach(), t, c0, ncrops, sched[e])
        close(l2, fx[f"e{e}_loss"], 1e-5, 1e-6)
        close(g2, fx[f"e{e}_grad"], 1e-4, 1e-9)
        close(dino.update_center(c0, t), fx[f"e{e}_center_after"], 1e-5, 1e-7)


def test_f5_full_lafs_step_two_iterations():
    fx = load_golden("f5_lafs_step")
    cfg = vit.ViTConfig(patch_size=8, embed_dim=64, depth=2, num_heads=1, img_size=112)
    st = step.LafsState(cfg, out_dim=512, seed=0, hidden_dim=128, bottleneck_dim=64)
    init = sub(fx, "init.")
    assert set(init) == set(st.student)
    st.student = {k: v.clone() for k, v in init.items()}
    st.teacher = {k: v.clone() for k, v in init.items()}
    st.exp_avg = {k: torch.zeros_like(v) for k, v in init.items()}
    st.exp_avg_sq = {k: torch.zeros_like(v) for k, v in init.items()}
    st.steps = {k: 0 for k in init}
    # weight-decay membership (utils.get_params_groups)
    for n, reg in zip(fx["membership_names"], fx["membership_reg"].tolist()):
        assert optim.is_regularized(str(n), init[str(n)].shape) == bool(reg), n
    lrs, wds, moms = fx["hyper"].tolist()
    tt = dino.teacher_temp_schedule(0.07, 0.04, 3, 10)
    for s in range(2):
        crops = [fx[f"s{s}.crop{i}"] for i in range(5)]
        r = step.lafs_step(st, crops, epoch=s, lr=lrs[s], wd=wds[s], momentum=moms[s], teacher_temp=tt[s],
                           clip_grad=3.0, freeze_last_layer=1)
        close(r["loss"], fx[f"s{s}.loss"], 1e-5, 1e-6)
        close(r["teacher_out"], fx[f"s{s}.t_out"], 1e-4, 1e-6)
        close(r["student_out"], fx[f"s{s}.s_out"], 1e-4, 1e-6)
        names = [str(n) for n in fx["norm_names"]]
        np.testing.assert_allclose([r["norms"][n] for n in names], fx[f"s{s}.norms"].numpy(), rtol=2e-3, atol=1e-8)
        post = sub(fx, f"s{s}.grad_post.")
        for k, g in post.items():
            if s == 0 and "last_layer" in k:
                assert k not in r["grads"]          # cancelled while frozen
                continue
            close(r["grads"][k], g, 2e-3, 1e-7)
        # Adam's m/(sqrt(v)+eps) is ill-conditioned where the gradient is at round-off level
        # (e.g. the key bias, whose true gradient is 0): those elements may move by up to +-lr.
        for k, v in sub(fx, f"s{s}.student.").items():
            close_adam(st.student[k], v, lrs[s])
        for k, v in sub(fx, f"s{s}.teacher.").items():
            close_adam(st.teacher[k], v, lrs[s])
        close(st.center, fx[f"s{s}.center"], 1e-4, 1e-7)


def test_f16_lafs_step_on_partfvit_backbones():
    """The reference's real `mynet` pair: ViT_face_landmark_patch8 student / teacher on [B, n, 192] patch tokens."""
    from oracle import partfvit
    fx = load_golden("f16_lafs_step_partfvit")
    cfg = partfvit.PartFViTConfig(dim=64, depth=2, heads=2, mlp_dim=128, num_patches=196)
    st = step.LafsState(cfg, out_dim=256, seed=0, hidden_dim=64, bottleneck_dim=32)
    init = sub(fx, "init.")
    assert set(init) == set(st.student)
    st.student = {k: v.clone() for k, v in init.items()}
    st.teacher = {k: v.clone() for k, v in init.items()}
    st.exp_avg = {k: torch.zeros_like(v) for k, v in init.items()}
    st.exp_avg_sq = {k: torch.zeros_like(v) for k, v in init.items()}
    st.steps = {k: 0 for k in init}
    lrs, wds, moms = fx["hyper"].tolist()
    tt = dino.teacher_temp_schedule(0.07, 0.04, 3, 10)
    for s in range(2):
        crops = [fx[f"s{s}.crop{i}"] for i in range(4)]
        r = step.lafs_step(st, crops, epoch=s, lr=lrs[s], wd=wds[s], momentum=moms[s], teacher_temp=tt[s], clip_grad=3.0,
                           freeze_last_layer=1)
        close(r["loss"], fx[f"s{s}.loss"], 1e-5, 1e-6)
        close(r["teacher_out"], fx[f"s{s}.t_out"], 1e-4, 1e-6)
        close(r["student_out"], fx[f"s{s}.s_out"], 1e-4, 1e-6)
        for k, g in sub(fx, f"s{s}.grad_post.").items():
            if s == 0 and "last_layer" in k:
                assert k not in r["grads"]
                continue
            close(r["grads"][k], g, 2e-3, 1e-7)
        for k, v in sub(fx, f"s{s}.student.").items():
            close_adam(st.student[k], v, lrs[s])
        for k, v in sub(fx, f"s{s}.teacher.").items():
            close_adam(st.teacher[k], v, lrs[s])
        close(st.center, fx[f"s{s}.center"], 1e-4, 1e-7)


def test_f17_lafs_step_k8192_with_the_reference_droppath_masks():
    """One reference LAFS step at K = 8192 (loss ~ ln K, as at the benchmark's K = 100 000) with DropPath live in the student:
    the oracle is fed the masks the reference itself drew (vision_transformer.py:27-35) and must reproduce loss, outputs, clipped
    gradients, teacher EMA and center -- this pins the oracle's DropPath forward/backward scaling to the reference."""
    fx = load_golden("f17_lafs_step_k8192_droppath")
    cfg = vit.ViTConfig(patch_size=8, embed_dim=64, depth=3, num_heads=1, img_size=112)
    st = step.LafsState(cfg, out_dim=8192, seed=0, hidden_dim=128, bottleneck_dim=32)
    init = sub(fx, "init.")
    assert set(init) == set(st.student)
    st.student = {k: v.clone() for k, v in init.items()}
    st.teacher = {k: v.clone() for k, v in init.items()}
    st.exp_avg = {k: torch.zeros_like(v) for k, v in init.items()}
    st.exp_avg_sq = {k: torch.zeros_like(v) for k, v in init.items()}
    st.steps = {k: 0 for k in init}
    st.center = fx["center0"].clone()
    lr, wd, mom = fx["hyper"].tolist()
    scales = [fx["scales_global"], fx["scales_local"]]
    assert sum(int((s == 0).sum()) for s in scales) > 0                      # the fixture does drop paths
    crops = [fx[f"crop{i}"] for i in range(5)]
    tt = dino.teacher_temp_schedule(0.07, 0.04, 3, 10)
    r = step.lafs_step(st, crops, epoch=1, lr=lr, wd=wd, momentum=mom, teacher_temp=tt[1], clip_grad=3.0, freeze_last_layer=1,
                       drop_scales=scales)
    close(r["loss"], fx["loss"], 1e-5, 1e-6)
    close(r["teacher_out"], fx["t_out"], 1e-4, 1e-6)
    close(r["student_out"], fx["s_out"], 1e-4, 1e-6)
    names = [str(n) for n in fx["norm_names"]]
    np.testing.assert_allclose([r["norms"][n] for n in names], fx["norms"].numpy(), rtol=2e-3, atol=1e-8)
    for k, g in sub(fx, "grad_post.").items():
        close(r["grads"][k], g, 2e-3, 1e-7)
    for k, v in sub(fx, "teacher.").items():
        close_adam(st.teacher[k], v, lr)
    close(st.center, fx["center"], 1e-4, 1e-7)
    # without the masks the result must differ (the test would be vacuous otherwise)
    st2 = step.LafsState(cfg, out_dim=8192, seed=0, hidden_dim=128, bottleneck_dim=32)
    st2.student = {k: v.clone() for k, v in init.items()}; st2.teacher = {k: v.clone() for k, v in init.items()}
    st2.center = fx["center0"].clone()
    r2 = step.lafs_step(st2, crops, epoch=1, lr=lr, wd=wd, momentum=mom, teacher_temp=tt[1])
    assert abs(float(r2["loss"]) - float(fx["loss"])) > 1e-4 * float(fx["loss"])


def test_f6_schedules():
    fx = load_golden("f6_schedules")
    np.testing.assert_allclose(optim.cosine_scheduler(5e-4 * 64 / 256, 1e-6, 6, 11, warmup_epochs=2), fx["lr"].numpy(), rtol=1e-12)
    np.testing.assert_allclose(optim.cosine_scheduler(0.04, 0.4, 6, 11), fx["wd"].numpy(), rtol=1e-12)
    np.testing.assert_allclose(optim.cosine_scheduler(0.996, 1, 6, 11), fx["mom"].numpy(), rtol=1e-12)


def test_f7_partfvit_forward_backward():
    fx = load_golden("f7_partfvit")
    cfg = partfvit.PartFViTConfig(patch_size=8, dim=128, depth=2, heads=3, mlp_dim=256, num_patches=196)
    P = {k: v.clone().requires_grad_(True) for k, v in sub(fx, "p.").items()}
    e1 = partfvit.forward_embedding(P, fx["ximg"], cfg)
    e2 = partfvit.forward_embedding(P, fx["xpat"], cfg)
    close(e1.detach(), fx["e1"], 1e-4, 1e-5)
    close(e2.detach(), fx["e2"], 1e-4, 1e-5)
    ((e1 * fx["w1"]).sum() + (e2 * fx["w2"]).sum()).backward()
    G = sub(fx, "g.")
    for k, g in G.items():
        close(P[k].grad, g, 2e-3, 2e-5)


def test_f20_partfvit_standard_coordinates_and_token_dump():
    """F20: ViT_face_landmark_patch8(use_standcoord=True) of the reference (ViT_face.py:717-742) -- patches gathered at the centres of
    the regular grid, mosaic transposed -- with `save_token` (the patch tokens in front of the head's LayerNorm, :769-770), and the
    same with Random_prob + shuffle (the reference's own torch.randn / torch.randint draws are in the fixture)."""
    fx = load_golden("f20_partfvit_standcoord")
    cfg = partfvit.PartFViTConfig(patch_size=8, dim=128, depth=2, heads=3, mlp_dim=256, num_patches=196)
    P = sub(fx, "p.")
    x = fx["x"]

    def run(noise=None, ids=None):
        th = partfvit.standard_coords(196, noise=noise, shuffle_id=ids, batch=x.shape[0])
        mosaic = gather.extract_patches(x, th, 8).permute(0, 1, 3, 2)
        return partfvit.forward_embedding(P, mosaic, cfg, return_tokens=True)
    e, tok = run()
    close(e, fx["e_plain"], 1e-4, 1e-5)
    close(tok, fx["tok_plain"], 1e-4, 1e-5)
    e2, _ = run(fx["noise"], fx["ids"].view(2, 196))
    close(e2, fx["e_rand"], 1e-4, 1e-5)


@pytest.mark.parametrize("n", [196, 36])
def test_f8_landmark_patch_gather(n):
    fx = load_golden(f"f8_gather_n{n}")
    img = fx["img"].clone().requires_grad_(True)
    th = fx["theta"].clone().requires_grad_(True)
    out = gather.extract_patches(img, th, 8)
    close(out.detach(), fx["out"], 1e-4, 1e-5)
    (out * fx["w"]).sum().backward()
    close(th.grad, fx["gtheta"], 1e-3, 1e-4)
    close(img.grad, fx["gimg"], 1e-4, 1e-5)


def test_f10_cosface_hard_and_soft_labels():
    fx = load_golden("f10_cosface")
    x = fx["x"].clone().requires_grad_(True)
    W = fx["weight"].clone().requires_grad_(True)
    out = margin.cosface_logits(x, W, fx["y"])
    close(out.detach(), fx["out_hard"], 1e-5, 1e-5)
    (out * fx["w"]).sum().backward()
    close(x.grad, fx["gx_hard"], 1e-4, 1e-5)
    close(W.grad, fx["gw_hard"], 1e-4, 1e-5)
    x.grad = None; W.grad = None
    out = margin.cosface_logits(x, W, fx["ysoft"])
    close(out.detach(), fx["out_soft"], 1e-5, 1e-5)
    ce = margin.soft_target_cross_entropy(out, fx["ysoft"])
    close(ce.detach(), fx["ce_soft"], 1e-5, 1e-6)
    ce.backward()
    close(x.grad, fx["gx_soft"], 1e-4, 1e-6)
    close(W.grad, fx["gw_soft"], 1e-4, 1e-6)


def test_f11_mixup_batch_mode():
    fx = load_golden("f11_mixup")
    rng = np.random.RandomState(11)
    lam = margin.draw_mixup_lambda(rng, 0.2, prob=1.0)
    assert lam == float(fx["lam"])
    x, tgt = margin.mixup_batch(fx["x_in"].clone(), fx["y"], 50, lam)
    close(x, fx["x_out"], 1e-6, 1e-7)
    close(tgt, fx["target"], 1e-6, 1e-7)


def test_f13_oracle_gather_plus_partfvit_matches_reference_landmark_branch():
    """with_land=True forward of the reference (CNN -> theta -> gather -> Part-fViT) from its own theta: the oracle's
    gather + trunk reproduce the embedding (the CNN itself is stock PyTorch on both sides)."""
    from oracle import gather, partfvit
    fx = load_golden("f13_partfvit_land")
    cfg = partfvit.PartFViTConfig(patch_size=8, dim=128, depth=2, heads=3, mlp_dim=256, num_patches=196)
    mosaic = gather.extract_patches(fx["x"], fx["theta"])
    e = partfvit.forward_embedding(sub(fx, "p."), mosaic, cfg)
    torch.testing.assert_close(e, fx["e"], rtol=1e-4, atol=1e-5)


def test_f14_oracle_dropout_sites_match_reference():
    """Element-dropout sites of Part-fViT in train mode (embedding, to_out, after GELU, after fc2): the oracle fed with the
    masks the reference drew reproduces its embedding and gradients."""
    from oracle import partfvit
    fx = load_golden("f14_partfvit_dropout")
    cfg = partfvit.PartFViTConfig(patch_size=8, dim=128, depth=2, heads=3, mlp_dim=256, num_patches=196)
    P = {k: v.clone().requires_grad_(True) for k, v in sub(fx, "p.").items()}
    sc = 1.0 / (1.0 - float(fx["p"]))
    keeps = [fx[f"keep{i}"].float() * sc for i in range(7)]
    masks = {"emb": keeps[0]}
    for l in range(2):
        masks[(l, 0)], masks[(l, 1)], masks[(l, 2)] = keeps[1 + 3 * l], keeps[2 + 3 * l], keeps[3 + 3 * l]
    e = partfvit.forward_embedding(P, fx["x"], cfg, masks=masks)
    torch.testing.assert_close(e, fx["e"], rtol=1e-4, atol=1e-5)
    (e * fx["w"]).sum().backward()
    for k, g in sub(fx, "g.").items():
        close(P[k].grad, g, 2e-4)
