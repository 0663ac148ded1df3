"""GPU parity of the fine-tune path pieces (Part-fViT, CosFace, landmark gather, mixup) against the reference's golden vectors."""
import math
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
# per-tensor gradient gates = 2x the worst value observed on MI355X (conftest.gate_errors prints the observed value)
GATE_F7, GATE_FT, GATE_F13, GATE_F14 = 1.1e-2, 1.5e-2, 3.5e-2, 1.1e-2     # observed 5.2e-3, 7.1e-3, 1.7e-2, 5.4e-3
GATE_WINDOW = 3.0e-2                                                       # observed 1.5e-2 (three accumulated micro-steps, two of them mixed)

from conftest import gate_errors, load_golden, sub  # noqa: E402
from lafs_cvpr2024_amd.face_pre_pro.ViT_face import CosFace, ViT_face_landmark_patch8, extract_patches_pytorch_gridsample  # noqa: E402
from lafs_cvpr2024_amd.ops import _p, call  # noqa: E402
from lafs_cvpr2024_amd.vision_transformer import attach_arena  # noqa: E402

DEV = "cuda"


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def test_f20_partfvit_standard_coordinates_and_token_dump():
    """F20 on the HIP path: `use_standcoord` (regular-grid landmarks, transposed mosaic; with Random_prob + shuffle under the
    reference's own random draws: the module draws from the CPU generator in the reference's order) and `save_token`
    (reference ViT_face.py:717-742, 769-770, 778-779)."""
    from lafs_cvpr2024_amd.face_pre_pro.ViT_face import ViT_face_landmark_patch8
    fx = load_golden("f20_partfvit_standcoord")
    m = ViT_face_landmark_patch8(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8, dim=128, depth=2, heads=3,
                                 mlp_dim=256, dropout=0.0, emb_dropout=0.0, with_land=False, use_standcoord=True, drop_path_rate=0.0)
    m.load_state_dict(sub(fx, "p."))
    m.eval()
    x = fx["x"].to(DEV)
    with torch.no_grad():
        e, tok, _ = m(x, save_token=True)
        print(f"[F20] emb {rel_l2(e, fx['e_plain']):.2e}, tokens {rel_l2(tok, fx['tok_plain']):.2e}; plain vs jittered + re-drawn in the reference: "
              f"{rel_l2(fx['e_rand'], fx['e_plain']):.2e}")
        assert rel_l2(e, fx["e_plain"]) < 2e-2 and rel_l2(tok, fx["tok_plain"]) < 2e-2, (rel_l2(e, fx["e_plain"]), rel_l2(tok, fx["tok_plain"]))
        m.Random_prob, m.shuffle = True, True
        torch.manual_seed(2020)                   # the reference's draws: torch.randn(theta.shape), then torch.randint(0, c, (b, c, 1))
        e2 = m(x)
        assert rel_l2(e2, fx["e_rand"]) < 2e-2, rel_l2(e2, fx["e_rand"])
        assert rel_l2(e2, fx["e_plain"]) > 5 * rel_l2(e2, fx["e_rand"])      # (the jittered, re-drawn landmarks give ANOTHER embedding: that one)


def test_f7_partfvit_forward_backward():
    fx = load_golden("f7_partfvit")
    m = ViT_face_landmark_patch8(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8, dim=128, depth=2,
                                 heads=3, mlp_dim=256, dropout=0.0, emb_dropout=0.0, with_land=False, drop_path_rate=0.0)
    assert set(m.state_dict()) == set(sub(fx, "p."))
    m.load_state_dict(sub(fx, "p."))
    attach_arena(m, DEV)
    e1 = m(fx["ximg"].to(DEV)); e2 = m(fx["xpat"].to(DEV))
    assert rel_l2(e1, fx["e1"]) < 2e-2 and rel_l2(e2, fx["e2"]) < 2e-2
    ((e1 * fx["w1"].to(DEV)).sum() + (e2 * fx["w2"].to(DEV)).sum()).backward()
    gate_errors("F7 Part-fViT", {k: rel_l2(dict(m.named_parameters())[k].grad, g) for k, g in sub(fx, "g.").items()}, GATE_F7)


@pytest.mark.parametrize("n", [196, 36])
def test_f8_landmark_gather(n):
    fx = load_golden(f"f8_gather_n{n}")
    img = fx["img"].to(DEV).requires_grad_(True)
    th = fx["theta"].to(DEV).requires_grad_(True)
    out = extract_patches_pytorch_gridsample(img, th, patch_shape=torch.tensor([8, 8]), num_landm=n)
    torch.testing.assert_close(out.cpu(), fx["out"], rtol=1e-4, atol=1e-4)   # bilinear weights from fp32 coordinate arithmetic
    (out * fx["w"].to(DEV)).sum().backward()
    assert rel_l2(th.grad, fx["gtheta"]) < 1e-4
    assert rel_l2(img.grad, fx["gimg"]) < 1e-4


def test_f10_cosface_hard_and_soft():
    fx = load_golden("f10_cosface")
    cf = CosFace(64, 300, None, s=64.0, m=0.4).to(DEV)
    cf.weight.data.copy_(fx["weight"])
    x = fx["x"].to(DEV).requires_grad_(True)
    out = cf(x, fx["y"].to(DEV))
    assert rel_l2(out, fx["out_hard"]) < 1e-2
    (out * fx["w"].to(DEV)).sum().backward()
    assert rel_l2(x.grad, fx["gx_hard"]) < 3e-2 and rel_l2(cf.weight.grad, fx["gw_hard"]) < 3e-2
    x.grad = None; cf.weight.grad = None
    ys = fx["ysoft"].to(DEV)
    out = cf(x, ys)
    assert rel_l2(out, fx["out_soft"]) < 1e-2
    ce = torch.sum(-ys * torch.log_softmax(out, dim=-1), dim=-1).mean()
    assert abs(float(ce) - float(fx["ce_soft"])) / float(fx["ce_soft"]) < 2e-2


def test_f11_mixup_class_matches_reference():
    from lafs_cvpr2024_amd.util.mixup_my import Mixup
    fx = load_golden("f11_mixup")
    mix = Mixup(mixup_alpha=0.2, cutmix_alpha=0.0, prob=1.0, mode="batch", label_smoothing=0.0, num_classes=50)
    np.random.seed(11)
    x, t = mix(fx["x_in"].clone().to(DEV), fx["y"].to(DEV))
    torch.testing.assert_close(x.cpu(), fx["x_out"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(t.cpu(), fx["target"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("margin_type,m", [(0, 0.4), (1, 0.5)])
@pytest.mark.parametrize("B,C", [(8, 1000), (32, 20533)])
def test_capture_friendly_margin_kernel_equals_the_in_place_one(margin_type, m, B, C):
    """lafs_margin_softmax_ce_bf16 (row x chunk workgroups, bf16 gradient out, flipped labels implied, lambda from device memory) ==
    lafs_margin_softmax_ce (one workgroup per row, F10-pinned): same loss, same gradient up to the bf16 rounding of the output; plus
    the small re-indexing kernels of the captured fine-tune step (bf16 transpose, int64 -> int32 labels, unpatchify)."""
    from lafs_cvpr2024_amd import functional as Fn
    g = torch.Generator().manual_seed(31 + B)
    Cpad = (C + 127) // 128 * 128
    cos = (torch.rand(B, Cpad, generator=g) * 2 - 1).to(DEV)
    cos[:, C:] = 0
    lam = 1.0 if margin_type == 1 else 0.3
    y64 = torch.randint(0, C, (B,), generator=g).to(DEV)
    y1 = torch.empty(B, device=DEV, dtype=torch.int32)
    call("lafs_cast_i64_i32", _p(y64), _p(y1), B)
    assert torch.equal(y1.long(), y64)
    y2 = y1.flip(0).contiguous()
    ref = cos.clone(); loss_ref = torch.zeros(1, device=DEV); ws = torch.empty(B, device=DEV)
    call("lafs_margin_softmax_ce", _p(ref), Cpad, B, C, _p(y1), _p(y2), lam, 64.0, m, margin_type, 1.0, _p(loss_ref), _p(ws))
    lam_dev = torch.tensor([lam], device=DEV)
    for y2_arg, lam_arg, lam_ptr in ((None, -7.0, lam_dev), (y2, lam, None)):        # implied partner + device lambda / explicit + scalar
        dcos = torch.full((B, Cpad), 7.0, device=DEV, dtype=torch.bfloat16)
        loss = torch.zeros(1, device=DEV); part = torch.empty(B * 32, device=DEV)
        call("lafs_margin_softmax_ce_bf16", _p(cos), Cpad, B, C, _p(y1), _p(y2_arg), lam_arg, _p(lam_ptr), 64.0, m, margin_type, 1.0,
             _p(dcos), Cpad, _p(loss), _p(ws), _p(part))
        assert abs(float(loss) - float(loss_ref)) < 1e-5 * abs(float(loss_ref)), (float(loss), float(loss_ref))
        assert rel_l2(dcos[:, :C].float(), ref[:, :C]) < 4e-3                   # bf16 rounding of the gradient
        assert float(dcos[:, C:].float().abs().max()) == 0.0
    t = torch.empty(Cpad, B, device=DEV, dtype=torch.bfloat16)
    call("lafs_transpose_bf16", _p(dcos), B, Cpad, Cpad, _p(t), B)
    assert torch.equal(t, dcos.t().contiguous())
    for order in (0, 1):
        dp = torch.randn(B, 196, 192, generator=g).to(DEV)
        out = torch.empty(B, 3, 112, 112, device=DEV)
        call("lafs_unpatchify_f32", _p(dp), B, 112, order, _p(out))
        assert torch.equal(out, Fn.unpatchify_grad(dp, order).contiguous())


@pytest.mark.parametrize("with_land", [False, True])
def test_captured_finetune_step_equals_the_eager_single_stream_step(with_land, monkeypatch):
    """The fine-tune micro-step as hipGraphs (the default; with the landmark branch: weight gradients deferred onto a second stream
    beside the CNN's backward) against the same engine run eagerly: three accumulation windows of two micro-steps with live dropout / DropPath / mixup.  Without the
    landmark branch the two must agree to fp32 round-off (a buffer reused while the second stream still reads it -- the hazard round
    3's side-stream experiment ran into -- shows up at the percent level).  With the trainable landmark branch the losses before
    the first update agree as tightly (its BatchNorm sums are fp64 since round 4: the forward is deterministic); after updates the
    comparison is held against the branch's own run-to-run noise, measured in the same test (two eager runs): gradient atomics
    (depthwise weight gradients, bias column sums) through Adam's first steps and a batch-statistics CNN that amplifies weight
    perturbations ~100x."""
    from conftest import det_fill_random
    from lafs_cvpr2024_amd.finetune_engine import FinetuneEngine
    B, C = 16, 3000
    res = {}
    for mode in ("graph", "eager", "eager2"):
        torch.manual_seed(4)
        model = ViT_face_landmark_patch8(loss_type="CosFace", GPU_ID=None, num_class=C, image_size=112, patch_size=8, dim=128, depth=3,
                                         heads=2, mlp_dim=256, dropout=0.1, emb_dropout=0.1, with_land=with_land, drop_path_rate=0.1)
        if with_land:
            det_fill_random(model.stn); det_fill_random(model.output_layer)
        model.train()
        eng = FinetuneEngine(model, B, acc_step=2, device=DEV, use_graph=(mode == "graph"))
        g = torch.Generator(device=DEV).manual_seed(9)
        losses = []
        for it in range(6):
            u8 = torch.randint(0, 256, (B, 3, 112, 112), device=DEV, dtype=torch.uint8, generator=g)
            y = torch.randint(0, C, (B,), device=DEV, generator=g)
            losses.append(float(eng.micro_step(u8, y, lam=(0.3 if it % 3 == 0 else 1.0)).item()))
            if it % 2 == 1:
                eng.optimizer_step(lr=1e-3, weight_decay=0.1)
        torch.cuda.synchronize()
        if mode == "graph":
            assert len(eng._graphs) == 2, "two captured variants: first / later micro-step of a window"
        res[mode] = dict(losses=losses, master=eng.arena.master.clone())
    a, b, b2 = res["graph"], res["eager"], res["eager2"]
    noise = max(abs(x - y) / abs(y) for x, y in zip(b2["losses"], b["losses"]))
    print(f"[finetune graph vs eager, with_land={with_land}] losses", a["losses"], b["losses"], "eager run-to-run", noise)
    tol0 = 1e-5          # before the first update: the forward is deterministic with and without the landmark branch (fp64 BatchNorm sums)
    assert abs(a["losses"][0] - b["losses"][0]) < tol0 * abs(b["losses"][0]) and abs(a["losses"][1] - b["losses"][1]) < tol0 * abs(b["losses"][1])
    for x, y in zip(a["losses"][2:], b["losses"][2:]):          # after updates: Adam's first steps amplify atomics-order noise
        assert abs(x - y) < max(2e-3, 5 * noise) * abs(y), (a["losses"], b["losses"])
    d = (a["master"] - b["master"]).abs()
    dn = (b2["master"] - b["master"]).abs()
    frac, frac_n = float((d > 1e-5).float().mean()), float((dn > 1e-5).float().mean())
    assert float(d.max()) <= 3 * 2.2 * 1e-3 and frac < max(2e-2, 1.5 * frac_n + 1e-3), (float(d.max()), frac, frac_n)


def test_deferred_block_weight_gradients_equal_the_immediate_ones(monkeypatch):
    """The captured fine-tune micro-step with the trunk's weight gradients DEFERRED (lafs_trunk_wgrad on a second stream beside the
    landmark CNN's backward, the default with the HIP landmark plan) against the immediate form (LAFS_FT_WGRAD_DEFER=0): same kernels on
    the same operands, so after a window of two micro-steps (first one writes, second one accumulates; dropout / DropPath / mixup live)
    the loss and every gradient of the transformer trunk must be IDENTICAL (embedding / margin head: to fp32 round-off) -- an operand slot reused
    before its deferred launch has read it shows up here."""
    from conftest import det_fill_random
    from lafs_cvpr2024_amd.finetune_engine import FinetuneEngine
    B, C = 16, 3000
    res = {}
    for defer in ("1", "0"):
        torch.manual_seed(4)
        model = ViT_face_landmark_patch8(loss_type="CosFace", GPU_ID=None, num_class=C, image_size=112, patch_size=8, dim=128, depth=4,
                                         heads=2, mlp_dim=256, dropout=0.1, emb_dropout=0.1, with_land=True, drop_path_rate=0.1)
        det_fill_random(model.stn); det_fill_random(model.output_layer)
        model.train()
        monkeypatch.setenv("LAFS_FT_WGRAD_DEFER", defer)
        monkeypatch.setenv("LAFS_FT_DEFER_WG", "0")     # (a workgroup cap changes the number of token slices = the fp32 summation order)
        eng = FinetuneEngine(model, B, acc_step=2, device=DEV, use_graph=True)
        assert eng.defer == (defer == "1")
        g = torch.Generator(device=DEV).manual_seed(9)
        losses = []
        for it in range(2):
            u8 = torch.randint(0, 256, (B, 3, 112, 112), device=DEV, dtype=torch.uint8, generator=g)
            y = torch.randint(0, C, (B,), device=DEV, generator=g)
            losses.append(float(eng.micro_step(u8, y, lam=(0.3 if it == 0 else 1.0)).item()))
        torch.cuda.synchronize()
        assert len(eng._graphs) == 2
        res[defer] = dict(losses=losses, grads={k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    a, b = res["1"], res["0"]
    assert a["losses"] == b["losses"], (a["losses"], b["losses"])
    n_trunk = 0
    for k, ga in a["grads"].items():
        if k.startswith("stn.") or k.startswith("output_layer."):
            continue                                   # the landmark CNN's own gradients carry atomics-order noise (checked elsewhere)
        if k.startswith("transformer.") and ".norm." not in k and k.endswith(".weight"):     # the linears' weights: what the deferred launches write
            assert torch.equal(ga, b["grads"][k]), (k, float((ga - b["grads"][k]).abs().max()))
        else:                                          # (bias / LayerNorm affine / patch-embedding gradients are summed with atomics: fp32 round-off)
            assert float((ga - b["grads"][k]).abs().max()) <= 1e-6 * float(ga.abs().max()), k
        n_trunk += k.startswith("transformer.") and ".norm." not in k and k.endswith(".weight")
    assert n_trunk >= 4 * 4
    assert float(a["grads"]["transformer.layers.0.0.fn.fn.to_qkv.weight"].abs().max()) > 0


@pytest.mark.parametrize("lam", [1.0, 0.3])
def test_finetune_micro_step_against_oracle(lam):
    """u8 batch -> mixup -> Part-fViT -> CosFace -> soft-target CE -> backward, HIP engine vs CPU oracle (autograd)."""
    from lafs_cvpr2024_amd.finetune_engine import FinetuneEngine
    from oracle import margin, partfvit
    torch.manual_seed(5)
    B, C = 8, 1000
    model = ViT_face_landmark_patch8(loss_type="CosFace", GPU_ID=None, num_class=C, image_size=112, patch_size=8, dim=128, depth=2,
                                     heads=3, mlp_dim=256, dropout=0.0, emb_dropout=0.0, with_land=False, drop_path_rate=0.0)
    P = {k: v.clone().requires_grad_(True) for k, v in model.state_dict().items()}
    u8 = torch.randint(0, 256, (B, 3, 112, 112), dtype=torch.uint8)
    labels = torch.tensor([3, 999, 17, 3, 500, 0, 42, 999])
    eng = FinetuneEngine(model, B, acc_step=1, device=DEV)
    eng.zero_after_step = True                         # (the default leaves the dead gradients in place: the next micro-step overwrites them)
    loss = eng.micro_step(u8.to(DEV), labels.to(DEV), lam=lam)
    # oracle
    cfg = partfvit.PartFViTConfig(patch_size=8, dim=128, depth=2, heads=3, mlp_dim=256, num_patches=196)
    x = u8.float() / 255 * 2 - 1
    x, tgt = margin.mixup_batch(x, labels, C, lam)
    emb = partfvit.forward_embedding(P, x, cfg)
    ref = margin.soft_target_cross_entropy(margin.cosface_logits(emb, P["loss.weight"], tgt), tgt)
    ref.backward()
    assert abs(float(loss.item()) - float(ref)) / float(ref) < 5e-3, (float(loss.item()), float(ref))
    named = dict(model.named_parameters())
    keys = ("loss.weight", "patch_to_embedding.weight", "transformer.layers.0.0.fn.fn.to_qkv.weight", "transformer.layers.1.1.fn.fn.net.3.weight",
            "pos_embedding", "cls_token", "mlp_head.0.weight", "transformer.layers.0.1.fn.fn.net.0.bias")
    gate_errors("fine-tune micro step vs oracle", {k: rel_l2(named[k].grad, P[k].grad) for k in keys}, GATE_FT)
    # one AdamW step runs and moves the weights by about lr
    w0 = named["patch_to_embedding.weight"].detach().clone()
    eng.optimizer_step(lr=1e-3, weight_decay=0.1)
    d = (named["patch_to_embedding.weight"].detach() - w0).abs().max().item()
    assert 0.5e-3 < d < 2.5e-3, d
    assert float(eng.arena.grad.abs().max()) == 0.0


def test_finetune_accumulation_window_and_adamw_against_oracle():
    """An accumulation WINDOW of the fine-tune step held against the oracle (reference train_largescale.py:842-843, 870-891,
    122-173): acc_step = 3 micro-steps on different batches (two mixed, one plain), each loss divided by 3, gradients accumulated
    -- the first micro-step of a window WRITES the block / class-table gradients, the later ones accumulate (the two captured
    variants) --, then ONE AdamW step with param_groups_lrd's decay classes (F12: matrices 1e-1, 1-D tensors 0; the `stn*` class is
    pinned on the device by test_f12_arena_applies_the_reference_weight_decay_groups).  Checked: the three losses, the accumulated
    gradient of every tensor (relative L2), and the post-step weights in units of the learning rate as F5 does -- first against
    the oracle's window (gradients differ at the bf16 level, Adam's first step is lr * sign(g): distributional), then EXACTLY
    (fp32 round-off) against the oracle's AdamW applied to the engine's own accumulated gradients, which is what a wrong decay
    class, a wrong 1/acc_step or a lost micro-step cannot pass."""
    from lafs_cvpr2024_amd.finetune_engine import FinetuneEngine
    from oracle import margin, optim, partfvit
    torch.manual_seed(7)
    B, C, ACC = 8, 1000, 3
    lr, wd = 1e-3, 0.1
    model = ViT_face_landmark_patch8(loss_type="CosFace", GPU_ID=None, num_class=C, image_size=112, patch_size=8, dim=128, depth=2,
                                     heads=3, mlp_dim=256, dropout=0.0, emb_dropout=0.0, with_land=False, drop_path_rate=0.0)
    P = {k: v.clone().requires_grad_(True) for k, v in model.state_dict().items()}
    init = {k: v.clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    batches = [(torch.randint(0, 256, (B, 3, 112, 112), dtype=torch.uint8, generator=g), torch.randint(0, C, (B,), generator=g), lam)
               for lam in (0.3, 1.0, 0.8)]
    eng = FinetuneEngine(model, B, acc_step=ACC, device=DEV)
    losses, g_after = [], []
    for u8, y, lam in batches:
        losses.append(float(eng.micro_step(u8.to(DEV), y.to(DEV), lam=lam).item()))
        g_after.append(eng.arena.grad.clone())             # the arena after 1, 2, 3 micro-steps: each step's own contribution by difference
    torch.cuda.synchronize()
    assert eng._since_opt == ACC and (not eng.use_graph or len(eng._graphs) == 2)
    named = dict(model.named_parameters())
    g_eng = {k: p.grad.detach().cpu().clone() for k, p in named.items()}
    # ---- oracle window: loss / acc_step, gradients summed over the three micro-steps
    cfg = partfvit.PartFViTConfig(patch_size=8, dim=128, depth=2, heads=3, mlp_dim=256, num_patches=196)
    ref_losses, per_step = [], []
    for i, (u8, y, lam) in enumerate(batches):
        x = u8.float() / 255 * 2 - 1
        x, tgt = margin.mixup_batch(x, y, C, lam)
        emb = partfvit.forward_embedding(P, x, cfg)
        loss = margin.soft_target_cross_entropy(margin.cosface_logits(emb, P["loss.weight"], tgt), tgt) / ACC
        before = {k: (P[k].grad.clone() if P[k].grad is not None else None) for k in ("loss.weight", "patch_to_embedding.weight")}
        loss.backward()
        ref_losses.append(float(loss))
        for k, b in before.items():                          # this micro-step alone: engine (difference of arena snapshots) vs oracle
            step_ref = P[k].grad - (b if b is not None else 0)
            o, n = eng.arena.offsets[k], eng.arena.numels[k]
            mine = (g_after[i] - (g_after[i - 1] if i else 0))[o:o + n].view(step_ref.shape).cpu()
            per_step.append((i, k, rel_l2(mine, step_ref)))
    print("[fine-tune window] per micro-step gradient errors:", ", ".join(f"step {i} {k} {e:.2e}" for i, k, e in per_step))
    for a, b in zip(losses, ref_losses):
        assert abs(a - ACC * b) / (ACC * b) < 5e-3, (losses, ref_losses)   # the engine reports the micro-batch loss itself; 1 / acc_step is in its gradient
    errs = {k: rel_l2(g_eng[k], P[k].grad) for k in named if P[k].grad is not None and float(P[k].grad.abs().max()) > 0}
    assert len(errs) >= 25
    gate_errors("fine-tune accumulation window vs oracle", errs, GATE_WINDOW)
    # ---- one AdamW step (torch.optim.AdamW semantics, param_groups_lrd decay classes)
    eng.optimizer_step(lr=lr, weight_decay=wd)
    torch.cuda.synchronize()
    after = {k: p.detach().cpu().clone() for k, p in named.items()}

    def oracle_step(grads):
        out = {}
        for k in named:
            p = init[k].clone()
            optim.adamw_step_(p, grads[k].clone(), torch.zeros_like(p), torch.zeros_like(p), 1, lr,
                              optim.finetune_weight_decay(k, tuple(p.shape), wd))
            out[k] = p
        return out
    own = oracle_step(g_eng)                       # the engine's own accumulated gradients through the oracle's optimizer: exact
    for k in named:
        d = float((after[k] - own[k]).abs().max())
        # fp32 round-off of m / (sqrt(v) + eps) (|update| <= lr) and of p (1 - lr wd)
        assert d <= 2e-3 * lr + 2e-7 * float(init[k].abs().max()), (k, d)
    ref = oracle_step({k: (P[k].grad if P[k].grad is not None else torch.zeros_like(init[k])) for k in named})
    e = torch.cat([(after[k].double() - ref[k].double()).abs().flatten() for k in named]).numpy()
    assert np.median(e) < 0.05 * lr, np.median(e)                  # Adam's first step is lr * sign(g): bf16 noise flips round-off-sized gradients
    assert np.quantile(e, 0.9) < 0.6 * lr, np.quantile(e, 0.9)
    assert e.max() < 2.2 * lr, e.max()
    # the window is closed: the next micro-step is a "first" one again and overwrites the gradients
    assert eng._since_opt == 0
    loss2 = float(eng.micro_step(batches[0][0].to(DEV), batches[0][1].to(DEV), lam=batches[0][2]).item())
    assert np.isfinite(loss2) and 0.3 * losses[0] < loss2 < losses[0]                   # same batch, weights one AdamW step further: the loss went down


def test_f9_landmark_cnn_wrapper_matches_reference():
    """Frozen MobileNetV3 landmark regressor + min-max + jitter + random-36 selection + HIP gather vs the reference
    (weights: the same closed-form fill on both sides, conftest.det_fill)."""
    from conftest import det_fill
    from lafs_cvpr2024_amd.face_pre_pro.ViT_face import face_landmark_4simmin_glo_loc
    fx = load_golden("f9_landmark_cnn")
    lc = face_landmark_4simmin_glo_loc(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8, dim=64, depth=1,
                                       heads=1, mlp_dim=64)
    assert sorted(lc.state_dict().keys()) == [str(k) for k in fx["keys"]]
    det_fill(lc)
    lc = lc.to(DEV).eval()
    x, xa = fx["x"].to(DEV), fx["x_aug"].to(DEV)
    with torch.no_grad():
        th, mo = lc(x, x_Aug=xa, Random_prob=False)
        assert th.shape == (2, 196, 2) and mo.shape == (2, 3, 112, 112)
        torch.testing.assert_close(th.cpu(), fx["theta_plain"], rtol=1e-3, atol=2e-2)        # pixels in [0, 111]
        # the gather itself is exact given the same landmarks
        torch.manual_seed(123); th_b, mo_b = lc(x, x_Aug=xa, Random_prob=True, return_prob=True)
        torch.manual_seed(124); th_c, mo_c = lc(x, x_Aug=xa, Random_prob=True, ran_sample=True)
    torch.testing.assert_close(th_b.cpu(), fx["theta_jitter"], rtol=1e-3, atol=2e-2)
    assert th_c.shape == (2, 36, 2) and mo_c.shape == (2, 3, 48, 48)
    torch.testing.assert_close(th_c.cpu(), fx["theta_local"], rtol=1e-3, atol=2e-2)
    # mosaics: feed the REFERENCE landmarks so the comparison isolates the gather
    from lafs_cvpr2024_amd.face_pre_pro.ViT_face import extract_patches_pytorch_gridsample as gather
    for th_ref, mo_ref, n in ((fx["theta_plain"], fx["mosaic_plain"], 196), (fx["theta_local"], fx["mosaic_local"], 36)):
        out = gather(xa, th_ref.to(DEV), num_landm=n)
        torch.testing.assert_close(out.cpu(), mo_ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("margin_type,m", [(0, 0.4), (1, 0.5)])
@pytest.mark.parametrize("rate", [1.0, 0.25])
def test_partial_fc_single_rank_matches_unsharded_oracle(margin_type, m, rate):
    """PartialFC (parity unpinned: self-check against the unsharded oracle).  At sample_rate 1 the loss and
    both gradients equal CosFace/ArcFace + CE over all classes; at rate < 1 they equal the same loss restricted to the
    sampled centres (positives always kept)."""
    import torch.nn.functional as F
    from lafs_cvpr2024_amd.partial_fc import PartialFC
    from oracle import margin
    torch.manual_seed(0)
    C, D, B = 1000, 64, 16
    pfc = PartialFC(D, C, B, sample_rate=rate, s=64.0, m=m, margin_type=margin_type, device="cuda", seed=0)
    emb = torch.randn(B, D, device="cuda")
    lab = torch.randint(0, C, (B,), device="cuda")
    gen_state = pfc.gen.get_state()
    loss, demb = pfc.forward_backward(emb, lab)
    pfc.gen.set_state(gen_state)
    from lafs_cvpr2024_amd.partial_fc import sample_classes
    index, y = sample_classes(lab, 0, C, pfc.num_sample, pfc.gen)
    assert index.numel() == (C if rate == 1.0 else max(int(rate * C), 1))
    Wc = pfc.weight.detach().cpu().clone().requires_grad_(True)
    e = emb.cpu().clone().requires_grad_(True)
    idx, yl = index.cpu(), y.cpu().long()
    fn = margin.cosface_logits if margin_type == 0 else margin.arcface_logits
    ref = F.cross_entropy(fn(e, Wc[idx], yl, 64.0, m), yl)
    ref.backward()
    # bf16 MFMA operands for the cosine and the two gradient GEMMs: tolerance at the tensor scale
    assert abs(float(loss) - float(ref)) < 2e-2 * max(1.0, abs(float(ref)))
    scale = float(e.grad.abs().max())
    assert float((demb.cpu() - e.grad).abs().max()) < 3e-2 * scale
    gw = pfc.arena.view(pfc.arena.grad, "weight", (C, D)).cpu()
    assert float((gw - Wc.grad).abs().max()) < 3e-2 * float(Wc.grad.abs().max())
    w0 = pfc.weight.detach().clone()
    pfc.optimizer_step(lr=1e-3)
    assert float((pfc.weight.detach() - w0).abs().max()) > 0 and float(pfc.arena.grad.abs().max()) == 0.0


@pytest.mark.parametrize("rate", [1.0, 0.25])
def test_partial_fc_soft_mixup_targets_match_the_unsharded_cosface_soft_oracle(rate):
    """The class-sharded head with the targets the reference ALWAYS feeds its margin head (train_largescale.py:802: Mixup's dense
    lam e_y + (1 - lam) e_flip(y), entering the margin itself, ViT_face.py:69-73 -- the unsharded form is pinned by F10): two (class,
    weight) pairs per row in lafs_shard_margin_*.  At sample_rate 1 loss and both gradients equal CosFace(soft) + SoftTargetCE over
    all classes; at rate < 1 the same loss restricted to the sampled centres (both labels' classes always kept).  Rows whose partner
    carries the same class (the middle of an odd flip, repeated ids) collapse to a hard label."""
    from lafs_cvpr2024_amd.partial_fc import PartialFC, sample_classes
    from oracle import margin
    torch.manual_seed(1)
    C, D, B, lam = 1000, 64, 16, 0.3
    pfc = PartialFC(D, C, B, sample_rate=rate, s=64.0, m=0.4, margin_type=0, device="cuda", seed=0)
    emb = torch.randn(B, D, device="cuda")
    lab = torch.randint(0, C, (B,), device="cuda")
    lab[5] = lab[B - 1 - 5]                                            # a row mixed with its own class
    lab2 = lab.flip(0)
    state = pfc.gen.get_state()
    loss, demb = pfc.forward_backward(emb, lab, labels2=lab2, lam=lam)
    pfc.gen.set_state(state)
    index, y, y2 = sample_classes(lab, 0, C, pfc.num_sample, pfc.gen, labels2=lab2)
    idx = index.cpu()
    Wc = pfc.weight.detach().cpu().clone().requires_grad_(True)
    e = emb.cpu().clone().requires_grad_(True)
    tgt = torch.zeros(B, idx.numel())
    tgt[torch.arange(B), y.cpu().long()] += lam
    tgt[torch.arange(B), y2.cpu().long()] += 1.0 - lam
    assert float(tgt.sum(1).min()) == pytest.approx(1.0) and float(tgt[5].max()) == pytest.approx(1.0)
    ref = margin.soft_target_cross_entropy(margin.cosface_logits(e, Wc[idx], tgt, 64.0, 0.4), tgt)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 2e-2 * max(1.0, abs(float(ref))), (float(loss), float(ref))
    assert float((demb.cpu() - e.grad).abs().max()) < 3e-2 * float(e.grad.abs().max())
    gw = pfc.arena.view(pfc.arena.grad, "weight", (C, D)).cpu()
    assert float((gw - Wc.grad).abs().max()) < 3e-2 * float(Wc.grad.abs().max())
    # hard labels are the lam = 1 special case of the same kernels
    pfc.arena.zero_grad()
    pfc.gen.set_state(state)
    l1, d1 = pfc.forward_backward(emb, lab, labels2=lab2, lam=1.0)
    pfc.arena.zero_grad()
    pfc.gen.set_state(state)
    l0, d0 = pfc.forward_backward(emb, lab)
    if rate == 1.0:                                                    # (at rate < 1 the partners' classes change the sampled set)
        assert abs(float(l1) - float(l0)) < 1e-6 * abs(float(l0))
        assert float((d1 - d0).abs().max()) <= 1e-5 * float(d0.abs().max())      # (dE_n is a split-K sum with fp32 atomics: not bitwise)


def test_f13_partfvit_with_trainable_landmark_branch():
    """ViT_face_landmark_patch8(with_land=True) as train_largescale.py builds it (:432,556): CNN -> theta -> HIP gather ->
    Part-fViT, and the backward through the patch embedding and the gather into the CNN, against the reference
    (eval mode; identical closed-form weights on both sides)."""
    from conftest import det_fill
    fx = load_golden("f13_partfvit_land")
    m = ViT_face_landmark_patch8(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8, dim=128, depth=2,
                                 heads=3, mlp_dim=256, dropout=0.0, emb_dropout=0.0, with_land=True)
    det_fill(m.stn); det_fill(m.output_layer)
    missing, unexpected = m.load_state_dict(sub(fx, "p."), strict=False)
    assert not unexpected and all(k.startswith(("stn.", "output_layer.")) for k in missing)
    attach_arena(m, DEV)
    m.eval()
    e = m(fx["x"].to(DEV))
    torch.testing.assert_close(m.theta.detach().cpu(), fx["theta"], rtol=1e-3, atol=2e-2)
    assert rel_l2(e, fx["e"]) < 2e-2
    (e * fx["w"].to(DEV)).sum().backward()
    params = dict(m.named_parameters())
    gate_errors("F13 Part-fViT with_land", {k: rel_l2(params[k].grad, g) for k, g in sub(fx, "g.").items()}, GATE_F13)
    # every tensor the reference gives a gradient gets one here, with a matching norm
    ref = dict(zip([str(k) for k in fx["gnorm_keys"]], fx["gnorms"].tolist()))
    got = {k: float(p.grad.norm()) for k, p in params.items() if p.grad is not None and float(p.grad.abs().max()) > 0}
    assert set(ref) <= set(got) | {k for k, v in ref.items() if v == 0.0}
    off = {k: (got[k], v) for k, v in ref.items() if v > 0 and abs(got[k] - v) > 0.1 * v}
    assert not off, off


def test_finetune_engine_with_landmark_branch_matches_module_path():
    """FinetuneEngine on the with_land=True model (the train_largescale.py configuration) == the autograd module path
    (pinned to the reference by F13 and F10) on the same batch: loss and the gradients that flow through theta."""
    import torch.nn.functional as F
    from conftest import det_fill_random
    from lafs_cvpr2024_amd.finetune_engine import FinetuneEngine
    B, C = 8, 1000
    mk = lambda: ViT_face_landmark_patch8(loss_type="CosFace", GPU_ID=None, num_class=C, image_size=112, patch_size=8, dim=128,
                                          depth=2, heads=3, mlp_dim=256, dropout=0.0, emb_dropout=0.0, with_land=True,
                                          drop_path_rate=0.0)
    torch.manual_seed(6)
    u8 = torch.randint(0, 256, (B, 3, 112, 112), dtype=torch.uint8, device=DEV)
    labels = torch.tensor([3, 999, 17, 3, 500, 0, 42, 999], device=DEV)
    keys = ("output_layer.1.weight", "stn.features.0.0.weight", "stn.features.15.conv.7.weight", "patch_to_embedding.weight",
            "loss.weight")
    m1 = mk(); det_fill_random(m1.stn); det_fill_random(m1.output_layer); m1.eval()
    eng = FinetuneEngine(m1, B, acc_step=1, device=DEV)
    loss1 = float(eng.micro_step(u8, labels, lam=1.0).item())
    g1 = {k: dict(m1.named_parameters())[k].grad.clone() for k in keys}
    m2 = mk(); m2.load_state_dict(m1.state_dict()); attach_arena(m2, DEV); m2.eval()
    logits, _ = m2(u8.float() / 255 * 2 - 1, labels)
    loss2 = F.cross_entropy(logits, labels)
    loss2.backward()
    assert abs(loss1 - float(loss2)) < 5e-3 * abs(float(loss2)), (loss1, float(loss2))
    bad = {k: rel_l2(g1[k], dict(m2.named_parameters())[k].grad) for k in keys}
    cos = {k: float(torch.nn.functional.cosine_similarity(g1[k].flatten().double(), dict(m2.named_parameters())[k].grad.flatten().double(), dim=0))
           for k in keys}
    print("[eval-mode with_land] loss", loss1, float(loss2), "gradient rel-L2", {k: f"{v:.2e}" for k, v in bad.items()},
          "cosine", {k: f"{v:.3f}" for k, v in cos.items()})
    # Since round 5 the engine runs the landmark CNN of an EVAL-mode model on the HIP plan too (fp16, running statistics).  What does
    # not pass through theta is held as before; the CNN's own gradients pass through the min-max scaling's arg-min / arg-max, which a
    # 1 % change of the fp16 regressor re-selects (F18 measures 0.41 rel-L2 / cosine 0.92 for a torch fp16 statement of the same
    # branch), so they are held by direction here and exactly, kernel by kernel, in the BatchNorm eval test below.
    assert all(bad[k] < 5e-2 for k in ("patch_to_embedding.weight", "loss.weight")), bad
    assert all(math.isfinite(v) and v > 0.5 for v in cos.values()), cos


def test_hip_landmark_cnn_plan_against_f9_reference_landmarks():
    """The HIP launch plan of the frozen landmark CNN (what bench.py --frontend and the LAFS loop run) against the REFERENCE's own
    landmarks (F9 theta_plain: face_landmark_4simmin_glo_loc of the reference on det_fill weights, ViT_face.py:1338-1351), not
    against the torch-ROCm module: raw regressor -> per-image min-max to [0, 111] px.  bf16 NHWC activations; observed on MI355X
    with the det_fill weights (a narrow raw range that the min-max stretches): mean 0.62 px, max 1.23 px; gate mean < 1 px,
    max < 4 px (the reference jitters the landmarks by N(0, 5 px) right after)."""
    from conftest import det_fill
    from lafs_cvpr2024_amd.face_pre_pro.ViT_face import face_landmark_4simmin_glo_loc
    from lafs_cvpr2024_amd.landmark_cnn import HipLandmarkCNN
    fx = load_golden("f9_landmark_cnn")
    lc = face_landmark_4simmin_glo_loc(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8, dim=64, depth=1,
                                       heads=1, mlp_dim=64)
    det_fill(lc)
    lc = lc.to(DEV).eval()
    hip = HipLandmarkCNN(lc, DEV)
    x = fx["x"].to(DEV)
    t = hip(x)
    noise = torch.zeros(x.shape[0], 196, 2, device=DEV)
    th = torch.empty(x.shape[0], 196, 2, device=DEV)
    call("lafs_landmark_theta", _p(t), x.shape[0], 196, _p(noise), 0.0, None, 196, _p(th))
    torch.cuda.synchronize()
    d = (th.cpu() - fx["theta_plain"]).abs()
    print(f"[F9 vs HIP plan] landmark error mean {float(d.mean()):.3f} px, max {float(d.max()):.3f} px")
    assert float(d.mean()) < 1.0 and float(d.max()) < 4.0, (float(d.mean()), float(d.max()))
    # and the jittered / selected variants through the same theta kernel given the reference's own draws are covered by F9's module
    # test; here the mosaic gathered at the plan's landmarks must equal the gather at those landmarks (exact kernel, same theta)
    from lafs_cvpr2024_amd.face_pre_pro.ViT_face import extract_patches_pytorch_gridsample as gather
    mo = gather(fx["x_aug"].to(DEV), th, num_landm=196)
    assert mo.shape == fx["mosaic_plain"].shape and bool(torch.isfinite(mo).all())


@pytest.mark.parametrize("cnn_impl", ["torch", "hip"])
def test_landmark_frontend_matches_module_calls(cnn_impl):
    """LandmarkFrontEnd (1 CNN pass + 2 theta launches + 2 gather launches on a side stream) == the three
    face_landmark_4simmin_glo_loc calls of lafs_train.py:535-567 (module path pinned to the reference by F9) given the same
    jitter noise and landmark selection."""
    import types
    from conftest import det_fill
    from lafs_cvpr2024_amd.face_pre_pro.ViT_face import face_landmark_4simmin_glo_loc
    from lafs_cvpr2024_amd.landmark_frontend import LandmarkFrontEnd
    B, nl = 2, 3
    lc = face_landmark_4simmin_glo_loc(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8, dim=64, depth=1,
                                       heads=1, mlp_dim=64)
    det_fill(lc)
    lc = lc.to(DEV).eval()
    g = torch.Generator(device=DEV).manual_seed(3)
    views = [torch.randn(B, 3, 112, 112, device=DEV, generator=g).clamp(-1, 1) for _ in range(2 * (2 + nl))]
    noise = torch.randn((2 + nl) * B, 196, 2, device=DEV, generator=g)
    sel = torch.randint(0, 196, (nl * B, 36), device=DEV, generator=g, dtype=torch.int32)
    eng = types.SimpleNamespace(in_global_all=torch.zeros(2 * B, 3, 112, 112, device=DEV),
                                in_local_all=torch.zeros(nl * B, 3, 48, 48, device=DEV))
    fe = LandmarkFrontEnd(lc, B, n_local=nl, device=DEV, cnn_impl=cnn_impl)
    # the HIP plan's bf16 activations move a landmark by < 4 px (mean < 0.6 px, test above); the mosaics are then compared at the
    # front-end's OWN landmarks (the gather is exact), the fp32 module path stays at its tight tolerances
    th_tol = dict(rtol=1e-4, atol=5e-3) if cnn_impl == "torch" else dict(rtol=0, atol=4.0)
    fe(views, eng, noise=noise, sel=sel)
    fe(torch.stack(views), eng, noise=noise, sel=sel)            # stacked input, second round trip through the staging buffers
    torch.cuda.synchronize()
    with torch.no_grad():
        for i in range(2):
            th = lc.landmarks(views[2 * i]) + 5 * noise[i * B:(i + 1) * B]
            ref = extract_patches_pytorch_gridsample(views[2 * i + 1], th, num_landm=196)
            torch.testing.assert_close(fe.theta_g[i * B:(i + 1) * B], th, **th_tol)
            if cnn_impl == "hip":
                assert float((fe.theta_g[i * B:(i + 1) * B] - th).abs().mean()) < 1.0
                ref = extract_patches_pytorch_gridsample(views[2 * i + 1], fe.theta_g[i * B:(i + 1) * B], num_landm=196)
            torch.testing.assert_close(eng.in_global_all[i * B:(i + 1) * B], ref, rtol=1e-3, atol=2e-2)
        for j in range(nl):
            rows = slice((2 + j) * B, (3 + j) * B)
            th = lc.landmarks(views[4 + 2 * j]) + 5 * noise[rows]
            th = torch.gather(th, 1, sel[j * B:(j + 1) * B].long()[:, :, None].repeat(1, 1, 2))
            ref = extract_patches_pytorch_gridsample(views[5 + 2 * j], th, num_landm=36)
            torch.testing.assert_close(fe.theta_l[j * B:(j + 1) * B], th, **th_tol)
            if cnn_impl == "hip":
                ref = extract_patches_pytorch_gridsample(views[5 + 2 * j], fe.theta_l[j * B:(j + 1) * B], num_landm=36)
            torch.testing.assert_close(eng.in_local_all[j * B:(j + 1) * B], ref, rtol=1e-3, atol=2e-2)
    # device-drawn jitter / selection: right statistics, landmarks stay a perturbed subset of the clean ones
    fe(views, eng)
    torch.cuda.synchronize()
    clean = lc.landmarks(views[0]).detach()
    d = fe.theta_g[:B] - clean
    assert 3.5 < float(d.std()) < 6.5 and abs(float(d.mean())) < 1.0


def test_partfvit_element_dropout_matches_oracle_with_same_masks():
    """dropout=0.1 / emb_dropout=0.1 as train_largescale.py:552-555 configures the model: the masks are regenerated from
    (seed, row, col) in every forward/backward kernel.  The factor matrices are exported (lafs_debug_dropout_mask) and fed
    to the oracle (whose dropout sites are pinned to the reference by F14): embedding and gradients must agree."""
    from lafs_cvpr2024_amd import functional as Fn, ops
    from oracle import partfvit
    fx = load_golden("f14_partfvit_dropout")
    p = 0.1
    m = ViT_face_landmark_patch8(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8, dim=128, depth=2,
                                 heads=3, mlp_dim=256, dropout=p, emb_dropout=p, with_land=False, drop_path_rate=0.0)
    m.load_state_dict(sub(fx, "p."))
    attach_arena(m, DEV)
    m.train()
    x = fx["x"].to(DEV)
    e = m(x)
    seed = (m._drop_seed0 + 7919 * m._drop_step) & 0x3FFFFFFF
    (e * fx["w"].to(DEV)).sum().backward()
    B, N, D, H = 2, 197, 128, 256
    fac = lambda rows, cols, s: ops.dropout_mask(rows, cols, p, s, DEV).cpu()
    masks = {"emb": fac(B * N, D, seed + Fn.EMB_DROP_SITE).view(B, N, D)}
    for l in range(2):
        masks[(l, 0)] = fac(B * N, D, seed + 3 * l + 0).view(B, N, D)
        masks[(l, 1)] = fac(B * N, H, seed + 3 * l + 1).view(B, N, H)
        masks[(l, 2)] = fac(B * N, D, seed + 3 * l + 2).view(B, N, D)
    # the masks are Bernoulli(1-p) scaled by 1/(1-p), independent across sites
    for k, v in masks.items():
        kept = float((v > 0).float().mean())
        assert abs(kept - (1 - p)) < 0.01, (k, kept)
        assert torch.all((v == 0) | ((v - 1 / (1 - p)).abs() < 1e-6))
    assert float(((masks[(0, 0)] > 0) == (masks[(0, 2)] > 0)).float().mean()) < 0.85
    cfg = partfvit.PartFViTConfig(patch_size=8, dim=128, depth=2, heads=3, mlp_dim=256, num_patches=196)
    P = {k: v.clone().requires_grad_(True) for k, v in sub(fx, "p.").items()}
    ref = partfvit.forward_embedding(P, fx["x"], cfg, masks=masks)
    (ref * fx["w"]).sum().backward()
    assert rel_l2(e, ref) < 2e-2, rel_l2(e, ref)
    named = dict(m.named_parameters())
    gate_errors("F14 dropout sites", {k: rel_l2(named[k].grad, v.grad) for k, v in P.items() if v.grad is not None}, GATE_F14)
    # a second forward draws a different mask; eval mode draws none and is deterministic
    e2 = m(x)
    assert rel_l2(e2, e) > 1e-3
    m.eval()
    assert rel_l2(m(x), m(x)) == 0.0


def test_finetune_engine_with_sharded_head_single_rank():
    """FinetuneEngine(sharded_head=PartialFC) at world 1 == the dense CosFace engine path on the same weights (hard labels)."""
    from lafs_cvpr2024_amd.finetune_engine import FinetuneEngine
    from lafs_cvpr2024_amd.partial_fc import PartialFC
    B, C = 8, 512
    mk = lambda lt: ViT_face_landmark_patch8(loss_type=lt, GPU_ID=None, num_class=C, image_size=112, patch_size=8, dim=128, depth=2,
                                             heads=3, mlp_dim=256, dropout=0.0, emb_dropout=0.0, with_land=False, drop_path_rate=0.0)
    torch.manual_seed(8)
    dense = mk("CosFace")
    bare = mk("None")
    bare.load_state_dict({k: v for k, v in dense.state_dict().items() if not k.startswith("loss.")})
    u8 = torch.randint(0, 256, (B, 3, 112, 112), dtype=torch.uint8, device=DEV)
    labels = torch.randint(0, C, (B,), device=DEV)
    e1 = FinetuneEngine(dense, B, acc_step=1, device=DEV)
    l1 = float(e1.micro_step(u8, labels, lam=1.0).item())
    head = PartialFC(128, C, B, sample_rate=1.0, device=DEV)
    with torch.no_grad():
        head.weight.copy_(dense.loss.weight.detach())
    head.arena.refresh_shadows()
    e2 = FinetuneEngine(bare, B, acc_step=1, device=DEV, sharded_head=head)
    e2.zero_after_step = True
    l2 = float(e2.micro_step(u8, labels, lam=1.0).item())              # (lam = None would draw the mixup decision from the global RNG: order-dependent)
    assert abs(l1 - l2) < 5e-3 * abs(l1), (l1, l2)
    g1, g2 = dict(dense.named_parameters()), dict(bare.named_parameters())
    for k in ("patch_to_embedding.weight", "transformer.layers.0.0.fn.fn.to_qkv.weight", "transformer.layers.1.1.fn.fn.net.3.weight"):
        assert rel_l2(g2[k].grad, g1[k].grad) < 3e-2, k
    assert rel_l2(head.arena.view(head.arena.grad, "weight", (C, 128)), g1["loss.weight"].grad) < 3e-2
    e2.optimizer_step(lr=1e-3)
    assert float(head.arena.grad.abs().max()) == 0.0 and float(e2.arena.grad.abs().max()) == 0.0


def test_hip_landmark_cnn_matches_torch_module():
    """HipLandmarkCNN (folded BN, NHWC bf16 activations, 1x1 convolutions on the MFMA GEMM, HIP stem / depthwise /
    squeeze-excite kernels) vs the fp32 nn.Module it was compiled from (which F9 pins to the reference): raw regressor output
    and the resulting landmarks."""
    from lafs_cvpr2024_amd.face_pre_pro.ViT_face import face_landmark_4simmin_glo_loc
    from lafs_cvpr2024_amd.landmark_cnn import HipLandmarkCNN
    torch.manual_seed(21)
    lc = face_landmark_4simmin_glo_loc(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8, dim=64, depth=1,
                                       heads=1, mlp_dim=64)
    with torch.no_grad():                                   # non-trivial BatchNorm statistics / affine, a wider theta head
        for m in lc.stn.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.6, 1.4); m.weight.uniform_(0.8, 1.2); m.bias.normal_(0, 0.1)
        lc.output_layer[1].weight.normal_(0, 0.05); lc.output_layer[1].bias.normal_(0, 0.1)
    lc = lc.to(DEV).eval()
    x = torch.randn(6, 3, 112, 112, device=DEV).clamp(-1, 1)
    hip = HipLandmarkCNN(lc, DEV)
    with torch.no_grad():
        t_ref = lc.output_layer(lc.stn(x).mean(dim=(-2, -1)))
        th_ref = lc.landmarks(x)
    t = hip(x)
    assert t.shape == t_ref.shape
    assert rel_l2(t, t_ref) < 3e-2, rel_l2(t, t_ref)
    tmin, tmax = t.min(1, keepdim=True)[0], t.max(1, keepdim=True)[0]
    th = ((t - tmin) / (tmax - tmin) * 111).view(-1, 196, 2)
    d = (th - th_ref).abs()
    assert float(d.mean()) < 0.6 and float(d.max()) < 4.0, (float(d.mean()), float(d.max()))       # pixels (jitter is 5 px)
    # batch-size independent plan cache + a second resolution-free call gives identical results
    assert torch.equal(hip(x), t)


def test_full_size_c4_finetune_step_properties():
    """BASELINE.json configs[3] at FULL size (Part-fViT ViT-B, batch 128, 205 990 classes, CosFace s=64 m=0.4) through
    size-independent properties: the initial loss matches its closed-form expectation for random embeddings
    (ln C + s^2 sigma^2 / 2 + s m with sigma^2 = 1/dim), the softmax-gradient rows sum to zero, an optimizer step with lr = 0 and
    wd = 0 is the identity, a real one moves the weights by about lr."""
    import math
    from lafs_cvpr2024_amd.finetune_engine import FinetuneEngine
    torch.manual_seed(0)
    B, C, D = 128, 205990, 768
    model = ViT_face_landmark_patch8(loss_type="CosFace", GPU_ID=None, num_class=C, image_size=112, patch_size=8, dim=D, depth=12,
                                     heads=11, mlp_dim=2048, dropout=0.0, emb_dropout=0.0, with_land=False, drop_path_rate=0.0)
    eng = FinetuneEngine(model, B, acc_step=1, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(2)
    u8 = torch.randint(0, 256, (B, 3, 112, 112), device=DEV, dtype=torch.uint8, generator=g)
    y = torch.randint(0, C, (B,), device=DEV, generator=g)
    loss = float(eng.micro_step(u8, y, lam=1.0).item())
    # dL/dcos (bf16, eng.dcos): rows sum to s * sum_k (p_k - y_k) / B = 0
    dc = eng.dcos[:, :C].float()
    assert float(dc.sum(1).abs().max()) < 1e-3 * float(dc.abs().sum(1).mean())
    assert float(eng.dcos[:, C:].float().abs().max()) == 0.0                    # pad columns of the gradient operand
    # closed-form expectation.  The embeddings of one random-init network on random images are strongly correlated, so their
    # cosines to the random class centres share less variance than independent vectors: allow the variance term to be partial
    lo, hi = math.log(C) + 64 * 0.4, math.log(C) + 64 * 0.4 + 64 ** 2 / D / 2 + 0.5
    assert lo - 0.5 < loss < hi, (loss, lo, hi)
    w0 = eng.arena.master.clone()
    eng.optimizer_step(lr=0.0, weight_decay=0.0)
    assert torch.equal(eng.arena.master, w0)
    eng.micro_step(u8, y, lam=1.0)
    eng.optimizer_step(lr=1e-4, weight_decay=0.1)
    d = (eng.arena.master - w0).abs()
    assert 0.5e-4 < float(d.max()) < 3e-4


def test_finetune_engine_dense_arcface_matches_oracle():
    """--head ArcFace (config C4's margin; parity unpinned, Deng et al. formula in oracle.margin.arcface_logits): the fused
    margin + softmax + CE kernel with margin_type = 1 against the autograd oracle, hard labels."""
    import torch.nn.functional as F
    from lafs_cvpr2024_amd.finetune_engine import FinetuneEngine
    from oracle import margin, partfvit
    torch.manual_seed(9)
    B, C = 8, 1000
    model = ViT_face_landmark_patch8(loss_type="CosFace", GPU_ID=None, num_class=C, image_size=112, patch_size=8, dim=128, depth=2,
                                     heads=3, mlp_dim=256, dropout=0.0, emb_dropout=0.0, with_land=False, drop_path_rate=0.0)
    P = {k: v.clone().requires_grad_(True) for k, v in model.state_dict().items()}
    u8 = torch.randint(0, 256, (B, 3, 112, 112), dtype=torch.uint8)
    labels = torch.tensor([3, 999, 17, 3, 500, 0, 42, 999])
    eng = FinetuneEngine(model, B, acc_step=1, s=64.0, m=0.5, margin_type=1, device=DEV)
    loss = eng.micro_step(u8.to(DEV), labels.to(DEV), lam=1.0)
    cfg = partfvit.PartFViTConfig(patch_size=8, dim=128, depth=2, heads=3, mlp_dim=256, num_patches=196)
    emb = partfvit.forward_embedding(P, u8.float() / 255 * 2 - 1, cfg)
    ref = F.cross_entropy(margin.arcface_logits(emb, P["loss.weight"], labels, 64.0, 0.5), labels)
    ref.backward()
    assert abs(float(loss.item()) - float(ref)) / float(ref) < 5e-3, (float(loss.item()), float(ref))
    named = dict(model.named_parameters())
    gate_errors("ArcFace micro step vs oracle", {k: rel_l2(named[k].grad, P[k].grad)
                                                  for k in ("loss.weight", "patch_to_embedding.weight", "transformer.layers.1.1.fn.fn.net.3.weight")}, GATE_FT)


def test_f12_arena_applies_the_reference_weight_decay_groups():
    """One fused AdamW step with zero gradients is a pure decoupled decay p <- p (1 - lr wd): every tensor of a with_land
    Part-fViT must shrink by exactly the rate param_groups_lrd gives it in the reference (F12: stn* matrices 5e-2, other
    matrices 1e-1, 1-D tensors 0)."""
    from lafs_cvpr2024_amd.finetune_engine import FinetuneEngine
    fx = load_golden("f12_param_groups_lrd")
    torch.manual_seed(0)
    model = ViT_face_landmark_patch8(loss_type="CosFace", GPU_ID=None, num_class=50, image_size=112, patch_size=8, dim=64, depth=2, heads=2,
                                     mlp_dim=128, dropout=0.1, emb_dropout=0.1, with_land=True)
    eng = FinetuneEngine(model, 8, acc_step=1, device=DEV)
    named = dict(model.named_parameters())
    assert sorted(named) == sorted(str(n) for n in fx["names"])
    with torch.no_grad():
        for p in named.values():
            p.add_(1.0)                                   # keep every element away from 0
    eng.arena.refresh_shadows()
    before = {k: v.detach().clone() for k, v in named.items()}
    lr = 0.5
    eng.optimizer_step(lr=lr, weight_decay=1e-1)
    for n, wd in zip([str(n) for n in fx["names"]], fx["weight_decay"].tolist()):
        ratio = (named[n].detach() / before[n]).flatten()
        assert float((ratio - (1 - lr * wd)).abs().max()) < 1e-6, (n, wd, float(ratio.mean()))


def test_full_size_c4_reference_configuration_step_properties():
    """BASELINE.json configs[3] as train_largescale.py really builds it (:432, 542-561): Part-fViT ViT-B with the TRAINABLE landmark
    branch (with_land=True), dropout = emb_dropout 0.1, DropPath 0.1, batch 128, 205 990 classes, a mixed batch (lambda = 0.3),
    acc_step 3.  Size-independent properties: finite loss in the closed-form band of the soft-target CE, softmax-gradient rows
    sum to zero, every tensor that can receive a gradient has one (incl. stn.* through theta and the gather), lr = 0 is the
    identity, a real step moves matrices by about lr and decays stn.* at 5e-2 / the rest at 1e-1."""
    import math
    from lafs_cvpr2024_amd.finetune_engine import FinetuneEngine
    torch.manual_seed(0)
    B, C, D = 128, 205990, 768
    model = ViT_face_landmark_patch8(loss_type="CosFace", GPU_ID=None, num_class=C, image_size=112, patch_size=8, dim=D, depth=12,
                                     heads=11, mlp_dim=2048, dropout=0.1, emb_dropout=0.1, with_land=True)
    eng = FinetuneEngine(model, B, acc_step=3, device=DEV)
    eng.zero_after_step = True
    g = torch.Generator(device=DEV).manual_seed(2)
    losses = []
    for _ in range(3):
        u8 = torch.randint(0, 256, (B, 3, 112, 112), device=DEV, dtype=torch.uint8, generator=g)
        y = torch.randint(0, C, (B,), device=DEV, generator=g)
        losses.append(float(eng.micro_step(u8, y, lam=0.3).item()))
        dc = eng.dcos[:, :C].float()
        assert float(dc.sum(1).abs().max()) < 1e-3 * float(dc.abs().sum(1).mean())
    # dense soft target y = 0.3 e_a + 0.7 e_b enters the margin itself (ViT_face.py:69-73: s (cos - m y)), so the margin term of the
    # soft-target CE is s m (0.3^2 + 0.7^2) = 0.58 s m; plus ln C and the variance term of the random cosines
    lo, hi = math.log(C) + 0.58 * 64 * 0.4 - 0.5, math.log(C) + 0.58 * 64 * 0.4 + 64 ** 2 / D / 2 + 0.5
    assert all(math.isfinite(v) and lo < v < hi for v in losses), (losses, lo, hi)
    named = dict(model.named_parameters())
    dead = [k for k, p in named.items() if p.requires_grad and (p.grad is None or float(p.grad.abs().max()) == 0.0)]
    assert not dead, dead[:5]
    assert float(named["stn.features.0.0.weight"].grad.abs().max()) > 0          # the CNN is reached through theta + gather
    w0 = eng.arena.master.clone()
    grad = eng.arena.grad.clone()
    eng.optimizer_step(lr=0.0, weight_decay=0.0)
    assert torch.equal(eng.arena.master, w0) and float(eng.arena.grad.abs().max()) == 0.0
    eng.arena.grad.copy_(grad)
    eng.optimizer_step(lr=1e-4, weight_decay=0.1)
    d = (eng.arena.master - w0).abs()
    assert 0.5e-4 < float(d.max()) < 3e-4


def test_full_size_c5_partial_fc_step_properties():
    """BASELINE.json configs[4] at the per-GPU size (Part-fViT embeddings 768-d, PartialFC with sample_rate 0.1 over ~200 000
    identities, batch 256, one class shard = the whole table at world 1): every positive class is among the sampled centres,
    exactly num_sample centres are scored, the gradient rows of the sampled softmax sum to zero, the loss equals the closed form
    ln S + s m + s^2 / (2 D) for random unit embeddings, lr = 0 is the identity, unsampled centres do not move."""
    import math
    from lafs_cvpr2024_amd.partial_fc import PartialFC, sample_classes
    torch.manual_seed(0)
    B, C, D = 256, 200000, 768
    pfc = PartialFC(D, C, B, sample_rate=0.1, device=DEV)
    assert pfc.num_sample == 20000 and pfc.num_local == C
    g = torch.Generator(device=DEV).manual_seed(4)
    emb = torch.randn(B, D, device=DEV, generator=g)
    y = torch.randint(0, C, (B,), device=DEV, generator=g)
    idx, yl = sample_classes(y, 0, C, pfc.num_sample, torch.Generator(device=DEV).manual_seed(1))
    assert idx.numel() == 20000 and bool((yl >= 0).all()) and torch.equal(idx[yl.long()], y)       # positives always present
    loss, demb = pfc.forward_backward(emb, y)
    S = 20000
    rows = pfc.cos[:, :S].sum(1)                                            # dL/dcos left in place
    assert float(rows.abs().max()) < 1e-3 * float(pfc.cos[:, :S].abs().sum(1).mean())
    ref = math.log(S) + 64 * 0.4 + 64 ** 2 / D / 2
    assert abs(float(loss) - ref) < 0.6, (float(loss), ref)
    assert math.isfinite(float(demb.abs().max())) and float(demb.abs().max()) > 0
    gw = pfc.arena.view(pfc.arena.grad, "weight", (C, D))
    touched = (gw.abs().sum(1) > 0)
    assert int(touched.sum()) <= S and bool(touched[y].all())
    w0 = pfc.arena.master.clone()
    gsave = pfc.arena.grad.clone()
    pfc.optimizer_step(lr=0.0, weight_decay=0.0)
    assert torch.equal(pfc.arena.master, w0)
    pfc.arena.grad.copy_(gsave)
    pfc.optimizer_step(lr=1e-3, weight_decay=0.0)
    moved = ((pfc.arena.view(pfc.arena.master, "weight", (C, D)) - pfc.arena.view(w0, "weight", (C, D))).abs().sum(1) > 0)
    assert torch.equal(moved, touched)


def test_full_size_c5_whole_step_backbone_plus_partial_fc():
    """BASELINE.json configs[4] EXECUTED WHOLE at the per-GPU size: the Part-fViT ViT-B backbone (dim 768, depth 12, 11 heads,
    mlp 2048, as train_largescale.py:542-561 builds it, dropout / DropPath live) under FinetuneEngine with the class-sharded head
    PartialFC(768, 200 000 ids, batch 256, sample_rate 0.1) -- backbone forward, sampled margin softmax, dE back through the trunk,
    AdamW on both.  Size-independent properties: loss in the closed-form band ln S + s m (+ variance term), gradient rows of the
    sampled softmax sum to zero, every backbone tensor receives a gradient, lr = 0 is the identity for backbone AND class centres,
    a real step moves the backbone by about lr and only the sampled centres."""
    import math
    from lafs_cvpr2024_amd.finetune_engine import FinetuneEngine
    from lafs_cvpr2024_amd.partial_fc import PartialFC
    torch.manual_seed(0)
    B, C, D = 256, 200000, 768
    model = ViT_face_landmark_patch8(loss_type="None", GPU_ID=None, num_class=C, image_size=112, patch_size=8, dim=D, depth=12,
                                     heads=11, mlp_dim=2048, dropout=0.1, emb_dropout=0.1, with_land=False)
    head = PartialFC(D, C, B, sample_rate=0.1, device=DEV)
    eng = FinetuneEngine(model, B, acc_step=1, device=DEV, sharded_head=head)
    g = torch.Generator(device=DEV).manual_seed(2)
    u8 = torch.randint(0, 256, (B, 3, 112, 112), device=DEV, dtype=torch.uint8, generator=g)
    y = torch.randint(0, C, (B,), device=DEV, generator=g)
    loss = float(eng.micro_step(u8, y).item())
    S = head.num_sample
    assert S == 20000
    rows = head.cos[:, :S].sum(1)                                            # dL/dcos left in place
    assert float(rows.abs().max()) < 1e-3 * float(head.cos[:, :S].abs().sum(1).mean())
    lo, hi = math.log(S) + 64 * 0.4 - 0.5, math.log(S) + 64 * 0.4 + 64 ** 2 / D / 2 + 0.5
    assert math.isfinite(loss) and lo < loss < hi, (loss, lo, hi)
    named = dict(model.named_parameters())
    dead = [k for k, p in named.items() if p.requires_grad and (p.grad is None or float(p.grad.abs().max()) == 0.0)]
    assert not dead, dead[:5]
    gw = head.arena.view(head.arena.grad, "weight", (C, D))
    touched = gw.abs().sum(1) > 0
    assert int(touched.sum()) <= S and bool(touched[y].all())
    w0, c0 = eng.arena.master.clone(), head.arena.master.clone()
    gb, gh = eng.arena.grad.clone(), head.arena.grad.clone()
    eng.optimizer_step(lr=0.0, weight_decay=0.0)
    assert torch.equal(eng.arena.master, w0) and torch.equal(head.arena.master, c0)
    eng.arena.grad.copy_(gb); head.arena.grad.copy_(gh)
    eng.optimizer_step(lr=1e-4, weight_decay=0.0)            # (with decay every centre of the dense AdamW shard would move)
    d = (eng.arena.master - w0).abs()
    assert 0.5e-4 < float(d.max()) < 3e-4
    moved = (head.arena.view(head.arena.master, "weight", (C, D)) - head.arena.view(c0, "weight", (C, D))).abs().sum(1) > 0
    assert int(moved.sum()) <= S and bool(moved[y].all())
    # a second micro-step on the updated weights still runs and gives a finite loss (fresh sample of centres)
    assert math.isfinite(float(eng.micro_step(u8, y).item()))


@pytest.mark.parametrize("act", [0, 1, 2])
def test_batchnorm_eval_mode_kernels_against_torch(act):
    """nn.BatchNorm2d in EVAL mode on the landmark plan's kernels (round 5: a model.eval() forward that is still differentiated stays on
    the HIP plan): lafs_cnn_bn_eval_sums + lafs_cnn_bn_apply normalise with the running statistics and leave them alone;
    lafs_cnn_bn_bwd_eval treats them as constants (dx = gamma rstd dz), with the same affine gradients.  Against torch's batch_norm
    (training=False) + activation + residual in fp32 on the same fp16 inputs; the squeeze-excite pooling gradient operand included."""
    import torch.nn.functional as F
    from lafs_cvpr2024_amd.ops import _p, call
    torch.manual_seed(act)
    N, HW, C, ld = 6, 49, 24, 32
    R = N * HW
    x = torch.zeros(R, ld, device=DEV, dtype=torch.float16); x[:, :C] = torch.randn(R, C, device=DEV) * 1.5 + 0.3
    res = torch.zeros(R, ld, device=DEV, dtype=torch.float16); res[:, :C] = torch.randn(R, C, device=DEV)
    dy = torch.zeros(R, ld, device=DEV, dtype=torch.float16); dy[:, :C] = torch.randn(R, C, device=DEV)
    add = torch.zeros(N, ld, device=DEV, dtype=torch.float16); add[:, :C] = torch.randn(N, C, device=DEV)
    gamma, beta = torch.randn(C, device=DEV) * 0.5 + 1.0, torch.randn(C, device=DEV) * 0.2
    rm, rv = torch.randn(C, device=DEV) * 0.3, torch.rand(C, device=DEV) + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    eps = 1e-3
    sums = torch.zeros(2 * C, device=DEV, dtype=torch.float64); dsums = torch.zeros(2 * C, device=DEV, dtype=torch.float64)
    stat = torch.empty(2 * C, device=DEV); y = torch.full((R, ld), 7.0, device=DEV, dtype=torch.float16)
    call("lafs_cnn_bn_eval_sums", _p(rm), _p(rv), R, C, _p(sums))
    call("lafs_cnn_bn_apply", _p(x), ld, R, C, _p(sums), _p(gamma), _p(beta), eps, 0.1, None, None, act, _p(res), ld, _p(y), ld, _p(stat))
    dx = torch.empty(R, ld, device=DEV, dtype=torch.float16); dg = torch.zeros(C, device=DEV); db = torch.zeros(C, device=DEV)
    call("lafs_cnn_bn_bwd_eval", _p(dy), ld, _p(x), ld, R, C, _p(stat), _p(gamma), _p(beta), act, _p(add), ld, HW, _p(dsums), _p(dx), ld,
         _p(dg), _p(db), None)
    torch.cuda.synchronize()
    assert torch.equal(rm, rm0) and torch.equal(rv, rv0)                 # eval mode: no momentum update
    xr = x[:, :C].float().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    z = F.batch_norm(xr, rm, rv, gr, br, False, 0.1, eps)
    a = F.relu(z) if act == 1 else F.hardswish(z) if act == 2 else z
    out = a + res[:, :C].float()
    dz_in = dy[:, :C].float() + add[:, :C].float().repeat_interleave(HW, dim=0) / HW
    out.backward(dz_in)
    assert rel_l2(y[:, :C].float(), out.detach()) < 2e-3 and float(y[:, C:].float().abs().max()) == 0.0
    assert rel_l2(dx[:, :C].float(), xr.grad) < 3e-3
    assert rel_l2(dg, gr.grad) < 2e-3 and rel_l2(db, br.grad) < 2e-3


def _train_cnn_pair(B, seed, fill):
    """(model in TRAIN mode with its arena, HipLandmarkTrainer) for the landmark-branch tests."""
    from lafs_cvpr2024_amd.landmark_train import HipLandmarkTrainer
    torch.manual_seed(seed)
    m = ViT_face_landmark_patch8(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8, dim=64, depth=1, heads=1,
                                 mlp_dim=64, dropout=0.0, emb_dropout=0.0, with_land=True)
    fill(m)
    arena = attach_arena(m, DEV)
    m.train()
    return m, arena, HipLandmarkTrainer(m, arena, B, 112, device=DEV)


def _bn_eval_fn(x, bn):
    rm, rv = bn.running_mean.view(1, -1, 1, 1), bn.running_var.view(1, -1, 1, 1)
    return (x - rm) / torch.sqrt(rv + bn.eps) * bn.weight.view(1, -1, 1, 1) + bn.bias.view(1, -1, 1, 1)


def _bn_train(x, bn):
    mu = x.mean(dim=(0, 2, 3), keepdim=True)
    var = x.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    return (x - mu) / torch.sqrt(var + bn.eps) * bn.weight.view(1, -1, 1, 1) + bn.bias.view(1, -1, 1, 1)


_bn_train_fn = _bn_train


def _landmark_branch_torch(m, x, keep, rnd):
    """The trainable landmark branch (ViT_face.py:679-706, train mode) in plain differentiable torch: `rnd` is applied wherever the HIP
    plan stores a tensor as bf16 or feeds a bf16 GEMM operand (identity -> the fp32 formula of the nn.Module)."""
    import torch.nn.functional as F
    feats = m.stn.features
    cur = rnd(feats[0][2](_bn_train(rnd(F.conv2d(rnd(x), rnd(feats[0][0].weight), None, 2, 1)), feats[0][1])))
    for blk in feats[1:]:
        c = blk.conv
        e = rnd(c[2](_bn_train(rnd(F.conv2d(cur, rnd(c[0].weight))), c[1])))
        d_raw = rnd(F.conv2d(e, c[3].weight, None, c[3].stride, c[3].padding, 1, c[3].groups))
        if isinstance(c[5], torch.nn.Identity):
            d = rnd(c[6](_bn_train(d_raw, c[4])))
        else:
            zb = rnd(_bn_train(d_raw, c[4]))
            hid = rnd(F.relu(F.linear(rnd(zb.mean(dim=(2, 3))), rnd(c[5].fc[0].weight))))
            gate = rnd(F.hardsigmoid(F.linear(hid, rnd(c[5].fc[2].weight))))
            d = rnd(c[6](zb * gate[:, :, None, None]))
        y = _bn_train(rnd(F.conv2d(d, rnd(c[7].weight))), c[8])
        cur = rnd(y + cur if blk.residual else y)
    feat = rnd(rnd(cur.mean(dim=(2, 3))) * keep)
    t = F.linear(feat, rnd(m.output_layer[1].weight), m.output_layer[1].bias)
    tmax, tmin = t.max(1, keepdim=True)[0], t.min(1, keepdim=True)[0]
    return t, ((t - tmin) / (tmax - tmin) * 111).view(x.shape[0], -1, 2)


def _ste_bf16(t):
    return t + (t.to(torch.bfloat16).float() - t).detach()


def _ste_f16(t):
    """straight-through IEEE fp16 rounding: what the HIP plan stores (round 4) and what the reference's autocast run computes in"""
    return t + (t.to(torch.float16).float() - t).detach()


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_landmark_cnn_training_plan_stage_by_stage(mode):
    """(mode "eval", round 5: the same plan for a model.eval() forward that is still differentiated -- BatchNorm with the running
    statistics as constants of the backward, no dropout; the statistics are first set to this batch's own by one momentum-1 training
    forward, so that the activations stay in fp16 range.)
    Kernel correctness of the trainable landmark branch's HIP plan (landmark_train.py), batch 8: EVERY stage of the forward and
    of the backward against the same stage stated in torch (fp32 ops, an fp16 rounding wherever the plan stores fp16), each stage fed
    the PLAN'S OWN inputs.  End-to-end comparisons of two 16-bit pipelines cannot be tight here: roundings that fall on the
    other side of a boundary in 0.01 % of the elements decorrelate the two runs' rounding noise within a few blocks (measured:
    1e-5 per stage in isolation, 4 % end to end; tools/lab/debug_cnn_train.py) -- the network-level distance to the reference is the
    subject of the F18 test below.  Also: accumulation over two micro-steps, operand refresh after a weight change."""
    import torch.nn.functional as F
    from conftest import det_fill_random

    def fill(m):
        det_fill_random(m.stn); det_fill_random(m.output_layer)
    B = 8
    m, arena, tr = _train_cnn_pair(B, 3, fill)
    x = torch.randn(B, 3, 112, 112, device=DEV).clamp(-1, 1)
    keep = (torch.rand(B, 160, device=DEV) >= 0.5).float() / 0.5
    train = mode == "train"
    if not train:
        bns = [mod for mod in m.stn.modules() if isinstance(mod, torch.nn.BatchNorm2d)]
        for mod in bns:
            mod.momentum = 1.0
        with torch.no_grad():
            m.stn(x)                                         # running statistics := this batch's
        for mod in bns:
            mod.momentum = 0.1
        m.eval()
        keep = torch.ones_like(keep)                         # Dropout(0.5) is the identity in eval mode
    _bn_train = _bn_train_fn if train else _bn_eval_fn
    dth = torch.randn(B, 196, 2, device=DEV) * 0.05
    rm0 = m.stn.features[0][1].running_mean.clone()
    tr.fixed_drop, tr.keep_trace = keep, True
    theta = tr.forward(x).clone()
    tr.backward(dth)
    torch.cuda.synchronize()
    rnd = _ste_f16
    # the plan's activation gradients carry the power-of-two loss scale chosen on the device: bring the recorded ones back to units
    S = float(tr.gscale[0])
    assert S >= 1.0 and abs(float(tr.gscale[1]) * S - 1.0) < 1e-6 and float(np.log2(S)) == int(np.log2(S)), S
    for T_ in tr.trace:
        for k_ in list(T_):
            T_[k_] = T_[k_].float() / S
    nchw = lambda buf, H, C: buf.view(B, H, H, -1)[..., :C].permute(0, 3, 1, 2).float().contiguous()
    named = dict(m.named_parameters())
    feats = m.stn.features
    fwd, bwd, par = {}, {}, {}

    def grads(out, ins, gout):
        return torch.autograd.grad(out, ins, gout, allow_unused=True)

    def pgrad(key, g):
        par[key] = rel_l2(named[key].grad, g)
    trace = list(reversed(tr.trace))                        # block order
    for i, (blk, L, D, T) in enumerate(zip(feats[1:], tr.blocks, tr.B["layers"], trace), start=1):
        c, H, Ho, pre = blk.conv, D["H"], D["Ho"], f"stn.features.{i}.conv."
        cur = nchw(D["x_in"], H, L["cin"]).requires_grad_(True)
        # expand + BN1 + activation
        e_raw = rnd(F.conv2d(cur, rnd(c[0].weight)))
        e = rnd(c[2](_bn_train(e_raw, c[1])))
        fwd[f"b{i}.e_raw"] = rel_l2(nchw(D["e_raw"], H, L["cexp"]), e_raw)
        fwd[f"b{i}.e"] = rel_l2(nchw(D["e"], H, L["cexp"]), e)
        g_cur, g_w0, g_g1, g_b1 = grads(e, [cur, c[0].weight, c[1].weight, c[1].bias], nchw(T["de"], H, L["cexp"]))
        if blk.residual:
            g_cur = g_cur + nchw(T["dy"], Ho, L["cout"])
        bwd[f"b{i}.dcur"] = rel_l2(nchw(T["dcur"], H, L["cin"]), g_cur)
        pgrad(pre + "0.weight", g_w0); pgrad(pre + "1.weight", g_g1); pgrad(pre + "1.bias", g_b1)
        # depthwise
        e_in = nchw(D["e"], H, L["cexp"]).requires_grad_(True)
        d_raw = rnd(F.conv2d(e_in, c[3].weight, None, c[3].stride, c[3].padding, 1, c[3].groups))
        fwd[f"b{i}.d_raw"] = rel_l2(nchw(D["d_raw"], Ho, L["cexp"]), d_raw)
        g_e, g_w3 = grads(d_raw, [e_in, c[3].weight], nchw(T["dd_raw"], Ho, L["cexp"]))
        bwd[f"b{i}.de"] = rel_l2(nchw(T["de"], H, L["cexp"]), g_e)
        pgrad(pre + "3.weight", g_w3)
        # BN2 (+ squeeze-excite) + activation
        dr = nchw(D["d_raw"], Ho, L["cexp"]).requires_grad_(True)
        if L["se"] is None:
            d = rnd(c[6](_bn_train(dr, c[4])))
            ins = [dr, c[4].weight, c[4].bias]
        else:
            zb = rnd(_bn_train(dr, c[4]))
            hid = rnd(F.relu(F.linear(rnd(zb.mean(dim=(2, 3))), rnd(c[5].fc[0].weight))))
            gate = rnd(F.hardsigmoid(F.linear(hid, rnd(c[5].fc[2].weight))))
            d = rnd(c[6](zb * gate[:, :, None, None]))
            ins = [dr, c[4].weight, c[4].bias, c[5].fc[0].weight, c[5].fc[2].weight]
            fwd[f"b{i}.gate"] = rel_l2(D["gate"][:, :L["cexp"]].float(), gate)
        fwd[f"b{i}.d"] = rel_l2(nchw(D["d"], Ho, L["cexp"]), d)
        gs = grads(d, ins, nchw(T["dd"], Ho, L["cexp"]))
        bwd[f"b{i}.dd_raw"] = rel_l2(nchw(T["dd_raw"], Ho, L["cexp"]), gs[0])
        pgrad(pre + "4.weight", gs[1]); pgrad(pre + "4.bias", gs[2])
        if L["se"] is not None:
            pgrad(pre + "5.fc.0.weight", gs[3]); pgrad(pre + "5.fc.2.weight", gs[4])
        # project + BN3 (+ residual)
        d_in = nchw(D["d"], Ho, L["cexp"]).requires_grad_(True)
        y_raw = rnd(F.conv2d(d_in, rnd(c[7].weight)))
        y = _bn_train(y_raw, c[8])
        y = rnd(y + nchw(D["x_in"], H, L["cin"]) if blk.residual else y)
        fwd[f"b{i}.y"] = rel_l2(nchw(D["y"], Ho, L["cout"]), y)
        g_d, g_w7, g_g3, g_b3 = grads(y, [d_in, c[7].weight, c[8].weight, c[8].bias], nchw(T["dy"], Ho, L["cout"]))
        bwd[f"b{i}.dd"] = rel_l2(nchw(T["dd"], Ho, L["cexp"]), g_d)
        pgrad(pre + "7.weight", g_w7); pgrad(pre + "8.weight", g_g3); pgrad(pre + "8.bias", g_b3)
    # stem (its input gradient is not needed) and regressor head + min-max scaling
    s_raw = rnd(F.conv2d(rnd(x), rnd(feats[0][0].weight), None, 2, 1))
    x0 = rnd(feats[0][2](_bn_train(s_raw, feats[0][1])))
    fwd["x0"] = rel_l2(nchw(tr.B["x0"], 56, 16), x0)
    gs = grads(x0, [feats[0][0].weight, feats[0][1].weight, feats[0][1].bias], nchw(trace[0]["dcur"], 56, 16))
    pgrad("stn.features.0.0.weight", gs[0]); pgrad("stn.features.0.1.weight", gs[1]); pgrad("stn.features.0.1.bias", gs[2])
    Hl = tr.H_last
    y_last = nchw(tr.B["layers"][-1]["y"], Hl, 160).requires_grad_(True)
    feat = rnd(rnd(y_last.mean(dim=(2, 3))) * keep)
    t = F.linear(feat, rnd(m.output_layer[1].weight), m.output_layer[1].bias)
    tmax, tmin = t.max(1, keepdim=True)[0], t.min(1, keepdim=True)[0]
    th = ((t - tmin) / (tmax - tmin) * 111).view(B, 196, 2)
    fwd["t"] = rel_l2(tr.B["t"], t); fwd["theta"] = rel_l2(theta, th)
    g_y, g_wh, g_bh = grads(th, [y_last, m.output_layer[1].weight, m.output_layer[1].bias], dth)
    bwd["d(y_last)"] = rel_l2(nchw(trace[-1]["dy"], Hl, 160), g_y)
    pgrad("output_layer.1.weight", g_wh); pgrad("output_layer.1.bias", g_bh)
    gate_errors("landmark CNN plan, forward stages", fwd, 4e-4)                                   # observed 8.6e-5 (fp16; bf16 was 2e-3)
    gate_errors("landmark CNN plan, backward stages (activation gradients)", bwd, 1.5e-3)          # observed 3.1e-4 (bf16: 6.5e-3)
    # BatchNorm affine gradients are plain column sums of dz (xhat): where the terms cancel (a shift that the next squeeze-excite
    # or BatchNorm mostly removes) round 3's fp32 atomics showed at the 1e-2 level; the fp64 sums of round 4 do not
    bn_affine = {k: v for k, v in par.items() if k.endswith((".1.weight", ".1.bias", ".4.weight", ".4.bias", ".8.weight", ".8.bias"))}
    gate_errors("landmark CNN plan, BatchNorm affine gradients per stage", bn_affine, 2.5e-3)      # observed 4.6e-4 (fp64 sums; bf16 + fp32 atomics: 2.5e-2)
    gate_errors("landmark CNN plan, weight gradients per stage", {k: v for k, v in par.items() if k not in bn_affine}, 2e-3)   # observed 4.3e-4
    assert len(par) == 156 and len(bwd) >= 60 and len(fwd) >= 75              # every CNN tensor, every stage
    if train:
        assert float((m.stn.features[0][1].running_mean - rm0).abs().max()) > 0      # running statistics were updated
    else:
        assert torch.equal(m.stn.features[0][1].running_mean, rm0)                    # ... and left alone in eval mode
    # a second micro-step ACCUMULATES into the same gradient arena (two runs are not bit-identical: the BatchNorm sums are fp32
    # atomics, and a bf16 rounding that flips decorrelates the rounding noise downstream -- compare per tensor by direction and size)
    tr.keep_trace = False
    g1 = {k: p.grad.clone() for k, p in named.items() if k.startswith(("stn.", "output_layer."))}
    tr.forward(x); tr.backward(dth)
    # (tensors whose exact gradient is zero -- BatchNorm shifts removed again by the next batch-statistics BatchNorm -- hold rounding
    # noise only, see the F18 test: they are told apart by their norm in the yardstick run of that test, here by a norm floor)
    big = float(np.median([float(g.norm()) for g in g1.values()])) * 5e-2
    checked = 0
    for k, g in g1.items():
        if float(g.norm()) > big and not k.endswith(("conv.8.bias", "conv.1.bias", "conv.4.bias", "0.1.bias")):
            second = named[k].grad - g
            cs = float(torch.nn.functional.cosine_similarity(second.flatten(), g.flatten(), dim=0))
            assert cs > 0.85 and 0.6 < float(second.norm() / g.norm()) < 1.6, (k, cs, float(second.norm() / g.norm()))
            checked += 1
    assert checked >= 90
    # operand images follow the master weights
    with torch.no_grad():
        m.stn.features[1].conv[0].weight.mul_(0.5)
    tr.mark_stale()
    th2 = tr.forward(x)
    assert float((th2 - theta).abs().max()) > 1e-3


def test_f18_hip_training_plan_of_the_landmark_cnn_against_the_reference():
    """The plan against the REFERENCE's own train-mode run of the landmark branch (F18: ViT_face.py:679-706 with BatchNorm batch
    statistics, Dropout(0.5) mask recorded, backward from a given d(loss)/d(theta)): raw regressor, landmarks, gradients of every
    kind of tensor (stem, depthwise, 1x1, squeeze-excite FCs, BatchNorm affine, regressor), gradient norms of ALL tensors, running
    statistics.  Round 4: the plan stores fp16 -- the format the reference's own GPU run computes in under autocast
    (train_largescale.py:803-804) -- so it is held ABSOLUTELY: regressor within 2 % of the fp32 reference, landmarks within 0.5 px
    on average, and within 1.3x of the distance the same branch stated in torch with straight-through fp16 roundings has (the
    yardstick, measured here; the bf16 statement of round 3 -- 9 % / 1.5 px -- is printed beside it).  The gradient tolerances stay
    those of ANY 16-bit evaluation: the min-max scaling routes gradient through arg-min / arg-max, which a 1 % change of the
    regressor re-selects (fp16 statement: up to 0.41 rel-L2, cosine 0.92)."""
    from conftest import det_fill_random
    fx = load_golden("f18_landmark_train")

    def fill(m):
        det_fill_random(m.stn); det_fill_random(m.output_layer)
    m, arena, tr = _train_cnn_pair(4, 0, fill)
    keep = fx["drop_keep"].float().to(DEV) / 0.5                             # the reference's mask, scaled as nn.Dropout does
    x, dth = fx["x"].to(DEV), fx["dtheta"].to(DEV)
    named = dict(m.named_parameters())
    gref = sub(fx, "g.")
    cosine = lambda a, b: float(torch.nn.functional.cosine_similarity(a.flatten().cpu().double(), b.flatten().double(), dim=0))
    # this build's fp32 statement == the reference; its fp16 statement = the yardstick
    t32, th32 = _landmark_branch_torch(m, x, keep, lambda t: t)
    assert rel_l2(t32, fx["t"]) < 1e-3 and float((th32.detach().cpu() - fx["theta"]).abs().max()) < 0.1
    tb16, _ = _landmark_branch_torch(m, x, keep, _ste_bf16)
    tb, thb = _landmark_branch_torch(m, x, keep, _ste_f16)
    (thb * dth).sum().backward()
    yard_t = rel_l2(tb, fx["t"])
    yard_g = {k: rel_l2(named[k].grad, g) for k, g in gref.items()}
    yard_c = {k: cosine(named[k].grad, g) for k, g in gref.items()}
    arena.zero_grad()
    tr.fixed_drop = keep
    theta = tr.forward(x)
    tr.backward(dth)
    torch.cuda.synchronize()
    e_t = rel_l2(tr.B["t"], fx["t"])
    d = (theta.cpu() - fx["theta"]).abs()
    errs = {k: rel_l2(named[k].grad, g) for k, g in gref.items()}
    cos = {k: cosine(named[k].grad, g) for k, g in gref.items()}
    print(f"[F18] regressor rel-L2: plan {e_t:.3e}, fp16 torch statement {yard_t:.3e} (bf16 statement {rel_l2(tb16, fx['t']):.3e}); landmarks "
          f"mean {float(d.mean()):.2f} px, max {float(d.max()):.2f} px; worst gradient rel-L2: plan {max(errs.values()):.3f}, statement "
          f"{max(yard_g.values()):.3f}; min cosine: plan {min(cos.values()):.3f}, statement {min(yard_c.values()):.3f}; loss scale {float(tr.gscale[0]):.0f}")
    assert e_t < 0.02 and e_t < 1.3 * yard_t + 0.003, (e_t, yard_t)
    assert float(d.mean()) < 0.5 and float(d.max()) < 6.0, (float(d.mean()), float(d.max()))
    assert max(errs.values()) < 1.3 * max(yard_g.values()) + 0.05, (errs, yard_g)
    assert min(cos.values()) > min(yard_c.values()) - 0.08, (cos, yard_c)
    # ... and ABSOLUTELY (round-5 review: a gate relative to a yardstick computed in the same test would pass a joint regression of
    # plan and statement): observed on MI355X 0.31 / 0.95
    assert max(errs.values()) < 0.35 and min(cos.values()) > 0.93, (max(errs.values()), min(cos.values()))
    # gradient norms of ALL tensors.  (A BatchNorm bias whose output only feeds a 1x1 convolution + batch-statistics BatchNorm has
    # an exactly zero gradient -- the shift is removed again by the next mean subtraction -- which the reference reproduces down to
    # fp32 round-off (1e-4) and any 16-bit evaluation only down to ITS round-off: those tensors are left out of the comparison.)
    ref = dict(zip([str(k) for k in fx["gnorm_keys"]], fx["gnorms"].tolist()))
    floor = 1e-2 * float(np.median(list(ref.values())))
    dev_n = {k: abs(float(named[k].grad.norm()) - v) / v for k, v in ref.items() if v > floor}
    print(f"[F18] gradient norms: worst relative deviation {max(dev_n.values()):.3f} over {len(dev_n)} tensors")
    off = {k: (float(named[k].grad.norm()), ref[k]) for k, r_ in dev_n.items() if r_ > 0.25}          # (round 5: 0.5)
    assert not off, off
    assert sum(v > floor for v in ref.values()) >= 130
    tr.flush_batches_tracked()
    for k in ("stn.features.0.1", "stn.features.4.conv.4", "stn.features.15.conv.8"):
        torch.testing.assert_close(m.state_dict()[k + ".running_mean"].cpu(), fx["rm." + k], rtol=5e-2, atol=2e-2)
        torch.testing.assert_close(m.state_dict()[k + ".running_var"].cpu(), fx["rv." + k], rtol=1e-1, atol=2e-2)
    assert int(m.state_dict()["stn.features.0.1.num_batches_tracked"]) == int(fx["nbt"]) == 1


def test_captured_finetune_steps_count_batchnorm_forwards_and_survive_a_gradient_overflow():
    """Two state-keeping duties of the trainable landmark branch inside the CAPTURED fine-tune step (round-4 advisor findings):
    (1) nn.BatchNorm2d.num_batches_tracked advances by one per micro-step although the CNN's Python forward only runs at capture time
        (the reference increments it every training forward; a checkpoint after N replays must say N);
    (2) the reference wraps the step in torch.cuda.amp.GradScaler (train_largescale.py:739, 867-880): an inf / NaN in the 16-bit backward
        skips the update and backs the scale off.  Here: the loss-scale target is forced up to 2^40 (every fp16 activation gradient
        overflows), the device-side guard must zero the CNN's gradient range, halve the target and count the dropped backward; the
        AdamW step that follows must leave every weight finite, and the next windows -- target backing off by itself -- train again."""
    from conftest import det_fill_random
    from lafs_cvpr2024_amd.finetune_engine import FinetuneEngine
    torch.manual_seed(3)
    B, C = 16, 500
    model = ViT_face_landmark_patch8(loss_type="CosFace", GPU_ID=None, num_class=C, image_size=112, patch_size=8, dim=128, depth=2,
                                     heads=2, mlp_dim=256, dropout=0.1, emb_dropout=0.1, with_land=True, drop_path_rate=0.1)
    det_fill_random(model.stn); det_fill_random(model.output_layer)
    model.train()
    eng = FinetuneEngine(model, B, acc_step=1, device=DEV, use_graph=True)
    assert eng.cnn is not None
    g = torch.Generator(device=DEV).manual_seed(5)
    batch = lambda: (torch.randint(0, 256, (B, 3, 112, 112), device=DEV, dtype=torch.uint8, generator=g),
                     torch.randint(0, C, (B,), device=DEV, generator=g))
    rv0 = model.stn.features[0][1].running_var.clone()
    for _ in range(3):
        eng.micro_step(*batch(), lam=1.0)
        eng.optimizer_step(lr=1e-3)
    torch.cuda.synchronize()
    assert len(eng._graphs) == 1 and eng._warm
    assert int(model.state_dict()["stn.features.0.1.num_batches_tracked"]) == 3          # (the warm-up run before the capture is not counted,
    assert not torch.equal(model.stn.features[0][1].running_var, rv0)                     #  and its running-statistics update was undone: three momentum steps)
    target0, skipped0 = eng.cnn.overflow_state()
    assert target0 == 1024.0 and skipped0 == 0
    lo, hi = eng.cnn.grad_lo, eng.cnn.grad_hi
    w_stn = eng.arena.master[lo:hi].clone()
    # ---- force an overflow
    eng.cnn.gscale[2] = float(2 ** 40)
    eng.micro_step(*batch(), lam=1.0)
    torch.cuda.synchronize()
    target1, skipped1 = eng.cnn.overflow_state()
    assert skipped1 == 1 and target1 == float(2 ** 39), (target1, skipped1)
    assert float(eng.arena.grad[lo:hi].abs().max()) == 0.0, "the poisoned CNN gradients must have been zeroed"
    assert bool(torch.isfinite(eng.arena.grad).all())
    eng.optimizer_step(lr=1e-3)
    assert bool(torch.isfinite(eng.arena.master).all()) and bool(torch.isfinite(eng.arena.exp_avg_sq).all())
    # (zero gradient: the CNN's weights only see their decoupled decay and the decaying first moment)
    assert float((eng.arena.master[lo:hi] - w_stn).abs().max()) < 3e-3
    # ---- the target backs off by itself until the backward fits fp16 again
    for _ in range(40):
        eng.micro_step(*batch(), lam=1.0)
        eng.optimizer_step(lr=1e-3)
        if float(eng.arena.grad[lo:hi].abs().max()) > 0:
            break
    target2, skipped2 = eng.cnn.overflow_state()
    assert target2 < target1 and skipped2 > skipped1 and float(eng.arena.grad[lo:hi].abs().max()) > 0, (target2, skipped2)
    assert bool(torch.isfinite(eng.arena.master).all())
