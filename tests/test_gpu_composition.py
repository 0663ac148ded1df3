"""The composition `bench.py` actually runs, under the oracle: ViT-S/8 (D = 384, depth 12, DropPath 0.1) with 2 global + 8 local
crops at a batch where BOTH crop-resolution groups exceed the thresholds of the fast routes (two row chains, K-resident GEMM,
128x384 wide tiles), K = 100 000, one hipGraph -- loss, center, teacher EMA and every gradient against `oracle.step` fed the very
DropPath masks the device drew.  And F17: one step of the reference itself at K = 8192 with the reference's own DropPath masks.
Reference: lafs_train.py:577-613, vision_transformer.py:27-46,107-113."""
import ctypes as C
from functools import partial

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from conftest import gate_errors, load_golden, sub  # noqa: E402
from lafs_cvpr2024_amd import _lib, vision_transformer as vits  # noqa: E402
from lafs_cvpr2024_amd.dino_loss import DINOLoss  # noqa: E402
from lafs_cvpr2024_amd.engine import LafsPretrainEngine  # noqa: E402
from lafs_cvpr2024_amd.utils import MultiCropWrapper  # noqa: E402

DEV = "cuda"
LN6 = partial(nn.LayerNorm, eps=1e-6)


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _oracle_state(cfg, init, K, center=None, **kw):
    from oracle import step as ostep
    st = ostep.LafsState(cfg, out_dim=K, seed=0, **kw)
    st.student = {k: v.clone() for k, v in init.items()}
    st.teacher = {k: v.clone() for k, v in init.items()}
    st.exp_avg = {k: torch.zeros_like(v) for k, v in init.items()}
    st.exp_avg_sq = {k: torch.zeros_like(v) for k, v in init.items()}
    if center is not None:
        st.center = center.clone()
    return st


# ------------------------------------------------------------------------------------------------ F17: the reference itself
@pytest.mark.parametrize("use_graph", [False, True])
def test_f17_reference_step_at_k8192_with_the_reference_droppath_masks(use_graph):
    """Loss to 1e-3 relative against REFERENCE output (not only the oracle) at K = 8192, with stochastic depth live: the engine is
    handed the masks the reference drew; logits, center, clipped gradients and the EMA teacher follow."""
    fx = load_golden("f17_lafs_step_k8192_droppath")
    K, B = 8192, 2
    mk = lambda dpr: vits.VisionTransformer(img_size=[112], patch_size=8, embed_dim=64, depth=3, num_heads=1, qkv_bias=True,
                                            drop_path_rate=dpr, norm_layer=LN6)
    student = MultiCropWrapper(mk(0.3), vits.DINOHead(64, K, hidden_dim=128, bottleneck_dim=32, norm_last_layer=True))
    teacher = MultiCropWrapper(mk(0.0), vits.DINOHead(64, K, hidden_dim=128, bottleneck_dim=32))
    init = sub(fx, "init.")
    student.load_state_dict(init); teacher.load_state_dict(init)
    crit = DINOLoss(K, 5, 0.07, 0.04, 3, 10)
    crit.center.copy_(fx["center0"])
    eng = LafsPretrainEngine(student, teacher, crit, B, n_local=3, clip_grad=3.0, freeze_last_layer=1, use_graph=use_graph, device=DEV)
    np.testing.assert_allclose([1 - float(k) for k in eng.keep_s.cpu()], fx["rates"].numpy(), atol=1e-6)   # same per-block rates
    scales = torch.cat([fx["scales_global"], fx["scales_local"]], dim=2)                # packed order: global sequences first
    eng.set_droppath_scales(student=scales)
    lr, wd, mom = fx["hyper"].tolist()
    crops = [fx[f"crop{i}"] for i in range(5)]
    loss = eng.step(crops, lr=lr, wd=wd, momentum=mom, teacher_temp=float(crit.teacher_temp_schedule[1]), epoch=1)
    torch.cuda.synchronize()
    ref_loss = float(fx["loss"])
    rel = abs(float(loss.item()) - ref_loss) / ref_loss
    print(f"[F17] loss {float(loss.item()):.6f} vs reference {ref_loss:.6f}: rel {rel:.2e}")
    assert rel < 1e-3, (float(loss.item()), ref_loss)
    assert rel_l2(eng.logits_s[:, :K], fx["s_out"]) < 2e-2
    assert rel_l2(eng.logits_t[:, :K], fx["t_out"]) < 2e-2
    torch.testing.assert_close(crit.center.cpu(), fx["center"], rtol=0, atol=2e-3 * float(fx["t_out"].abs().max()))
    norms = dict(zip([str(n) for n in fx["norm_names"]], fx["norms"].tolist()))
    errs = {}
    for k, g in sub(fx, "grad_post.").items():
        mine = dict(student.named_parameters())[k].grad
        clip = min(1.0, 3.0 / (norms[k] + 1e-6))            # the arena keeps unclipped gradients; the clip lives in the AdamW kernel
        if float(g.abs().max()) > 1e-6:
            errs[k] = rel_l2(mine * clip, g)
    gate_errors("F17 (reference, K=8192, DropPath)", errs, 3e-2)
    e = torch.cat([(teacher.state_dict()[k].cpu().double() - v.double()).abs().flatten() for k, v in sub(fx, "teacher.").items()]).numpy()
    scale = lr * (1 - mom) * 2
    assert np.median(e) < 0.05 * scale and np.quantile(e, 0.9) < 0.6 * scale, (np.median(e), np.quantile(e, 0.9), scale)
    # the masks matter: a dropped path must show as an exact zero contribution -- rerun the forward with all-ones scales
    if not use_graph:
        eng.set_droppath_scales(student=torch.ones_like(scales))
        eng.set_inputs(crops)
        eng._seg_forward()
        assert abs(float(eng.loss.item()) - ref_loss) > 1e-4 * ref_loss


# ------------------------------------------------------------------------------------------------ the benchmarked composition
def _routes(eng):
    """(row chains of the student trunk, kernel route of the student's GEMMs per row chain: 0 tiled 128x128, 1 K-resident,
    2 K-resident ping-pong, 3 tiled 128x384)."""
    h = _lib.lib()
    d = eng._st["vit"].desc
    chains = h.lafs_trunk_row_ranges(C.byref(d))
    rows = (eng.geom_s.tok_start[1], eng.geom_s.n_tok - eng.geom_s.tok_start[1])          # the two row chains' row counts
    routes = {}
    for name, (N, K, epi) in dict(qkv=(1152, 384, _lib.EPI_BF16), fc1=(1536, 384, _lib.EPI_BF16_GELU), proj=(384, 384, _lib.EPI_RESID_F32),
                                  dgelu=(1536, 384, _lib.EPI_DGELU_BF16), fc2=(384, 1536, _lib.EPI_RESID_F32),
                                  fc1_dgrad=(384, 1536, _lib.EPI_BF16), qkv_dgrad=(384, 1152, _lib.EPI_BF16)).items():
        r = []
        for M in rows:
            a = _lib.GemmNTArgs()
            a.M, a.N, a.K, a.epilogue, a.splits = M, N, K, epi, 1
            a.A = a.B = a.C = a.C2 = a.resid = a.aux = 256                         # routing reads shapes / nullness only
            a.lda = a.ldb = K
            a.ldc = a.ldc2 = a.ldr = a.ldaux = N
            r.append(h.lafs_gemm_nt_route(C.byref(a)))
        routes[name] = tuple(r)
    return chains, rows, routes


@pytest.mark.parametrize("B", [14, 64])
def test_benchmark_composition_vit_small_droppath_against_the_oracle(B):
    """ViT-S/8, 12 blocks, D = 384, DropPath 0.1, 2 global + 8 local crops, K = 100 000, graph-captured -- the configuration
    bench.py times (BASELINE.json configs[1]).  B = 64 IS the benchmark (25 216 + 18 944 token rows: two row chains, the K-resident
    GEMM for the five K = 384 shapes, 128x384 wide tiles for the long reductions of the 197-token chain and 128x128 tiles -- 444 of
    them per GEMM -- for the 37-token chain); B = 14 is the smallest batch at which both crop-resolution groups carry >= 4096 token
    rows (5516 / 4144: same chains and K-resident routes, 128x128 tiles everywhere else).  The masks the device drew are read back
    and handed to the CPU oracle."""
    from oracle import step as ostep, vit as ovit
    torch.manual_seed(2)
    K, nl = 100000, 8
    student = MultiCropWrapper(vits.vit_small(patch_size=8, drop_path_rate=0.1), vits.DINOHead(384, K, use_bn=False, norm_last_layer=True))
    teacher = MultiCropWrapper(vits.vit_small(patch_size=8), vits.DINOHead(384, K, use_bn=False))
    teacher.load_state_dict(student.state_dict())
    init = {k: v.clone() for k, v in student.state_dict().items()}
    crops = [torch.randn(B, 3, 112, 112).clamp(-1, 1) for _ in range(2)] + [torch.randn(B, 3, 48, 48).clamp(-1, 1) for _ in range(nl)]
    crit = DINOLoss(K, 2 + nl, 0.07, 0.04, 30, 41)
    center0 = 0.05 * torch.randn(1, K)
    crit.center.copy_(center0)
    eng = LafsPretrainEngine(student, teacher, crit, B, n_local=nl, clip_grad=3.0, freeze_last_layer=1, use_graph=True, device=DEV)
    lr, wd, mom, tt = 5e-4, 0.04, 0.99, 0.05
    loss = eng.step(crops, lr=lr, wd=wd, momentum=mom, teacher_temp=tt, epoch=1)
    torch.cuda.synchronize()
    assert eng._graphs is not None and len(eng._graphs) == 1                       # one captured graph, as in the benchmark
    chains, rows, routes = _routes(eng)
    assert chains == 2, chains
    for name in ("qkv", "fc1", "proj", "dgelu"):
        assert all(r in (1, 2) for r in routes[name]), (name, routes)               # K-resident (either form)
    if B == 64:
        assert rows == (25216, 18944), rows
        assert routes["fc2"] == routes["fc1_dgrad"] == routes["qkv_dgrad"] == (3, 0), routes    # 197 wide tiles / 148 x 3 = 444 tiles of 128x128
    else:
        assert routes["fc2"] == routes["fc1_dgrad"] == routes["qkv_dgrad"] == (0, 0), routes
    ds = eng.drop_s.cpu()                                                          # [depth, 2, n_seq] as drawn inside the graph
    assert ds.shape == (12, 2, 10 * B) and int((ds == 0).sum()) > 0               # some paths were dropped
    keep = eng.keep_s.cpu()
    for l in range(12):                                                            # values are 0 or 1/keep of the block's rate
        v = ds[l].unique()
        assert all(abs(float(x)) < 1e-12 or abs(float(x) - 1 / float(keep[l])) < 1e-6 for x in v), (l, v)
    cfg = ovit.ViTConfig(patch_size=8, embed_dim=384, depth=12, num_heads=6, img_size=224)
    st = _oracle_state(cfg, init, K, center0)
    ref = ostep.lafs_step(st, crops, epoch=1, lr=lr, wd=wd, momentum=mom, teacher_temp=tt, clip_grad=3.0, freeze_last_layer=1,
                          drop_scales=[ds[:, :, :2 * B], ds[:, :, 2 * B:]])
    rel = abs(float(loss.item()) - float(ref["loss"])) / float(ref["loss"])
    print(f"[composition B={B}] loss {float(loss.item()):.6f} vs oracle {float(ref['loss']):.6f}: rel {rel:.2e}")
    assert rel < 1e-3
    assert rel_l2(eng.logits_t[:, :K], ref["teacher_out"]) < 2e-2 and rel_l2(eng.logits_s[:, :K], ref["student_out"]) < 2e-2
    c = crit.center.detach().cpu().view(-1)
    assert float((c - st.center.view(-1)).abs().max()) < 2e-3 * float(ref["teacher_out"].abs().max())
    # every gradient, per tensor (the arena holds them unclipped; the oracle's are clipped by min(1, 3 / (norm + 1e-6)))
    errs, named = {}, dict(student.named_parameters())
    for k, g in ref["grads"].items():
        clip = min(1.0, 3.0 / (ref["norms"][k] + 1e-6))
        if float(g.abs().max()) > 1e-9:
            errs[k] = rel_l2(named[k].grad * clip, g)
    assert len(errs) >= 150
    gate_errors(f"ViT-S composition B={B} (K=100000, DropPath 0.1, row chains + K-resident + wide tiles)", errs, 2e-2)      # observed 1.08e-2 at B = 14
    for k in ("head.last_layer.weight_v", "backbone.patch_embed.proj.weight", "backbone.blocks.0.attn.qkv.weight", "backbone.blocks.0.mlp.fc2.weight",
              "backbone.blocks.11.attn.qkv.weight", "backbone.blocks.11.mlp.fc1.weight"):
        print(f"[composition B={B}] {k}: gradient rel-L2 {errs[k]:.3e}")
    # teacher EMA: (1 - m) * (student update): distributional, in units of lr * (1 - m)
    e = torch.cat([(teacher.state_dict()[k].cpu().double() - v.double()).abs().flatten() for k, v in st.teacher.items()
                   if "last_layer" not in k]).numpy()
    scale = lr * (1 - mom) * 2
    assert np.median(e) < 0.05 * scale and np.quantile(e, 0.9) < 0.6 * scale, (np.median(e), np.quantile(e, 0.9), scale)
    # and the K = 100 000 last layer's gradient on its own (never compared with the oracle before)
    k_last = "head.last_layer.weight_v"
    print(f"[composition] last-layer weight_v gradient rel-L2 {errs[k_last]:.3e}")
