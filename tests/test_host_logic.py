"""CPU tests of the host-side logic (packing geometry, crop grouping, schedules, mixup, FLOP model)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from lafs_cvpr2024_amd import functional as Fn
from lafs_cvpr2024_amd.utils import MultiCropWrapper, cosine_scheduler
from oracle import optim as ooptim


def test_packed_geometry_two_resolution_groups():
    g = Fn.PackedGeometry(((4, 112), (6, 48)), 8, None)
    assert g.n_seq == 10 and g.n_tok == 4 * 197 + 6 * 37 and g.max_len == 197
    cu = g.cu_seqlens.tolist()
    assert cu[:5] == [0, 197, 394, 591, 788] and cu[-1] == g.n_tok and cu[5] == 788 + 37
    r2s = g.row2seq.tolist()
    assert r2s[0] == 0 and r2s[196] == 0 and r2s[197] == 1 and r2s[788] == 4 and r2s[-1] == 9
    assert g.tok_start == [0, 788] and g.seq_start == [0, 4] and g.npatch(0) == 196 and g.npatch(1) == 36


def test_multicrop_group_ends_matches_reference_rule():
    crops = [torch.zeros(2, 3, 112, 112)] * 2 + [torch.zeros(2, 3, 48, 48)] * 8
    assert MultiCropWrapper.group_ends(crops) == [2, 10]
    assert MultiCropWrapper.group_ends([torch.zeros(2, 196, 192)] * 2 + [torch.zeros(2, 36, 192)] * 3) == [2, 5]
    assert MultiCropWrapper.group_ends([torch.zeros(1, 3, 112, 112)]) == [1]


def test_cosine_scheduler_equals_reference_vectors():
    fx = load_golden("f6_schedules")
    np.testing.assert_allclose(cosine_scheduler(5e-4 * 64 / 256, 1e-6, 6, 11, warmup_epochs=2), fx["lr"].numpy(), rtol=1e-12)
    np.testing.assert_allclose(cosine_scheduler(0.996, 1, 6, 11), fx["mom"].numpy(), rtol=1e-12)
    np.testing.assert_allclose(cosine_scheduler(0.04, 0.4, 6, 11), ooptim.cosine_scheduler(0.04, 0.4, 6, 11), rtol=0)


def test_mixup_class_on_cpu_matches_reference():
    from lafs_cvpr2024_amd.util.mixup_my import Mixup
    fx = load_golden("f11_mixup")
    mix = Mixup(mixup_alpha=0.2, cutmix_alpha=0.0, prob=1.0, mode="batch", label_smoothing=0.0, num_classes=50)
    np.random.seed(11)
    x, t = mix(fx["x_in"].clone(), fx["y"], device="cpu")
    torch.testing.assert_close(x, fx["x_out"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(t, fx["target"], rtol=1e-6, atol=1e-7)


def test_bench_flop_model_matches_survey_numbers():
    import bench
    fl = bench.step_flops((384, 12, 6), 64, 8, 100000)
    assert abs(fl / 64 / 113.3e9 - 1) < 0.01            # 113.3 GFLOP per image


def test_finetune_lr_schedule_shape():
    from lafs_cvpr2024_amd.train_largescale import warmup_cosine
    assert warmup_cosine(1e-3, 0.0, 5, 34) == 0.0 and abs(warmup_cosine(1e-3, 5.0, 5, 34) - 1e-3) < 1e-12
    assert abs(warmup_cosine(1e-3, 34.0, 5, 34) - 1e-6) < 1e-12
    assert warmup_cosine(1e-3, 2.5, 5, 34) == ooptim.warmup_cosine_lr(1e-3, 2.5, 5, 34)


def test_partial_fc_shard_range_and_sampling():
    """PartialFC host logic (InsightFace partial_fc_v2 semantics): the shards tile the class range, positives always
    survive the negative sampling, and the remapped labels point at the right sampled centre."""
    from lafs_cvpr2024_amd.partial_fc import sample_classes, shard_range
    for C, W in ((10, 3), (2059906, 8), (7, 7), (100, 1)):
        spans = [shard_range(C, r, W) for r in range(W)]
        assert spans[0][0] == 0 and sum(n for _, n in spans) == C
        for (s0, n0), (s1, _) in zip(spans, spans[1:]):
            assert s0 + n0 == s1
    g = torch.Generator().manual_seed(0)
    labels = torch.tensor([5, 17, 17, 3, 40, 29, 11, 5])
    start, n_local = 10, 20                                   # this rank owns classes [10, 30)
    index, y = sample_classes(labels, start, n_local, 6, g)
    assert index.numel() == 6 and torch.equal(index, index.sort()[0]) and index.unique().numel() == 6
    own = (labels >= start) & (labels < start + n_local)
    assert torch.equal(y[~own], torch.full((int((~own).sum()),), -1, dtype=torch.int32))
    assert torch.equal(index[y[own].long()] + start, labels[own])
    # more positives than the sample budget: only the positives are kept
    index2, y2 = sample_classes(labels, start, n_local, 2, g)
    assert torch.equal(index2 + start, torch.tensor([11, 17, 29]))
    # sample_rate 1: identity
    index3, y3 = sample_classes(labels, start, n_local, n_local, g)
    assert torch.equal(index3, torch.arange(n_local)) and torch.equal(y3[own].long(), labels[own] - start)


def test_f15_checkpoint_layout_matches_reference_at_full_scale():
    """state_dict names AND shapes of the real configurations (ViT-S/8 + DINOHead(100000) student/teacher, DINOLoss, the
    with_land=True ViT-B fine-tune backbone with CosFace, the landmark CNN) equal the reference's: a checkpoint written by one
    side loads strictly on the other (lafs_train.py:451-460, train_largescale.py:639-661)."""
    import json
    import numpy as np
    import os
    from lafs_cvpr2024_amd import vision_transformer as vits
    from lafs_cvpr2024_amd.dino_loss import DINOLoss
    from lafs_cvpr2024_amd.face_pre_pro.ViT_face import ViT_face_landmark_patch8, face_landmark_4simmin_glo_loc
    from lafs_cvpr2024_amd.utils import MultiCropWrapper
    path = os.path.join(os.path.dirname(__file__), "golden", "f15_checkpoint_manifests.npz")
    ref = json.loads(str(np.load(path, allow_pickle=False)["manifest"]))
    man = lambda m: {k: list(v.shape) for k, v in m.state_dict().items()}
    mine = {
        "student": man(MultiCropWrapper(vits.vit_small(patch_size=8, drop_path_rate=0.1),
                                        vits.DINOHead(384, 100000, use_bn=False, norm_last_layer=True))),
        "teacher": man(MultiCropWrapper(vits.vit_small(patch_size=8), vits.DINOHead(384, 100000, False))),
        "dino_loss": man(DINOLoss(100000, 10, 0.07, 0.04, 30, 41)),
        "finetune_backbone": man(ViT_face_landmark_patch8(loss_type="CosFace", GPU_ID=None, num_class=1000, image_size=112,
                                                          patch_size=8, dim=768, depth=12, heads=11, mlp_dim=2048, dropout=0.1,
                                                          emb_dropout=0.1, with_land=True)),
        "landmark_cnn": man(face_landmark_4simmin_glo_loc(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8,
                                                          dim=768, depth=12, heads=11, mlp_dim=2048)),
    }
    for name in ref:
        missing = sorted(set(ref[name]) - set(mine[name]))
        extra = sorted(set(mine[name]) - set(ref[name]))
        assert not missing and not extra, (name, missing[:5], extra[:5])
        bad = {k: (mine[name][k], v) for k, v in ref[name].items() if mine[name][k] != v}
        assert not bad, (name, dict(list(bad.items())[:5]))


def test_f12_finetune_weight_decay_groups_match_param_groups_lrd():
    """F12: the reference's own param_groups_lrd (train_largescale.py:122-173, exec'ed out of the reference file by
    tools/make_golden.py on a with_land Part-fViT) against finetune_decay_group: 1-D tensors 0, `stn*` matrices 5e-2, the rest 1e-1.
    (The per-group lr_scale = 0.58^k the reference also stores is recorded but never applied by torch.optim.AdamW.)"""
    import types
    from lafs_cvpr2024_amd.finetune_engine import LOW_WEIGHT_DECAY, finetune_decay_group
    fx = load_golden("f12_param_groups_lrd")
    wd_of = {"none": 0.0, "low": LOW_WEIGHT_DECAY, "decay": 1e-1}
    names = [str(n) for n in fx["names"]]
    assert len(names) == 185 and any(n.startswith("stn.") for n in names)
    for n, wd, nd in zip(names, fx["weight_decay"].tolist(), fx["ndim"].tolist()):
        p = types.SimpleNamespace(dim=lambda nd=nd: nd)
        assert wd_of[finetune_decay_group(n, p)] == pytest.approx(wd), (n, wd)
    assert {0.0, 0.05, 0.1} == {round(float(w), 6) for w in fx["weight_decay"].tolist()}


def test_recordio_epoch_shards_are_equal_disjoint_and_reshuffled():
    """Every rank must see the same number of samples per epoch (else the ranks run different step counts and the last all-reduce
    hangs) and the partition is re-drawn each epoch, like DistributedSampler.set_epoch (reference lafs_train.py:186-191, 440)."""
    from lafs_cvpr2024_amd.lafs_train import epoch_shard
    n, world = 1003, 8
    e0 = [epoch_shard(n, r, world, 7, 0) for r in range(world)]
    e1 = [epoch_shard(n, r, world, 7, 1) for r in range(world)]
    assert {len(s) for s in e0} == {n // world}
    flat = [i for s in e0 for i in s]
    assert len(set(flat)) == len(flat) == (n // world) * world
    assert e0[3] != e1[3] and sorted(i for s in e1 for i in s) != sorted(flat) or e0 != e1


def test_mynet_optimizer_index_order_is_the_reference_adamw_order():
    """checkpoint['optimizer'] of the reference's real configuration (--arch mynet): torch.optim.AdamW(get_params_groups(student))
    registers Part-fViT's CosFace `loss.weight` (requires_grad stays True in the reference, lafs_train.py:316-335, 385-392) in the
    regularised group although the SSL step never gives it a gradient (no state entry).  The engine freezes that tensor but must
    keep its index, or every later index shifts and a reference checkpoint cannot be resumed (and vice versa)."""
    import types
    from lafs_cvpr2024_amd.engine import LafsPretrainEngine
    from lafs_cvpr2024_amd.lafs_train import build_backbones
    from lafs_cvpr2024_amd.utils import get_params_groups
    from lafs_cvpr2024_amd.vision_transformer import DINOHead
    args = types.SimpleNamespace(arch="mynet", mynet_dims="64,2,2,128", mynet_dropout=0.1)
    sb, _, dim = build_backbones(args)
    student = MultiCropWrapper(sb, DINOHead(dim, 256, hidden_dim=64, bottleneck_dim=32, norm_last_layer=True))
    assert not sb.loss.weight.requires_grad                                  # frozen in the arena ...
    reg, noreg = LafsPretrainEngine._adamw_order(types.SimpleNamespace(student=student))
    assert "backbone.loss.weight" in reg                                      # ... but it keeps its index
    # the reference's view of the same model: loss.weight trainable, weight_g frozen by norm_last_layer
    sb.loss.weight.requires_grad_(True)
    groups = get_params_groups(student)
    by_id = {id(p): n for n, p in student.named_parameters()}
    assert [by_id[id(p)] for p in groups[0]["params"]] == reg
    assert [by_id[id(p)] for p in groups[1]["params"]] == noreg
    # a reference-style state_dict (no state for loss.weight) round-trips through torch's own loader with these group sizes
    opt = torch.optim.AdamW(groups)
    sd = opt.state_dict()
    assert [len(g["params"]) for g in sd["param_groups"]] == [len(reg), len(noreg)]


def test_gemm_route_selection_on_the_benchmark_shapes():
    """lafs_gemm_nt_route is host logic (no launch): which kernel form every long GEMM of the three measured workloads takes.
    0 = 128x128 tiles, 1 = K-resident, 3 = 128x384 wide tile (its tiles fit one round of the chip), 4 = 160-row tiles (one round of
    the 512 workgroup slots less than 128-row tiles would need), 5 = 192x256 tiles on one persistent workgroup per CU (plain epilogue,
    wide long-K shapes whose tiles fill whole rounds of the 256 CUs)."""
    import ctypes as C
    from lafs_cvpr2024_amd import _lib
    try:
        h = _lib.lib()
    except Exception as e:                                  # pragma: no cover
        pytest.skip(f"library not built: {e}")

    def route(M, N, K, epi, act=0):
        a = _lib.GemmNTArgs()
        a.M, a.N, a.K, a.epilogue, a.splits = M, N, K, epi, 1
        a.lda, a.ldb, a.ldc, a.ldc2, a.ldr, a.ldaux = K, K, N, N, N, N
        a.act = act
        a.A = a.B = a.C = a.C2 = a.resid = a.aux = 1 << 20          # (never dereferenced: the route is a function of shapes and presence)
        return int(h.lafs_gemm_nt_route(C.byref(a)))

    B16, GELU, RES, DG = _lib.EPI_BF16, _lib.EPI_BF16_GELU, _lib.EPI_RESID_F32, _lib.EPI_DGELU_BF16
    # ViT-S step (C2): two row chains of 25 216 / 18 944 rows
    assert route(25216, 1152, 384, B16) in (1, 2) and route(18944, 1536, 384, GELU) in (1, 2)        # K = 384: K-resident
    assert route(25216, 384, 1536, RES) == 3 and route(25216, 384, 1152, B16) == 3                    # 197 wide tiles: one round
    assert route(18944, 384, 1536, RES) == 0 and route(18944, 384, 1536, B16) == 0                    # 444 tiles: one round of 128x128
    assert route(44160, 384, 1536, B16) == 4                                                          # merged rows: 1035 -> 828 tiles
    # Part-fViT (ViT-B: C4 fine-tune at 25 216 rows, the mynet pair's chains)
    for epi, N, K in ((B16, 704, 768), (RES, 768, 2048), (RES, 768, 704), (GELU, 2048, 768)):
        assert route(25216, N, K, epi) == 4, (N, K, epi)                                              # 3 -> 2 / 7 -> 5 rounds
    assert route(25216, 768, 2048, B16) == 5 and route(25216, 768, 2112, B16) == 5                    # 160x256: 474 tiles = 1.85 rounds (K >= 1024)
    assert route(25216, 2112, 768, B16) == 5                                                          # 132 x 9 tiles of 192x256 = 4.6 rounds
    assert route(44160, 768, 2048, B16) == 5 and route(44160, 2112, 768, B16) == 5                    # merged rows: 753 / 2259 tiles of 176x256, 98 % full
    assert route(44160, 704, 768, B16) == 5
    # heavy epilogues (round 5: their operand loads batched in front of the stores): GELU' everywhere it fills the rounds, the
    # residual from three rounds on, the GELU pair that saves gelu'(u); the pair that writes u stays tiled
    assert route(44160, 2048, 768, DG) == 5 and route(25216, 2048, 768, DG) == 5
    assert route(44160, 768, 2048, RES) == 5 and route(44160, 768, 704, RES) == 5 and route(25216, 768, 2048, RES) == 4
    assert route(44160, 2048, 768, GELU, act=1) == 5 and route(25216, 2048, 768, GELU, act=1) == 5 and route(44160, 2048, 768, GELU) != 5
    assert route(18944, 768, 2048, B16) == 0                                                          # 888 tiles: 2 rounds either way
    assert route(1024, 768, 2048, B16) == 0
