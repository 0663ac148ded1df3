"""GPU parity of the fused LAFS step engine (HIP, through the C ABI) against the reference's own two-step run (golden F5)
and against the CPU oracle."""
from functools import partial

import math

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from conftest import gate_errors, load_golden, sub  # noqa: E402
from lafs_cvpr2024_amd import vision_transformer as vits  # noqa: E402
from lafs_cvpr2024_amd.dino_loss import DINOLoss  # noqa: E402
from lafs_cvpr2024_amd.engine import LafsPretrainEngine  # noqa: E402
from lafs_cvpr2024_amd.utils import MultiCropWrapper  # noqa: E402

DEV = "cuda"
LN6 = partial(nn.LayerNorm, eps=1e-6)
# per-tensor relative-L2 gate on bf16-MFMA gradients against the fp32 reference (depth-2 fixtures): 2x the worst error
# observed on MI355X (DESIGN.md section 2 holds the observed table)
GRAD_GATE = 4e-2          # observed: F5 2.1e-2 / 2.7e-2 (steps 0 / 1; qkv weight), F16 2.3e-2 -- bf16 dL/dlogits at K = 512


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _build(fx, use_graph):
    mk = lambda: vits.VisionTransformer(img_size=[112], patch_size=8, embed_dim=64, depth=2, num_heads=1, qkv_bias=True, norm_layer=LN6)
    student = MultiCropWrapper(mk(), vits.DINOHead(64, 512, hidden_dim=128, bottleneck_dim=64, norm_last_layer=True))
    teacher = MultiCropWrapper(mk(), vits.DINOHead(64, 512, hidden_dim=128, bottleneck_dim=64))
    init = sub(fx, "init.")
    student.load_state_dict(init); teacher.load_state_dict(init)
    crit = DINOLoss(512, 5, 0.07, 0.04, 3, 10)
    eng = LafsPretrainEngine(student, teacher, crit, 2, n_local=3, clip_grad=3.0, freeze_last_layer=1, use_graph=use_graph, device=DEV)
    return student, teacher, crit, eng


@pytest.mark.parametrize("use_graph", [False, True])
def test_f5_two_steps_against_reference(use_graph):
    fx = load_golden("f5_lafs_step")
    student, teacher, crit, eng = _build(fx, use_graph)
    lrs, wds, moms = fx["hyper"].tolist()
    tt = crit.teacher_temp_schedule
    for s in range(2):
        crops = [fx[f"s{s}.crop{i}"] for i in range(5)]
        loss = eng.step(crops, lr=lrs[s], wd=wds[s], momentum=moms[s], teacher_temp=float(tt[s]), epoch=s)
        torch.cuda.synchronize()
        ref_loss = float(fx[f"s{s}.loss"])
        # K = 512 with tau_s = 0.1: the bf16 logits' ~1e-2 relative error shows up at the 1e-3 level in this tiny
        # config (the K = 100k workload is ln(K)-dominated and is held to 1e-3 in test_engine_matches_cpu_oracle / smoke)
        assert abs(float(loss.item()) - ref_loss) / ref_loss < 3e-3, (float(loss.item()), ref_loss)
        assert rel_l2(eng.logits_s[:, :512], fx[f"s{s}.s_out"]) < 2e-2
        assert rel_l2(eng.logits_t[:, :512], fx[f"s{s}.t_out"]) < 2e-2
        # center = EMA of the mean raw teacher logit: inherits the teacher logits' bf16-level error (~1e-2 relative)
        torch.testing.assert_close(crit.center.cpu(), fx[f"s{s}.center"], rtol=0, atol=2e-3 * float(fx[f"s{s}.t_out"].abs().max()))
        # post-clip gradients (per-tensor), as left in the arena (scaled by 1/world = 1)
        post = sub(fx, f"s{s}.grad_post.")
        norms = dict(zip([str(n) for n in fx["norm_names"]], fx[f"s{s}.norms"].tolist()))
        errs = {}
        for k, g in post.items():
            mine = dict(student.named_parameters())[k].grad
            clip = min(1.0, 3.0 / (norms[k] + 1e-6))        # the arena keeps UNclipped grads; the clip lives in the AdamW kernel
            if float(g.abs().max()) > 1e-6:
                errs[k] = rel_l2(mine * clip, g)
        gate_errors(f"F5 step {s}", errs, GRAD_GATE)
        # weights after clip + AdamW, teacher after EMA.  Adam's m/sqrt(v) amplifies bf16 gradient noise where the
        # gradient is ~0, so the check is distributional in units of the learning rate.
        for prefix, mod in (("student", student), ("teacher", teacher)):
            errs = []
            for k, v in sub(fx, f"s{s}.{prefix}.").items():
                errs.append((mod.state_dict()[k].cpu().double() - v.double()).abs().flatten())
            e = torch.cat(errs).numpy()
            scale = lrs[s] if prefix == "student" else lrs[s] * (1 - moms[s]) * 2
            assert np.median(e) < 0.05 * scale, (prefix, np.median(e), scale)
            assert np.quantile(e, 0.9) < 0.6 * scale, (prefix, np.quantile(e, 0.9), scale)
            # an element whose gradient is pure round-off can walk +-lr in opposite directions on every step
            assert e.max() < 2.2 * sum(lrs[:s + 1]), (prefix, e.max())
    # step 0 ran with the last layer frozen: its AdamW step counter lags by one
    sa = eng.sa
    idx = sa.names.index("head.last_layer.weight_v")
    assert int(sa.seg_step[idx].item()) == 1 and int(sa.seg_step[0].item()) == 2


def test_engine_matches_cpu_oracle():
    """Same init / crops through the CPU oracle and the HIP engine: loss to 1e-3 relative, teacher EMA weights close."""
    from oracle import step as ostep, vit as ovit
    torch.manual_seed(3)
    B, K, nl = 4, 1024, 2
    mk = lambda: vits.VisionTransformer(img_size=[224], patch_size=8, embed_dim=128, depth=3, num_heads=2, qkv_bias=True, norm_layer=LN6)
    student = MultiCropWrapper(mk(), vits.DINOHead(128, K, hidden_dim=256, bottleneck_dim=64))
    teacher = MultiCropWrapper(mk(), vits.DINOHead(128, K, hidden_dim=256, bottleneck_dim=64))
    teacher.load_state_dict(student.state_dict())
    init = {k: v.clone() for k, v in student.state_dict().items()}
    crops = [torch.randn(B, 3, 112, 112).clamp(-1, 1) for _ in range(2)] + [torch.randn(B, 3, 48, 48).clamp(-1, 1) for _ in range(nl)]
    crit = DINOLoss(K, 2 + nl, 0.07, 0.04, 3, 10)
    eng = LafsPretrainEngine(student, teacher, crit, B, n_local=nl, use_graph=False, device=DEV)
    cfg = ovit.ViTConfig(patch_size=8, embed_dim=128, depth=3, num_heads=2, img_size=224)
    st = ostep.LafsState(cfg, out_dim=K, seed=0, hidden_dim=256, bottleneck_dim=64)
    st.student = {k: v.clone() for k, v in init.items()}; st.teacher = {k: v.clone() for k, v in init.items()}
    st.exp_avg = {k: torch.zeros_like(v) for k, v in init.items()}; st.exp_avg_sq = {k: torch.zeros_like(v) for k, v in init.items()}
    # the forward segment is a pure function of (weights, crops): two runs must agree to fp32 atomics-order noise
    eng.set_inputs(crops)
    eng.temps.copy_(torch.tensor([0.1, 0.05]))
    eng._seg_forward(); l1 = float(eng.loss.item())
    eng._seg_forward(); l2 = float(eng.loss.item())
    assert abs(l1 - l2) < 1e-5 * abs(l1), (l1, l2)
    for it in range(2):
        loss = eng.step(crops, lr=1e-3, wd=0.04, momentum=0.9, teacher_temp=0.05, epoch=1)
        ref = ostep.lafs_step(st, crops, epoch=1, lr=1e-3, wd=0.04, momentum=0.9, teacher_temp=0.05)
        assert abs(float(loss.item()) - float(ref["loss"])) / float(ref["loss"]) < 5e-3     # small-K config, see above
    errs = torch.cat([(teacher.state_dict()[k].cpu() - v).abs().flatten() for k, v in st.teacher.items()])
    assert float(errs.median()) < 0.05 * 1e-3 * 0.2 and float(errs.max()) < 3e-3


@pytest.mark.parametrize("nl,K", [(2, 100000), (2, 65536), (0, 65536)])
def test_c1_config_vit_tiny_step_matches_oracle(nl, K):
    """BASELINE.json configs[0] (the reference's CPU-runnable case): ViT-Tiny/8, 2 global + 2 local crops, batch 8, DINO head
    with the reference's default out_dim = 100000 (lafs_train.py:44; 65536, DINO's own default, is kept as a second size) --
    one graph-captured step vs the CPU oracle: loss to 1e-3, center and teacher EMA.  nl = 0 is the degenerate multi-crop
    (global views only, two loss terms)."""
    from oracle import step as ostep, vit as ovit
    torch.manual_seed(11)
    B = 8
    student = MultiCropWrapper(vits.vit_tiny(patch_size=8, drop_path_rate=0.0), vits.DINOHead(192, K, use_bn=False, norm_last_layer=True))
    teacher = MultiCropWrapper(vits.vit_tiny(patch_size=8), vits.DINOHead(192, K, use_bn=False))
    teacher.load_state_dict(student.state_dict())
    init = {k: v.clone() for k, v in student.state_dict().items()}
    crops = [torch.randn(B, 3, 112, 112).clamp(-1, 1) for _ in range(2)] + [torch.randn(B, 3, 48, 48).clamp(-1, 1) for _ in range(nl)]
    crit = DINOLoss(K, 2 + nl, 0.07, 0.04, 3, 10)
    eng = LafsPretrainEngine(student, teacher, crit, B, n_local=nl, clip_grad=3.0, freeze_last_layer=1, use_graph=True, device=DEV)
    cfg = ovit.ViTConfig(patch_size=8, embed_dim=192, depth=12, num_heads=3, img_size=224)
    st = ostep.LafsState(cfg, out_dim=K, seed=0)
    st.student = {k: v.clone() for k, v in init.items()}; st.teacher = {k: v.clone() for k, v in init.items()}
    st.exp_avg = {k: torch.zeros_like(v) for k, v in init.items()}; st.exp_avg_sq = {k: torch.zeros_like(v) for k, v in init.items()}
    tt = float(crit.teacher_temp_schedule[1])
    loss = eng.step(crops, lr=5e-4, wd=0.04, momentum=0.996, teacher_temp=tt, epoch=1)
    ref = ostep.lafs_step(st, crops, epoch=1, lr=5e-4, wd=0.04, momentum=0.996, teacher_temp=tt, clip_grad=3.0, freeze_last_layer=1)
    assert abs(float(loss.item()) - float(ref["loss"])) < 1e-3 * float(ref["loss"]), (float(loss.item()), float(ref["loss"]))
    c = crit.center.detach().cpu().view(-1)
    assert float((c - st.center.view(-1)).abs().max()) < 2e-3 * float(ref["teacher_out"].abs().max())
    errs = torch.cat([(teacher.state_dict()[k].cpu() - v).abs().flatten() for k, v in st.teacher.items()])
    assert float(errs.median()) < 1e-5 and float(errs.max()) < 5e-4 * 0.004 * 50      # EMA moves by (1-m)*|delta| <= 0.004*lr-ish


@pytest.mark.parametrize("use_graph", [False, True])
def test_overlapped_update_pieces_equal_the_serial_tail(use_graph, monkeypatch):
    """The optimizer runs in update pieces on a stream of their own beside later backward segments (LAFS_OPT_OVERLAP, default on);
    LAFS_OPT_OVERLAP=0 runs every piece in the final segment.  Arithmetic and order within a tensor are the same in both, so the
    per-tensor gradient norms, the student weights, the Adam moments and the EMA teacher must agree to fp32 round-off (the LayerNorm
    parameter gradients are fp32 atomics: ~1e-7 run to run) -- an update that read a gradient slice before it was final would be
    orders of magnitude off.  Two steps, so that the second forward runs on weights the overlapped update wrote."""
    fx = load_golden("f5_lafs_step")
    lrs, wds, moms = fx["hyper"].tolist()
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("LAFS_OPT_OVERLAP", mode)
        student, teacher, crit, eng = _build(fx, use_graph)
        if mode == "1":
            assert len(set(eng.piece_runs_at)) > 1, "the overlapped schedule spreads the pieces over the segments"
        else:
            assert len(set(eng.piece_runs_at)) == 1
        tt = crit.teacher_temp_schedule
        losses = []
        for s in range(2):
            crops = [fx[f"s{s}.crop{i}"] for i in range(5)]
            losses.append(float(eng.step(crops, lr=lrs[s], wd=wds[s], momentum=moms[s], teacher_temp=float(tt[s]), epoch=s).item()))
        torch.cuda.synchronize()
        res[mode] = dict(losses=losses, sumsq=eng.sa.seg_sumsq.clone(), master=eng.sa.master.clone(), m=eng.sa.exp_avg.clone(),
                         v=eng.sa.exp_avg_sq.clone(), teacher=eng.ta.master.clone(), center=crit.center.clone())
    a, b = res["1"], res["0"]
    assert abs(a["losses"][0] - b["losses"][0]) <= 1e-6 * abs(b["losses"][0]) and abs(a["losses"][1] - b["losses"][1]) <= 1e-5 * abs(b["losses"][1]), (a["losses"], b["losses"])
    torch.testing.assert_close(a["sumsq"], b["sumsq"], rtol=1e-4, atol=1e-12)
    for k in ("m", "v", "center"):
        torch.testing.assert_close(a[k], b[k], rtol=1e-4, atol=1e-9)
    # weights: Adam's m / sqrt(v) flips sign where a gradient is round-off; bound the count and the size of such flips
    for k in ("master", "teacher"):
        d = (a[k] - b[k]).abs()
        assert float(d.max()) <= 2.1 * sum(lrs[:2]), (k, float(d.max()))
        assert float((d > 1e-6).float().mean()) < 2e-3, (k, float((d > 1e-6).float().mean()))


def test_pinned_ring_uploads_survive_a_host_that_runs_ahead():
    """Per-step host->device uploads (hyper-parameters, augmentation records) while the GPU is far behind the host: every
    upload must deliver ITS values (a single reused pinned buffer would hand later steps' values to earlier steps)."""
    from lafs_cvpr2024_amd.utils import PinnedRing
    a = torch.randn(8192, 8192, device=DEV)
    ring = PinnedRing((4,), torch.float32, depth=4)
    dst = torch.zeros(4, device=DEV)
    out = torch.zeros(24, 4, device=DEV)
    for _ in range(6):                                   # ~tens of ms of queued GPU work ahead of the uploads
        a = a @ a * 1e-4
    for i in range(24):
        ring.upload(dst, lambda b, i=i: b.fill_(float(i)))
        out[i].copy_(dst)
        a = a @ a * 1e-4                                 # keep the queue deep between uploads
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), torch.arange(24.0).view(-1, 1).expand(24, 4))


def test_full_size_c2_step_properties():
    """BASELINE.json configs[1] at FULL size (ViT-S/8, batch 64, 2 global + 8 local crops, K = 100 000), checked through
    size-independent properties instead of the CPU oracle (which would take minutes):
      * the softmax-gradient rows of the fused loss kernel sum to zero;
      * lr = 0 and wd = 0 leave the student bit-identical, momentum = 1 leaves the teacher bit-identical, while the center moves;
      * the loss is invariant under a permutation of the images of the batch (all crops permuted alike);
      * a real step changes the weights, and the graph-captured step reproduces the eager step."""
    torch.manual_seed(0)
    B, K, nl = 64, 100000, 8

    def build(use_graph):
        torch.manual_seed(0)
        student = MultiCropWrapper(vits.vit_small(patch_size=8, drop_path_rate=0.0), vits.DINOHead(384, K, use_bn=False, norm_last_layer=True))
        teacher = MultiCropWrapper(vits.vit_small(patch_size=8), vits.DINOHead(384, K, use_bn=False))
        teacher.load_state_dict(student.state_dict())
        crit = DINOLoss(K, 2 + nl, 0.07, 0.04, 30, 41)
        return LafsPretrainEngine(student, teacher, crit, B, n_local=nl, clip_grad=3.0, freeze_last_layer=1, use_graph=use_graph, device=DEV)

    g = torch.Generator(device=DEV).manual_seed(1)
    crops = [torch.randn(B, 3, 112, 112, device=DEV, generator=g).clamp(-1, 1) for _ in range(2)] + \
            [torch.randn(B, 3, 48, 48, device=DEV, generator=g).clamp(-1, 1) for _ in range(nl)]
    eng = build(False)
    # (1) forward segment: gradient rows of the loss sum to zero
    eng.set_inputs(crops)
    eng.temps.copy_(torch.tensor([0.1, 0.04]))
    eng._seg_forward()
    loss0 = float(eng.loss.item())
    rows = eng.dlogits[:, :K].float().sum(1)
    scale = float(eng.dlogits[:, :K].float().abs().sum(1).mean())
    assert float(rows.abs().max()) < 2e-2 * scale                       # bf16 gradient entries: zero up to their rounding
    assert 0.0 < loss0 < 2 * math.log(K)
    # (2) permutation invariance of the loss
    perm = torch.randperm(B, device=DEV, generator=g)
    eng.set_inputs([c[perm] for c in crops])
    eng._seg_forward()
    assert abs(float(eng.loss.item()) - loss0) < 2e-4 * loss0
    # (3) a step with lr = wd = 0 and momentum = 1 is the identity on student and teacher; the center still moves
    s0, t0, c0 = eng.sa.master.clone(), eng.ta.master.clone(), eng.dino_loss.center.clone()
    eng.step(crops, lr=0.0, wd=0.0, momentum=1.0, teacher_temp=0.04, epoch=1)
    assert torch.equal(eng.sa.master, s0) and torch.equal(eng.ta.master, t0)
    assert float((eng.dino_loss.center - c0).abs().max()) > 0
    # (4) a real step: weights move by about lr; the graph engine reproduces the eager one
    eng.dino_loss.center.copy_(c0)
    l_eager = float(eng.step(crops, lr=5e-4, wd=0.04, momentum=0.996, teacher_temp=0.04, epoch=1).item())
    d = (eng.sa.master - s0).abs()
    assert 1e-4 < float(d.max()) < 2e-3 and float((eng.ta.master - t0).abs().max()) > 0
    eng_g = build(True)
    l_graph = float(eng_g.step(crops, lr=5e-4, wd=0.04, momentum=0.996, teacher_temp=0.04, epoch=1).item())
    assert abs(l_graph - l_eager) < 1e-4 * l_eager
    diff = (eng_g.sa.master - eng.sa.master).abs()
    assert float(diff.median()) < 1e-7 and float((diff > 1e-4).float().mean()) < 0.02     # Adam sign flips where g is round-off


def test_bench_owns_its_launch_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher must start two rank processes by itself (before touching the GPU) and report
    n_gpus = 2; here both ranks share the one GPU over gloo (LAFS_BENCH_SHARE_GPU=1).  A --gpus / WORLD_SIZE mismatch, more GPUs
    than visible, or kernel debug flags must fail loudly instead of printing a 1-GPU number."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    small = ["--steps", "2", "--warmup", "1", "--batch", "2", "--arch", "vit_tiny", "--out-dim", "1024", "--local-crops", "2", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LAFS_DEBUG_FLAGS")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2"] + small, env=dict(env, LAFS_BENCH_SHARE_GPU="1"), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 4
    assert line["debug_flags"] == 0 and line["roofline"]["kernel"].startswith(("wgrad_kernel", "gemm_nt_kernel", "gemm_kres_kernel", "mlp_fused_kernel"))
    r = subprocess.run([sys.executable, bench, "--gpus", "2"] + small, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "GPU(s) are visible" in r.stderr                      # a 1-GPU box cannot run 2 RCCL ranks
    r = subprocess.run([sys.executable, bench, "--gpus", "2"] + small, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr
    r = subprocess.run([sys.executable, bench] + small, env=dict(env, LAFS_DEBUG_FLAGS="16"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "debug flags" in r.stderr


def test_bench_eight_rank_launch_is_ready_for_an_eight_gpu_node():
    """The launch the driver uses on an 8-GPU node -- `bench.py --gpus 8` -- at toy dimensions with all eight ranks on the one GPU over
    gloo (LAFS_BENCH_SHARE_GPU=1; RCCL wants one device per rank): eight rank processes, segmented graphs with the gradient / center
    all-reduces between them, barrier + max-over-ranks timing, ONE JSON line from rank 0 that reports what the communication backend
    really saw.  No 1 -> 8 scaling curve has been measured on hardware (README); this pins the launch structure, not a rate."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (eight processes time-slicing one GPU pay a context switch per kernel: the trunk is cut to 2 blocks -- 250 s at depth 12)
    small = ["--steps", "2", "--warmup", "1", "--batch", "2", "--arch", "vit_tiny", "--depth", "2", "--out-dim", "1024", "--local-crops", "2",
             "--no-cpu-baseline", "--no-roofline"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LAFS_DEBUG_FLAGS")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"] + small, env=dict(env, LAFS_BENCH_SHARE_GPU="1"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["comm"]["ranks_seen"] == 8 and line["comm"]["backend"] == "gloo"
    assert line["config"]["parallelism"] == "dp8" and line["config"]["global_batch"] == 16 and line["scaling"] == "weak"
    assert line["value"] > 0 and abs(line["value"] - 8 * 2 * 4 / (line["ms_per_step"] * 1e-3)) < 1e-3 * line["value"]      # whole-job crops/s


# ------------------------------------------------------------------------------------------------ Part-fViT as the LAFS pair
def _build_partfvit(fx, use_graph, dropout=0.0, drop_path=0.0):
    from lafs_cvpr2024_amd.face_pre_pro.ViT_face import ViT_face_landmark_patch8
    mk = lambda: ViT_face_landmark_patch8(loss_type="None", GPU_ID=None, num_class=10, image_size=112, patch_size=8, dim=64, depth=2,
                                          heads=2, num_patches=196, mlp_dim=128, dropout=dropout, emb_dropout=dropout, with_land=False,
                                          drop_path_rate=drop_path)
    student = MultiCropWrapper(mk(), vits.DINOHead(64, 256, hidden_dim=64, bottleneck_dim=32, norm_last_layer=True))
    teacher = MultiCropWrapper(mk(), vits.DINOHead(64, 256, hidden_dim=64, bottleneck_dim=32))
    init = sub(fx, "init.")
    student.load_state_dict(init); teacher.load_state_dict(init)
    crit = DINOLoss(256, 4, 0.07, 0.04, 3, 10)
    eng = LafsPretrainEngine(student, teacher, crit, 2, n_local=2, clip_grad=3.0, freeze_last_layer=1, use_graph=use_graph, device=DEV)
    return student, teacher, crit, eng


@pytest.mark.parametrize("use_graph", [False, True])
def test_f16_partfvit_pair_two_steps_against_reference(use_graph, tmp_path):
    """The reference's real LAFS configuration (arch 'mynet', lafs_train.py:300-335, 538-583): ViT_face_landmark_patch8 student and
    teacher fed [B, n, 192] patch tokens, two full steps (clip, frozen last layer in step 0, AdamW, EMA, center) against the
    fixture generated by the reference itself (rates 0).  Then the stage-2 -> stage-3 hand-off: the saved `teacher` initialises a
    fine-tune backbone through load_ssl_teacher with every trunk tensor matched, and a DINO-ViT teacher is rejected."""
    fx = load_golden("f16_lafs_step_partfvit")
    student, teacher, crit, eng = _build_partfvit(fx, use_graph)
    lrs, wds, moms = fx["hyper"].tolist()
    tt = crit.teacher_temp_schedule
    worst = 0.0
    for s in range(2):
        crops = [fx[f"s{s}.crop{i}"] for i in range(4)]                     # 3-D [B, n, 192] tokens, as the reference feeds them
        loss = eng.step(crops, lr=lrs[s], wd=wds[s], momentum=moms[s], teacher_temp=float(tt[s]), epoch=s)
        torch.cuda.synchronize()
        ref_loss = float(fx[f"s{s}.loss"])
        assert abs(float(loss.item()) - ref_loss) / ref_loss < 3e-3, (float(loss.item()), ref_loss)
        assert rel_l2(eng.logits_s[:, :256], fx[f"s{s}.s_out"]) < 2e-2
        assert rel_l2(eng.logits_t[:, :256], fx[f"s{s}.t_out"]) < 2e-2
        torch.testing.assert_close(crit.center.cpu(), fx[f"s{s}.center"], rtol=0, atol=2e-3 * float(fx[f"s{s}.t_out"].abs().max()))
        post = sub(fx, f"s{s}.grad_post.")
        norms = dict(zip([str(n) for n in fx["norm_names"]], fx[f"s{s}.norms"].tolist()))
        bad = {}
        for k, g in post.items():
            mine = dict(student.named_parameters())[k].grad
            clip = min(1.0, 3.0 / (norms[k] + 1e-6))
            if float(g.abs().max()) > 1e-6:
                e = rel_l2(mine * clip, g)
                worst = max(worst, e)
                if e > GRAD_GATE:
                    bad[k] = e
        assert not bad, bad
        for prefix, mod in (("student", student), ("teacher", teacher)):
            e = torch.cat([(mod.state_dict()[k].cpu().double() - v.double()).abs().flatten() for k, v in sub(fx, f"s{s}.{prefix}.").items()]).numpy()
            scale = lrs[s] if prefix == "student" else lrs[s] * (1 - moms[s]) * 2
            assert np.median(e) < 0.05 * scale and np.quantile(e, 0.9) < 0.6 * scale, (prefix, np.median(e), np.quantile(e, 0.9), scale)
    print(f"F16 worst per-tensor gradient rel-L2 error: {worst:.3e} (gate {GRAD_GATE})")
    # ---- hand-off to stage 3 (train_largescale.py:639-657)
    from lafs_cvpr2024_amd.face_pre_pro.ViT_face import ViT_face_landmark_patch8
    from lafs_cvpr2024_amd.train_largescale import load_ssl_teacher
    ck = tmp_path / "checkpoint.pth"
    torch.save({"teacher": teacher.state_dict(), "student": {"module." + k: v for k, v in student.state_dict().items()}}, ck)
    ft = ViT_face_landmark_patch8(loss_type="CosFace", GPU_ID=None, num_class=77, image_size=112, patch_size=8, dim=64, depth=2, heads=2,
                                  mlp_dim=128, with_land=False)
    before = {k: v.clone() for k, v in ft.state_dict().items()}
    load_ssl_teacher(ft, str(ck))
    want = [str(k) for k in fx["teacher_backbone_keys"]]                # what the reference's own teacher checkpoint holds
    assert set(want) <= set(ft.state_dict()) | {"fc", "head"}
    for k, v in teacher.state_dict().items():
        if k.startswith("backbone."):
            assert torch.equal(ft.state_dict()[k[len("backbone."):]].cpu(), v.cpu()), k
    assert torch.equal(ft.state_dict()["loss.weight"], before["loss.weight"])          # the margin head is not in the SSL checkpoint
    # a DINO-ViT teacher shares (almost) no key with Part-fViT: strict=False would silently keep the random init
    vt = MultiCropWrapper(vits.VisionTransformer(img_size=[112], patch_size=8, embed_dim=64, depth=2, num_heads=1, qkv_bias=True, norm_layer=LN6),
                          vits.DINOHead(64, 256, hidden_dim=64, bottleneck_dim=32))
    torch.save({"teacher": vt.state_dict()}, ck)
    with pytest.raises(RuntimeError, match="trunk tensors"):
        load_ssl_teacher(ft, str(ck))


def test_partfvit_pair_with_live_dropout_and_droppath_is_graph_captured():
    """The reference never puts its Part-fViT teacher in eval mode: dropout 0.1 and DropPath 0.1 are live in BOTH networks
    (lafs_train.py:300-335, ViT_face.py:126-153,614).  The element-dropout seed is derived on the device from hyper[HP_STEP], so
    the pair runs as ONE captured graph: losses are finite and step-dependent, the teacher output differs between two replays on
    the same input (stochastic teacher, new masks per replay), and the captured step equals the eager step launch for launch."""
    fx = load_golden("f16_lafs_step_partfvit")
    crops = [fx[f"s0.crop{i}"] for i in range(4)]
    out = {}
    for use_graph in (True, False):
        student, teacher, crit, eng = _build_partfvit(fx, use_graph, dropout=0.1, drop_path=0.1)
        assert eng.use_graph is use_graph and eng.has_dropout
        l0 = float(eng.step(crops, lr=1e-3, wd=0.04, momentum=0.9, teacher_temp=0.05, epoch=1).item())
        t0 = eng.logits_t.clone()
        l1 = float(eng.step(crops, lr=1e-3, wd=0.04, momentum=0.9, teacher_temp=0.05, epoch=1).item())
        assert math.isfinite(l0) and math.isfinite(l1) and l0 != l1
        assert not torch.equal(t0, eng.logits_t)
        out[use_graph] = (l0, l1, t0)
        if use_graph:
            assert eng._graphs is not None and len(eng._graphs) == 1
    # same seeds, same step counters -> the same masks in the captured and the eager run
    assert abs(out[True][0] - out[False][0]) < 1e-5 * abs(out[False][0]), (out[True][:2], out[False][:2])
    assert abs(out[True][1] - out[False][1]) < 2e-3 * abs(out[False][1]), (out[True][:2], out[False][:2])
    torch.testing.assert_close(out[True][2], out[False][2], rtol=1e-4, atol=1e-4)


def test_partfvit_pair_dropout_masks_match_the_oracle_inside_the_engine():
    """The LAFS engine's element dropout against the oracle: the masks of student and teacher are exported for the seeds the
    engine derives (network seed + 7919 * step, sites seed + 3 l + {0, 1, 2}, embedding site) and fed to oracle.partfvit (whose
    dropout sites are pinned to the reference by F14); DropPath off so that dropout is the only stochastic part; loss to 3e-3
    (K = 256 toy head), logits rel-L2 2e-2."""
    from lafs_cvpr2024_amd import functional as Fn, ops
    from oracle import dino, partfvit, vit as ovit
    fx = load_golden("f16_lafs_step_partfvit")
    p = 0.1
    student, teacher, crit, eng = _build_partfvit(fx, False, dropout=p, drop_path=0.0)
    crops = [fx[f"s0.crop{i}"] for i in range(4)]
    eng.step_count = 5                                                    # any step: the seed follows the device counter
    loss = float(eng.step(crops, lr=0.0, wd=0.0, momentum=1.0, teacher_temp=0.05, epoch=1).item())
    B, D, H = 2, 64, 128
    cfg = partfvit.PartFViTConfig(dim=64, depth=2, heads=2, mlp_dim=128, num_patches=196)
    init = sub(fx, "init.")
    Pb = {k[len("backbone."):]: v for k, v in init.items() if k.startswith("backbone.")}
    Ph = {k[len("head."):]: v for k, v in init.items() if k.startswith("head.")}

    def masks_for(seed, rows0, n_img, n_tok, total_rows):
        """oracle mask dict of one crop group: rows [rows0, rows0 + n_img * n_tok) of the packed batch's masks."""
        eff = (seed + 7919 * 5) & 0xFFFFFFFF
        cut = lambda m, C: m[rows0:rows0 + n_img * n_tok].view(n_img, n_tok, C).cpu()
        out = {"emb": cut(ops.dropout_mask(total_rows, D, p, (eff + Fn.EMB_DROP_SITE) & 0xFFFFFFFF, DEV), D)}
        for l in range(2):
            out[(l, 0)] = cut(ops.dropout_mask(total_rows, D, p, eff + 3 * l + 0, DEV), D)
            out[(l, 1)] = cut(ops.dropout_mask(total_rows, H, p, eff + 3 * l + 1, DEV), H)
            out[(l, 2)] = cut(ops.dropout_mask(total_rows, D, p, eff + 3 * l + 2, DEV), D)
        return out
    Ts, Tt = eng.geom_s.n_tok, eng.geom_t.n_tok
    with torch.no_grad():
        t_feat = partfvit.forward_embedding(Pb, torch.cat(crops[:2]), cfg, masks=masks_for(eng.dropout_seed_t, 0, 2 * B, 197, Tt))
        t_out = ovit.dino_head_forward(Ph, t_feat)
        s_feat = torch.cat([partfvit.forward_embedding(Pb, torch.cat(crops[:2]), cfg, masks=masks_for(eng.dropout_seed_s, 0, 2 * B, 197, Ts)),
                            partfvit.forward_embedding(Pb, torch.cat(crops[2:]), cfg,
                                                       masks=masks_for(eng.dropout_seed_s, 2 * B * 197, 2 * B, 37, Ts))])
        s_out = ovit.dino_head_forward(Ph, s_feat)
        ref = float(dino.dino_loss(s_out, t_out, torch.zeros(1, 256), 4, 0.05, 0.1))
    assert rel_l2(eng.logits_t[:, :256], t_out) < 2e-2 and rel_l2(eng.logits_s[:, :256], s_out) < 2e-2
    assert abs(loss - ref) / ref < 3e-3, (loss, ref)
    # and without the masks the oracle is visibly elsewhere (the comparison is not vacuous)
    with torch.no_grad():
        t_plain = ovit.dino_head_forward(Ph, partfvit.forward_embedding(Pb, torch.cat(crops[:2]), cfg))
    assert rel_l2(eng.logits_t[:, :256], t_plain) > 3e-2


# ------------------------------------------------------------------------------------------------ checkpoint layout / resume
def _fresh_pair(drop_path=0.1):
    torch.manual_seed(21)
    mk = lambda dpr: vits.VisionTransformer(img_size=[112], patch_size=8, embed_dim=64, depth=3, num_heads=1, qkv_bias=True,
                                            drop_path_rate=dpr, norm_layer=LN6)
    student = MultiCropWrapper(mk(drop_path), vits.DINOHead(64, 512, hidden_dim=128, bottleneck_dim=64, norm_last_layer=True))
    teacher = MultiCropWrapper(mk(0.0), vits.DINOHead(64, 512, hidden_dim=128, bottleneck_dim=64))
    teacher.load_state_dict(student.state_dict())
    return student, teacher


def test_optimizer_state_is_a_torch_adamw_state_dict():
    """checkpoint['optimizer'] must be what the reference stores and restores: a torch.optim.AdamW(get_params_groups(student))
    state_dict (lafs_train.py:385-392, 428-463; utils.py:152-184).  torch's own optimizer must load it, hold the engine's moments
    and step counts (the last layer, frozen during step 0, is one step behind), and the engine must read its own output back."""
    from lafs_cvpr2024_amd.utils import get_params_groups
    student, teacher = _fresh_pair()
    crit = DINOLoss(512, 4, 0.07, 0.04, 3, 10)
    eng = LafsPretrainEngine(student, teacher, crit, 2, n_local=2, use_graph=True, device=DEV)
    g = torch.Generator().manual_seed(0)
    crops = [torch.randn(2, 3, 112, 112, generator=g) for _ in range(2)] + [torch.randn(2, 3, 48, 48, generator=g) for _ in range(2)]
    for ep in (0, 1):
        eng.step(crops, lr=1e-3, wd=0.05, momentum=0.99, teacher_temp=0.05, epoch=ep)
    torch.cuda.synchronize()
    sd = eng.optimizer_state_dict()
    ref_model, _ = _fresh_pair()
    opt = torch.optim.AdamW(get_params_groups(ref_model))
    opt.load_state_dict(sd)                                           # the reference's restart_from_checkpoint does exactly this
    named = [(n, p) for n, p in ref_model.named_parameters() if p.requires_grad]
    reg = [n for n, p in named if not (n.endswith(".bias") or p.dim() == 1)]
    order = reg + [n for n, _ in named if n not in reg]
    flat = [p for gq in opt.param_groups for p in gq["params"]]
    assert len(flat) == len(order) and opt.param_groups[1]["weight_decay"] == 0.0 and abs(opt.param_groups[0]["weight_decay"] - 0.05) < 1e-7
    sa = eng.sa
    for name, p in zip(order, flat):
        st = opt.state[p]
        assert float(st["step"]) == (1.0 if "last_layer" in name else 2.0), name
        assert torch.equal(st["exp_avg"], sa.view(sa.exp_avg, name, p.shape).cpu())
        assert torch.equal(st["exp_avg_sq"], sa.view(sa.exp_avg_sq, name, p.shape).cpu())
    m0, v0, s0 = sa.exp_avg.clone(), sa.exp_avg_sq.clone(), sa.seg_step.clone()
    sa.exp_avg.fill_(7.0); sa.exp_avg_sq.fill_(7.0); sa.seg_step.fill_(9)
    eng.load_optimizer_state_dict(opt.state_dict())                   # and back, through torch's own re-serialisation
    assert torch.equal(sa.exp_avg, m0) and torch.equal(sa.exp_avg_sq, v0)
    trainable = torch.tensor([p.requires_grad for p in sa.params], device=DEV)
    assert torch.equal(sa.seg_step[trainable], s0[trainable])


@pytest.mark.parametrize("use_graph", [True, False])
def test_checkpoint_resume_continues_the_uninterrupted_run(use_graph, tmp_path):
    """3 steps -> checkpoint.pth (reference layout, lafs_train.py:451-460) -> brand-new modules + engine (new arenas, new
    graphs) -> _load_checkpoint -> step 4 must equal step 4 of the uninterrupted run: loss, student master weights, both AdamW
    moments, per-tensor step counters, teacher, center.  The device RNG is re-seeded before step 4 on both sides (DropPath masks;
    the reference does not checkpoint RNG state either).  Equality is up to the order of the fp32 atomics that accumulate bias /
    LayerNorm gradients (1e-6 relative); everything written without atomics is compared bit for bit through the loss."""
    from lafs_cvpr2024_amd.lafs_train import _load_checkpoint
    g = torch.Generator().manual_seed(5)
    batches = [[torch.randn(2, 3, 112, 112, generator=g) for _ in range(2)] + [torch.randn(2, 3, 48, 48, generator=g) for _ in range(2)]
               for _ in range(4)]
    hp = lambda it: dict(lr=1e-3 * (1 + it), wd=0.04 + 0.01 * it, momentum=0.99, teacher_temp=0.05, epoch=1 if it else 0)

    def run(n_from, eng):
        out = None
        for it in range(n_from, 4):
            if it == 3:
                torch.cuda.manual_seed(99)
            out = eng.step(batches[it], **hp(it))
        torch.cuda.synchronize()
        return float(out.item())

    student, teacher = _fresh_pair()
    crit = DINOLoss(512, 4, 0.07, 0.04, 3, 10)
    eng = LafsPretrainEngine(student, teacher, crit, 2, n_local=2, use_graph=use_graph, device=DEV)
    for it in range(3):
        eng.step(batches[it], **hp(it))
    torch.cuda.synchronize()
    ck = tmp_path / "checkpoint.pth"
    torch.save({"student": {"module." + k: v.cpu() for k, v in student.state_dict().items()},
                "teacher": {k: v.cpu() for k, v in teacher.state_dict().items()},
                "optimizer": eng.optimizer_state_dict(), "epoch": 1, "dino_loss": {k: v.cpu() for k, v in crit.state_dict().items()}}, ck)
    loss_a = run(3, eng)
    snap_a = [t.clone() for t in (eng.sa.master, eng.sa.exp_avg, eng.sa.exp_avg_sq, eng.ta.master, crit.center)]
    steps_a = eng.sa.seg_step.clone()

    student2, teacher2 = _fresh_pair()
    with torch.no_grad():                                             # a different init: everything must come from the file
        for p in list(student2.parameters()) + list(teacher2.parameters()):
            p.add_(0.123)
    crit2 = DINOLoss(512, 4, 0.07, 0.04, 3, 10)
    eng2 = LafsPretrainEngine(student2, teacher2, crit2, 2, n_local=2, use_graph=use_graph, device=DEV)
    state = {"epoch": 0}
    _load_checkpoint(str(ck), student2, teacher2, crit2, eng2, state)
    assert state["epoch"] == 1
    loss_b = run(3, eng2)
    assert abs(loss_a - loss_b) <= 1e-6 * abs(loss_a), (loss_a, loss_b)
    assert torch.equal(steps_a, eng2.sa.seg_step)
    lr4 = hp(3)["lr"]
    for name, a, b in zip(("student", "exp_avg", "exp_avg_sq", "teacher", "center"), snap_a,
                          (eng2.sa.master, eng2.sa.exp_avg, eng2.sa.exp_avg_sq, eng2.ta.master, crit2.center)):
        d = (a - b).abs()
        scale = float(a.abs().max()) + 1e-30
        # bias / LayerNorm gradients are accumulated with fp32 atomics: their summation order differs from run to run, and
        # Adam turns a sign flip of a round-off-sized gradient into a step of up to 2 lr.  Everything else is bit-identical.
        assert float((d > 1e-6 * scale).float().mean()) < 0.02, (name, float((d > 1e-6 * scale).float().mean()))
        assert float(d.max()) <= 2.2 * lr4 + 1e-6 * scale, (name, float(d.max()))


_CHAINS_SNIPPET = r'''
import sys, torch
from functools import partial
import torch.nn as nn
import lafs_cvpr2024_amd.vision_transformer as vits
from lafs_cvpr2024_amd.utils import MultiCropWrapper
from lafs_cvpr2024_amd.dino_loss import DINOLoss
from lafs_cvpr2024_amd.engine import LafsPretrainEngine
torch.manual_seed(11)
B, K, nl = 14, 2048, 8                      # 2 x 14 x 197 = 5516 and 8 x 14 x 37 = 4144 token rows: both groups above the split threshold
LN6 = partial(nn.LayerNorm, eps=1e-6)
mk = lambda: vits.VisionTransformer(img_size=[224], patch_size=8, embed_dim=128, depth=3, num_heads=2, qkv_bias=True, norm_layer=LN6,
                                    drop_path_rate=0.1)
student = MultiCropWrapper(mk(), vits.DINOHead(128, K, hidden_dim=256, bottleneck_dim=64))
teacher = MultiCropWrapper(mk(), vits.DINOHead(128, K, hidden_dim=256, bottleneck_dim=64))
teacher.load_state_dict(student.state_dict())
crops = [torch.randn(B, 3, 112, 112).clamp(-1, 1) for _ in range(2)] + [torch.randn(B, 3, 48, 48).clamp(-1, 1) for _ in range(nl)]
eng = LafsPretrainEngine(student, teacher, DINOLoss(K, 2 + nl, 0.07, 0.04, 3, 10), B, n_local=nl, use_graph=True, device="cuda")
losses = [float(eng.step(crops, lr=1e-3, wd=0.04, momentum=0.9, teacher_temp=0.05, epoch=1).item()) for _ in range(4)]
torch.save({"losses": losses, "teacher": {k: v.cpu() for k, v in teacher.state_dict().items()},
            "student": {k: v.cpu() for k, v in student.state_dict().items()}}, sys.argv[1])
print("CHAINS_DONE")
'''


def test_row_chains_equal_the_single_chain(tmp_path):
    """The trunk passes as two chains of launches over the row ranges of the two crop-resolution groups (csrc/engine.hip:
    row_ranges; LAFS_ROW_CHAINS, read once per process) against ONE chain over all rows: four captured steps from the same
    initialisation -- and the two-chain run a second time, which must repeat BIT FOR BIT (losses and every weight), as must the run
    with every LayerNorm 1 launched on its own instead of written by the previous block's fused MLP; the first loss is identical, the later ones and the weights agree up to the grouping of the LayerNorm / bias
    gradients' partial sums (per chain, then folded in a fixed order) as Adam's first steps amplify it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for mode in ("2", "0", "2 again", "2 ln1 launched"):
        f = str(tmp_path / f"chains{mode[0]}{len(mode)}.pt")
        env = dict(os.environ, LAFS_ROW_CHAINS=mode[0], PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        if mode == "2 ln1 launched":                     # LAFS_OPT_MLP_FUSED without bit 64: every LayerNorm 1 as its own launch
            env["LAFS_MLP_FUSED"] = "15"
        r = subprocess.run([sys.executable, "-c", _CHAINS_SNIPPET, f], env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0 and "CHAINS_DONE" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        out[mode] = torch.load(f)
    # the SAME configuration in another process: the captured step has no fp32 atomic left -- four steps repeat bit for bit
    assert out["2"]["losses"] == out["2 again"]["losses"], (out["2"]["losses"], out["2 again"]["losses"])
    for name in ("teacher", "student"):
        for k, v in out["2"][name].items():
            assert torch.equal(v, out["2 again"][name][k]), (name, k)
    # LayerNorm 1 of blocks 1.. written by the previous block's fused MLP (the default) or by lafs_layernorm_fwd: the same bits
    assert out["2"]["losses"] == out["2 ln1 launched"]["losses"], (out["2"]["losses"], out["2 ln1 launched"]["losses"])
    for name in ("teacher", "student"):
        for k, v in out["2"][name].items():
            assert torch.equal(v, out["2 ln1 launched"][name][k]), (name, k)
    la, lb = out["2"]["losses"], out["0"]["losses"]
    print("[row-chains] relative loss differences, steps 0-3: " + " ".join(f"{abs(a - b) / abs(b):.2e}" for a, b in zip(la, lb)))
    assert abs(la[0] - lb[0]) < 1e-6 * abs(lb[0]), (la, lb)            # same weights: the forward is the same arithmetic
    # one update apart: every gradient sum of the step is folded in a fixed order per configuration (LayerNorm / bias column sums since
    # round 5; round 6 removed the last two fp32-atomic producers -- the position / cls sums of lafs_embed_bwd and the patch embedding's
    # weight gradient -- after measuring that one configuration did NOT repeat run to run: step-1 losses 3.7915149 x5 and 3.7904572 x1,
    # tools/lab/chains_probe.sh).  Observed between the two configurations: 2.1e-6
    assert abs(la[1] - lb[1]) < 2e-5 * abs(lb[1]), (la, lb)
    for a, b in zip(la[2:], lb[2:]):                                     # then Adam's first steps (lr * sign of a near-zero gradient)
        assert abs(a - b) < 5e-3 * abs(b), (la, lb)                      # amplify that noise; a race would be orders above this
    for name in ("teacher", "student"):
        for k, v in out["0"][name].items():
            dw = (out["2"][name][k].float() - v.float()).abs()
            assert float(dw.max()) <= 4 * 2 * 1e-3 + 1e-6, (name, k)    # at most every step's update flipped: 4 steps x 2 lr
            assert float(dw.median()) <= 2e-5 * (1.0 + float(v.float().abs().max())), (name, k)


def test_two_engines_on_two_contexts_and_two_caller_streams_equal_the_single_engine_runs():
    """The boundary's own claim (include/lafs_hip.h: "calls that share no context and no buffer are independent"): two LAFS engines in
    one process, each with its own lafs_ctx (side streams, fork / join events, event pool), each with TWO row chains (two crop-
    resolution groups of >= 4096 rows), stepped eagerly from two host threads on two caller streams at the same time -- against the
    same two runs made one after the other.  Before round 5 the side streams and events were process-wide statics of the library:
    two engines issued from two threads would have recorded / waited on the same `fork` / `join` events."""
    import ctypes as C
    import threading
    from lafs_cvpr2024_amd import _lib
    B, K, nl = 14, 512, 8

    def build(seed):
        torch.manual_seed(0)
        mk = lambda: vits.VisionTransformer(img_size=[112], patch_size=8, embed_dim=128, depth=2, num_heads=2, qkv_bias=True, norm_layer=LN6)
        student = MultiCropWrapper(mk(), vits.DINOHead(128, K, hidden_dim=128, bottleneck_dim=64))
        teacher = MultiCropWrapper(mk(), vits.DINOHead(128, K, hidden_dim=128, bottleneck_dim=64))
        teacher.load_state_dict(student.state_dict())
        crit = DINOLoss(K, 2 + nl, 0.07, 0.04, 3, 10)
        eng = LafsPretrainEngine(student, teacher, crit, B, n_local=nl, use_graph=False, device=DEV)
        g = torch.Generator().manual_seed(seed)
        crops = [torch.randn(B, 3, 112, 112, generator=g).clamp(-1, 1).to(DEV) for _ in range(2)] + \
                [torch.randn(B, 3, 48, 48, generator=g).clamp(-1, 1).to(DEV) for _ in range(nl)]
        return eng, teacher, crops

    def run(eng, crops, stream, out):
        with torch.cuda.stream(stream):
            out["losses"] = [eng.step(crops, lr=1e-3, wd=0.04, momentum=0.9, teacher_temp=0.05, epoch=1).clone() for _ in range(3)]
        stream.synchronize()
        out["losses"] = [float(x.item()) for x in out["losses"]]

    # reference: the two runs one after the other
    ref = {}
    for seed in (1, 2):
        eng, teacher, crops = build(seed)
        assert _lib.lib().lafs_trunk_row_ranges(C.byref(Fn_desc(eng))) == 2
        o = {}
        run(eng, crops, torch.cuda.Stream(), o)
        ref[seed] = (o["losses"], eng.ta.master.clone(), eng.sa.master.clone())
    # concurrent: two engines, two contexts, two threads, two streams
    pair = {seed: build(seed) for seed in (1, 2)}
    assert pair[1][0].ctx.handle != pair[2][0].ctx.handle
    outs, threads = {1: {}, 2: {}}, []
    streams = {1: torch.cuda.Stream(), 2: torch.cuda.Stream()}
    torch.cuda.synchronize()
    for seed in (1, 2):
        t = threading.Thread(target=run, args=(pair[seed][0], pair[seed][2], streams[seed], outs[seed]))
        t.start(); threads.append(t)
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    for seed in (1, 2):
        l_ref, t_ref, s_ref = ref[seed]
        l = outs[seed]["losses"]
        assert abs(l[0] - l_ref[0]) <= 1e-6 * abs(l_ref[0]), (seed, l, l_ref)              # same weights, same arithmetic
        for a, b in zip(l[1:], l_ref[1:]):
            assert abs(a - b) <= 5e-3 * abs(b), (seed, l, l_ref)                            # (position-table gradient atomics through Adam's first steps)
        eng = pair[seed][0]
        for mine, want in ((eng.ta.master, t_ref), (eng.sa.master, s_ref)):
            d = (mine - want).abs()
            assert float(d.max()) <= 3 * 2.1e-3 and float((d > 1e-5).float().mean()) < 2e-2, (seed, float(d.max()))


def Fn_desc(eng):
    """The student trunk's C descriptor as the engine builds it (for lafs_trunk_row_ranges)."""
    from lafs_cvpr2024_amd import functional as Fn
    return Fn.make_trunk_desc(eng.sa, eng.spec_s.trunk, eng.geom_s, None, with_grad=True)
