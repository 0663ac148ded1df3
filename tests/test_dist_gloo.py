"""world_size = 2 on CPU (gloo): the data-parallel semantics of the LAFS step -- gradient mean and center all-reduce --
through the same FlatReducer the engine uses, checked against a single-process run on the full batch."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import step as ostep, vit as ovit


def _make(seed=0):
    cfg = ovit.ViTConfig(patch_size=8, embed_dim=64, depth=1, num_heads=1, img_size=112)
    return cfg, ostep.LafsState(cfg, out_dim=256, seed=seed, hidden_dim=64, bottleneck_dim=32)


def _crops(B):
    g = torch.Generator().manual_seed(7)
    return [torch.randn(B, 3, 112, 112, generator=g).clamp(-1, 1) for _ in range(2)] + [torch.randn(B, 3, 48, 48, generator=g).clamp(-1, 1)]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from lafs_cvpr2024_amd.distributed import FlatReducer, center_from_colsum, world_size
    assert world_size() == world
    red = FlatReducer()

    def all_reduce(t):                       # the engine's reducer: async SUM launches, one wait
        red.launch(t); red.wait_all()

    cfg, st = _make()
    full = _crops(4)
    mine = [c[rank * 2:(rank + 1) * 2] for c in full]
    r = ostep.lafs_step(st, mine, epoch=1, lr=1e-3, wd=0.04, momentum=0.9, teacher_temp=0.05, world_size=world, all_reduce=all_reduce)
    # center through the helper == oracle's update_center
    colsum = r["teacher_out"].sum(0, keepdim=True); all_reduce(colsum)
    c2 = center_from_colsum(torch.zeros(1, 256), colsum, r["teacher_out"].shape[0], 0.9)
    assert torch.allclose(c2, st.center, atol=1e-7)
    if rank == 0:
        torch.save({"student": st.student, "center": st.center, "loss": r["loss"]}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process_full_batch(tmp_path):
    out = str(tmp_path / "r0.pt")
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=False)
    cfg, st = _make()
    ostep.lafs_step(st, _crops(4), epoch=1, lr=1e-3, wd=0.04, momentum=0.9, teacher_temp=0.05)
    torch.testing.assert_close(got["center"], st.center, rtol=1e-5, atol=1e-7)
    errs = torch.cat([(got["student"][k] - v).abs().flatten() for k, v in st.student.items()])
    # identical up to fp32 summation order, except Adam's amplification where the gradient is round-off
    assert float(errs.median()) < 1e-7 and float((errs > 1e-5).float().mean()) < 0.02 and float(errs.max()) < 2.5e-3


_PFC_LAMS = (0.3, 0.85)


def _pfc_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from lafs_cvpr2024_amd.partial_fc import shard_range
    from oracle import margin
    g = torch.Generator().manual_seed(3)
    C, D, B = 37, 16, 4
    Wfull = torch.randn(C, D, generator=g)
    emb = torch.randn(world * B, D, generator=g)
    lab = torch.randint(0, C, (world * B,), generator=g)
    start, n = shard_range(C, rank, world)

    def gather(t):
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t.contiguous())
        return torch.cat(parts)

    def rmax(t):
        dist.all_reduce(t, op=dist.ReduceOp.MAX); return t

    def rsum(t):
        dist.all_reduce(t); return t

    loss, dE = margin.partial_fc_sharded(emb[rank * B:(rank + 1) * B], lab[rank * B:(rank + 1) * B], Wfull[start:start + n], start,
                                         all_gather=gather, all_reduce_max=rmax, all_reduce_sum=rsum)
    dist.all_reduce(dE)                       # gloo has no reduce_scatter: all-reduce and slice (same sum)
    mine = dE[rank * B:(rank + 1) * B]
    # soft (mixup) targets: every rank mixes its own batch with its flip and draws its own lambda (util/mixup_my.py:189-200)
    lab_r = lab[rank * B:(rank + 1) * B]
    loss_s, dE_s = margin.partial_fc_sharded(emb[rank * B:(rank + 1) * B], lab_r, Wfull[start:start + n], start, all_gather=gather,
                                             all_reduce_max=rmax, all_reduce_sum=rsum, labels2_local=lab_r.flip(0), lam_local=_PFC_LAMS[rank])
    dist.all_reduce(dE_s)
    torch.save({"loss": loss, "demb": mine, "loss_soft": loss_s, "demb_soft": dE_s[rank * B:(rank + 1) * B]}, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_partial_fc_two_shards_equal_unsharded(tmp_path):
    """The class-sharded softmax exchange (MAX, SUM, SUM; reduce-scatter of dE) reproduces the unsharded CosFace + CE -- with hard
    labels and with the reference's soft (mixup) targets (two (class, weight) pairs per row, a lambda per rank)."""
    from oracle import margin
    out = str(tmp_path / "pfc")
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    mp.spawn(_pfc_worker, args=(2, port, out), nprocs=2, join=True)
    g = torch.Generator().manual_seed(3)
    C, D, B = 37, 16, 4
    Wfull = torch.randn(C, D, generator=g)
    emb = torch.randn(2 * B, D, generator=g).requires_grad_(True)
    lab = torch.randint(0, C, (2 * B,), generator=g)
    ref = margin.partial_fc_reference(emb, Wfull, lab)
    ref.backward()
    # the reference's own loss on the same data: dense soft target through CosFace's soft branch (F10-pinned) + soft-target CE
    emb2 = emb.detach().clone().requires_grad_(True)
    tgt = torch.cat([margin.mixup_batch(torch.zeros(B, 1), lab[r * B:(r + 1) * B], C, _PFC_LAMS[r])[1] for r in range(2)])
    ref_s = margin.soft_target_cross_entropy(margin.cosface_logits(emb2, Wfull, tgt), tgt)
    ref_s.backward()
    for r in range(2):
        got = torch.load(out + f".{r}", weights_only=False)
        torch.testing.assert_close(got["loss"], ref.detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(got["demb"], emb.grad[r * B:(r + 1) * B], rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(got["loss_soft"], ref_s.detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(got["demb_soft"], emb2.grad[r * B:(r + 1) * B], rtol=1e-4, atol=1e-6)


def _exchange_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from lafs_cvpr2024_amd import partial_fc as pfc
    calls = []
    real_gather = pfc._gather_rows
    pfc._gather_rows = lambda t, w: (calls.append(tuple(t.shape)), real_gather(t, w))[1]
    B, C = 4, 37
    g = torch.Generator().manual_seed(9)
    lab = torch.randint(0, C, (world * B,), generator=g)
    mine = lab[rank * B:(rank + 1) * B]
    lam = (1.0, 0.3)[rank]                                    # rank 0 drew "no mixup", rank 1 mixes: the advisor's hang scenario
    labels2 = None if lam == 1.0 else mine.flip(0)
    # --- the product's own target normalisation + exchange helpers (the CPU-safe part of PartialFC.forward_backward)
    l2, lam_n, soft = pfc.normalize_targets(world, 0, mine, labels2, lam)
    assert soft and l2 is not None
    L = pfc._gather_rows(mine, world)
    L2 = pfc._gather_rows(l2, world)
    lam_rows = pfc._gather_rows(torch.full((B,), float(lam_n)), world)
    start, n = pfc.shard_range(C, rank, world)
    index, y, y2 = pfc.sample_classes(L, start, n, n, labels2=L2)
    part = torch.arange(world * B * 3, dtype=torch.float32).view(world * B, 3) * (rank + 1)
    rs = pfc._reduce_scatter_rows(part.clone(), world, rank)
    torch.save({"calls": calls, "L": L, "L2": L2, "lam_rows": lam_rows, "y": y, "y2": y2, "start": start, "n": n, "rs": rs}, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_partial_fc_exchange_helpers_issue_the_same_collectives_whatever_each_rank_drew(tmp_path):
    """lafs_cvpr2024_amd/partial_fc.py's own exchange logic on CPU tensors over gloo (the kernels between the collectives need the
    GPU: tests/test_gpu_dp.py): with lambdas (1.0, 0.3) -- rank 0 drew no mixup, rank 1 mixes -- both ranks issue the SAME sequence
    of all-gathers (normalize_targets turns rank 0's hard labels into the soft form with a zero-weight partner), the gathered rows
    are rank-major, rank 0's rows carry lambda 1 and their own class as partner, and the reduce-scatter hands every rank the sum of
    its slice."""
    out = str(tmp_path / "ex")
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    mp.spawn(_exchange_worker, args=(2, port, out), nprocs=2, join=True)
    r = [torch.load(out + f".{k}", weights_only=False) for k in range(2)]
    assert r[0]["calls"] == r[1]["calls"] and len(r[0]["calls"]) == 3
    B, C = 4, 37
    lab = torch.randint(0, C, (2 * B,), generator=torch.Generator().manual_seed(9))
    for k in range(2):
        assert torch.equal(r[k]["L"], lab)
        assert torch.equal(r[k]["L2"], torch.cat([lab[:B], lab[B:].flip(0)]))           # rank 0: own classes; rank 1: its flipped batch
        assert torch.equal(r[k]["lam_rows"], torch.tensor([1.0] * B + [0.3] * B))
        own = (lab >= r[k]["start"]) & (lab < r[k]["start"] + r[k]["n"])
        assert torch.equal(r[k]["y"][own].long(), lab[own] - r[k]["start"]) and bool((r[k]["y"][~own] == -1).all())
        full = torch.arange(2 * B * 3, dtype=torch.float32).view(2 * B, 3) * 3.0       # (1 + 2) x the ramp
        assert torch.equal(r[k]["rs"], full[k * B:(k + 1) * B])
