"""LAFS / DINO pre-training entry point with the reference's flags, launch contract and checkpoint layout
(reference lafs_train.py:30-122 flags, :156-471 train_lafs, :474-623 train_one_epoch), driving the fused HIP engine.

Launch (one process per GPU, RCCL):
    python -m torch.distributed.run --nproc-per-node=N --master-addr 127.0.0.1 lafs_train.py --arch vit_small ...

Differences from the reference, all forced by its own breakage or by scope:
  * --arch defaults to 'mynet' exactly like the reference (lafs_train.py:34): the Part-fViT pair ViT_face_landmark_patch8(dim 768,
    depth 12, heads 11, mlp 2048, dropout = emb_dropout = 0.1) of :300-335 -- the only configuration the reference can run, and
    the one whose `teacher` checkpoint stage 3 (train_largescale.py:639-657) loads.  The DINO vit_* family is offered too (the
    reference lists it but crashes on it: landmarkcnn undefined, :360); its xcit choices need network access and are dropped;
  * the data pipeline (MXNet recordio + PIL augmentations + landmark CNN) is out of scope: `--data synthetic` feeds the
    crop shapes the landmark gather emits (2 x 112^2 + n x 48^2), or pass any Dataset through `train_lafs(args, dataset=...)`
    yielding lists of 2 + n crops;
  * the optimizer is the engine's fused per-tensor-clip + AdamW (the reference's default and only used one).
"""
import argparse
import datetime
import json
import math
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

from . import utils
from . import vision_transformer as vits
from .dino_loss import DINOLoss
from .engine import LafsPretrainEngine
from .vision_transformer import DINOHead

__all__ = ["DINOLoss", "get_args_parser", "train_lafs", "train_one_epoch", "SyntheticCrops"]


# torch.cuda.amp.GradScaler().state_dict() of an unused scaler (its documented defaults): see the checkpoint's 'fp16_scaler' entry
FP16_SCALER_STATE = {"scale": 65536.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000, "_growth_tracker": 0}


def get_args_parser():
    p = argparse.ArgumentParser('LAFS', add_help=False)
    p.add_argument('--arch', default='mynet', type=str, choices=['mynet', 'vit_tiny', 'vit_small', 'vit_base'])
    p.add_argument('--patch_size', default=8, type=int)
    p.add_argument('--out_dim', default=100000, type=int)
    p.add_argument('--norm_last_layer', default=True, type=utils.bool_flag)
    p.add_argument('--momentum_teacher', default=0.996, type=float)
    p.add_argument('--use_bn_in_head', default=False, type=utils.bool_flag)
    p.add_argument('--warmup_teacher_temp', default=0.07, type=float)
    p.add_argument('--teacher_temp', default=0.04, type=float)
    p.add_argument('--warmup_teacher_temp_epochs', default=30, type=int)
    p.add_argument('--use_fp16', type=utils.bool_flag, default=True, help="kept for CLI compatibility: the HIP path always "
                   "multiplies in bf16 with fp32 accumulation and needs no loss scaling")
    p.add_argument('--weight_decay', type=float, default=0.04)
    p.add_argument('--weight_decay_end', type=float, default=0.4)
    p.add_argument('--clip_grad', type=float, default=3.0)
    p.add_argument('--batch_size_per_gpu', default=64, type=int)
    p.add_argument('--epochs', default=41, type=int)
    p.add_argument('--freeze_last_layer', default=1, type=int)
    p.add_argument("--lr", default=0.0005, type=float)
    p.add_argument("--warmup_epochs", default=10, type=int)
    p.add_argument('--min_lr', type=float, default=1e-6)
    p.add_argument('--optimizer', default='adamw', type=str, choices=['adamw'])
    p.add_argument('--drop_path_rate', type=float, default=0.1)
    p.add_argument('--mynet_dims', default='768,12,11,2048', type=str, help="dim,depth,heads,mlp_dim of --arch mynet (reference :300-335)")
    p.add_argument('--mynet_dropout', default=0.1, type=float, help="dropout = emb_dropout of --arch mynet (reference: 0.1)")
    p.add_argument('--local_crops_number', type=int, default=8)
    p.add_argument('--data', default='synthetic', type=str,
                   help="'synthetic': random landmark-crop shaped tensors; 'synthetic_views': the 20 augmented 112x112 views of "
                        "DataAugmentation_LAFS (clean/augmented pairs) pushed through the landmark front-end; "
                        "'synthetic_u8': uint8 112x112 images, augmented ON THE DEVICE (augment.DeviceAugmenter, Pillow-exact) into "
                        "the 20 views, then the landmark front-end -- the whole input pipeline of the reference on the GPU; "
                        "'recordio': the same from --data_path/train.rec (MXNet RecordIO, InsightFace layout; JPEG decode on "
                        "--num_workers CPU workers, everything after it on the device)")
    p.add_argument('--landmark_ckpt', '--landmark_path', dest='landmark_ckpt', default='', type=str,
                   help="state_dict of the frozen landmark CNN (reference --landmark_path, :112, :262-268)")
    # flags of the reference's PIL / recordio input pipeline: parsed for command-line compatibility, unused with synthetic data
    p.add_argument('--data_path', default='', type=str)
    p.add_argument('--num_workers', default=6, type=int)
    p.add_argument('--global_crops_scale', type=float, nargs='+', default=(0.4, 1.))
    p.add_argument('--local_crops_scale', type=float, nargs='+', default=(0.05, 0.4))
    p.add_argument('--steps_per_epoch', default=100, type=int, help="iterations per epoch for --data synthetic")
    p.add_argument('--output_dir', default=".", type=str)
    p.add_argument('--saveckp_freq', default=10, type=int)
    p.add_argument('--seed', default=0, type=int)
    p.add_argument('--use_graph', default=True, type=utils.bool_flag, help="capture the step into hipGraphs")
    p.add_argument("--dist_url", default="env://", type=str)
    p.add_argument("--local_rank", default=0, type=int)
    return p


class SyntheticCrops:
    """Random crops with the geometry the landmark gather emits: 2 global 3x112x112 + n local 3x48x48 in [-1, 1]."""

    def __init__(self, steps, batch, n_local, device, seed):
        self.steps, self.batch, self.n_local, self.device = steps, batch, n_local, device
        self.gen = torch.Generator(device=device).manual_seed(seed)

    def __len__(self):
        return self.steps

    def __iter__(self):
        for _ in range(self.steps):
            g = [torch.randn(self.batch, 3, 112, 112, device=self.device, generator=self.gen).clamp_(-1, 1) for _ in range(2)]
            l = [torch.randn(self.batch, 3, 48, 48, device=self.device, generator=self.gen).clamp_(-1, 1) for _ in range(self.n_local)]
            yield g + l, None


class SyntheticViews:
    """The loader contract of the reference (lafs_train.py:790-886): per batch a list of 2*(2+n_local) tensors [B,3,112,112],
    (clean, augmented) pairs: g0, g0', g1, g1', l0, l0', ...  Random tensors here; the landmark front-end turns them into the
    2 global + n local mosaics."""

    def __init__(self, steps, batch, n_local, device, seed):
        self.steps, self.batch, self.n_local, self.device = steps, batch, n_local, device
        self.gen = torch.Generator(device=device).manual_seed(seed)

    def __len__(self):
        return self.steps

    def __iter__(self):
        for _ in range(self.steps):
            yield torch.randn(2 * (2 + self.n_local), self.batch, 3, 112, 112, device=self.device, generator=self.gen).clamp_(-1, 1), None


class SyntheticU8Views:
    """uint8 images [B,3,112,112] (what the recordio reader yields after decoding, lafs_train.py:176) -> 20 views through the
    device-side DataAugmentation_LAFS."""

    def __init__(self, steps, batch, n_local, device, seed):
        from .augment import DeviceAugmenter
        self.steps, self.batch, self.device = steps, batch, device
        self.gen = torch.Generator(device=device).manual_seed(seed)
        self.aug = DeviceAugmenter(batch, n_local=n_local, device=device, seed=seed)

    def __len__(self):
        return self.steps

    def __iter__(self):
        for _ in range(self.steps):
            u8 = torch.randint(0, 256, (self.batch, 3, 112, 112), device=self.device, dtype=torch.uint8, generator=self.gen)
            yield self.aug(u8), None


def epoch_shard(n, rank, world, seed, epoch):
    """Indices of this rank's samples in one epoch, torch DistributedSampler style: a permutation of range(n) seeded with
    (seed, epoch) -- the same on every rank --, cut to world * (n // world) and dealt out round-robin."""
    g = torch.Generator().manual_seed(int(seed) * 1000003 + int(epoch))
    perm = torch.randperm(n, generator=g).tolist()
    return perm[rank: (n // world) * world: world]


class RecordIOViews:
    """--data_path/train.rec (reference lafs_train.py:157-191: FaceDataset over MXNet recordio + DataLoader) -> decoded uint8
    batches -> device-side DataAugmentation_LAFS.  One pass over the dataset per epoch, sharded over the ranks."""

    def __init__(self, path, batch, n_local, device, seed, num_workers, rank=0, world=1):
        from .augment import DeviceAugmenter
        from .recordio import FaceRecordDataset
        self.ds = FaceRecordDataset(os.path.join(path, 'train.rec'))
        self.all_seq = list(self.ds.seq)
        self.rank, self.world = rank, world
        # torch DistributedSampler semantics (reference lafs_train.py:186-191): every rank gets the SAME number of samples
        # (len // world, the tail dropped), so all ranks run the same number of steps -- unequal shards would leave the last
        # gradient all-reduce of the longer ranks waiting for ever -- and the partition is re-drawn every epoch from a
        # permutation seeded with (seed, epoch), identical on all ranks (set_epoch)
        self.per_rank = len(self.all_seq) // world
        self.batch, self.device, self.seed, self.workers = batch, device, seed, num_workers
        self.aug = DeviceAugmenter(batch, n_local=n_local, device=device, seed=seed)
        self.epoch = 0
        self.set_epoch(0)
        print(f"Data loaded: there are {len(self.all_seq)} images ({self.per_rank} per rank).")

    def set_epoch(self, epoch):
        self.epoch = int(epoch)
        mine = epoch_shard(len(self.all_seq), self.rank, self.world, self.seed - self.rank, self.epoch)   # `seed` arrives as base + rank
        self.ds.seq = [self.all_seq[i] for i in mine]

    def __len__(self):
        return self.per_rank // self.batch

    def __iter__(self):
        from .recordio import device_batches
        for u8, _ in device_batches(self.ds, self.batch, self.device, num_workers=self.workers, shuffle=True,
                                    seed=self.seed + self.epoch):
            yield self.aug(u8), None


def build_landmark_frontend(args, device):
    """Frozen landmark CNN (reference lafs_train.py:241-269: face_landmark_4simmin_glo_loc, eval mode) + the fused front-end."""
    from .face_pre_pro.ViT_face import face_landmark_4simmin_glo_loc
    from .landmark_frontend import LandmarkFrontEnd
    cnn = face_landmark_4simmin_glo_loc(loss_type='None', GPU_ID=None, num_class=10, image_size=112, patch_size=8, dim=768,
                                        depth=12, heads=11, mlp_dim=2048)
    if args.landmark_ckpt:
        sd = torch.load(args.landmark_ckpt, map_location="cpu", weights_only=False)
        sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
        print("=> landmark CNN:", cnn.load_state_dict(sd, strict=False))
    return LandmarkFrontEnd(cnn, args.batch_size_per_gpu, n_local=args.local_crops_number, device=device)


def build_backbones(args):
    """(student backbone, teacher backbone, embed_dim).  'mynet' = the reference's hard-coded Part-fViT pair
    (lafs_train.py:300-335): both networks with dropout / emb_dropout 0.1 and Residual_droppath 0.1, and neither is ever put in
    eval mode, so the teacher is stochastic too.  `--mynet_dims dim,depth,heads,mlp` shrinks it for smoke runs."""
    if args.arch != 'mynet':
        sb = vits.__dict__[args.arch](patch_size=args.patch_size, drop_path_rate=args.drop_path_rate)
        tb = vits.__dict__[args.arch](patch_size=args.patch_size)
        return sb, tb, sb.embed_dim
    from .face_pre_pro.ViT_face import ViT_face_landmark_patch8
    dim, depth, heads, mlp = (int(v) for v in args.mynet_dims.split(","))
    mk = lambda: ViT_face_landmark_patch8(loss_type='CosFace', GPU_ID=None, num_class=30000, image_size=112, patch_size=8, dim=dim,
                                          depth=depth, heads=heads, num_patches=196, mlp_dim=mlp, dropout=args.mynet_dropout,
                                          emb_dropout=args.mynet_dropout, with_land=False, use_standcoord=False, Random_prob=False,
                                          shuffle=False)
    sb, tb = mk(), mk()
    # `loss.weight` (CosFace, 30000 x dim) is part of the checkpoint layout but never used by the SSL step: the reference's
    # AdamW registers it (it holds an index in the regularised param group) but skips it (its gradient is None: no state entry).
    # Here it is frozen in the arena, so the fused optimizer / zeroing / all-reduce leave it alone, and marked so that the
    # checkpoint's optimizer state_dict keeps the reference's parameter indices (engine._adamw_order)
    sb.loss.weight.requires_grad_(False)
    sb.loss.weight._lafs_optimizer_registered = True
    return sb, tb, dim


def train_lafs(args, dataset=None):
    utils.init_distributed_mode(args)
    utils.fix_random_seeds(args.seed)
    print("\n".join("%s: %s" % (k, str(v)) for k, v in sorted(dict(vars(args)).items())))
    device = torch.device("cuda", args.gpu)
    world = utils.get_world_size()

    # ---- student / teacher: MultiCropWrapper(backbone, DINOHead), as reference lafs_train.py:200-356 ----
    student_b, teacher_b, embed_dim = build_backbones(args)
    student = utils.MultiCropWrapper(student_b, DINOHead(embed_dim, args.out_dim, use_bn=args.use_bn_in_head,
                                                         norm_last_layer=args.norm_last_layer))
    teacher = utils.MultiCropWrapper(teacher_b, DINOHead(embed_dim, args.out_dim, args.use_bn_in_head))
    teacher.load_state_dict(student.state_dict())
    for p in teacher.parameters():
        p.requires_grad = False
    dino_loss = DINOLoss(args.out_dim, args.local_crops_number + 2, args.warmup_teacher_temp, args.teacher_temp,
                         args.warmup_teacher_temp_epochs, args.epochs)
    engine = LafsPretrainEngine(student, teacher, dino_loss, args.batch_size_per_gpu, n_local=args.local_crops_number,
                                clip_grad=args.clip_grad, freeze_last_layer=args.freeze_last_layer, use_graph=args.use_graph,
                                device=device)
    print(f"Student and Teacher are built: they are both {args.arch} networks.")

    frontend = None
    if dataset is not None:
        data_loader = dataset
    elif args.data == 'recordio':
        data_loader = RecordIOViews(args.data_path, args.batch_size_per_gpu, args.local_crops_number, device,
                                    args.seed + utils.get_rank(), args.num_workers, utils.get_rank(), world)
        frontend = build_landmark_frontend(args, device)
    elif args.data in ('synthetic_views', 'synthetic_u8'):
        cls = SyntheticViews if args.data == 'synthetic_views' else SyntheticU8Views
        data_loader = cls(args.steps_per_epoch, args.batch_size_per_gpu, args.local_crops_number, device, args.seed + utils.get_rank())
        frontend = build_landmark_frontend(args, device)
    else:
        data_loader = SyntheticCrops(args.steps_per_epoch, args.batch_size_per_gpu, args.local_crops_number, device,
                                     args.seed + utils.get_rank())
    n_it = len(data_loader)
    # ---- schedules (reference :411-424) ----
    lr_schedule = utils.cosine_scheduler(args.lr * (args.batch_size_per_gpu * world) / 256., args.min_lr, args.epochs, n_it,
                                         warmup_epochs=args.warmup_epochs)
    wd_schedule = utils.cosine_scheduler(args.weight_decay, args.weight_decay_end, args.epochs, n_it)
    momentum_schedule = utils.cosine_scheduler(args.momentum_teacher, 1, args.epochs, n_it)

    # ---- resume (reference :428-438; checkpoint keys :451-460) ----
    to_restore = {"epoch": 0}
    ckpt = os.path.join(args.output_dir, "checkpoint.pth")
    if os.path.isfile(ckpt):
        _load_checkpoint(ckpt, student, teacher, dino_loss, engine, to_restore)
    start_epoch = to_restore["epoch"]

    start = time.time()
    print("Starting LAFS training !")
    for epoch in range(start_epoch, args.epochs):
        if hasattr(data_loader, "set_epoch"):
            data_loader.set_epoch(epoch)                      # data_loader.sampler.set_epoch(epoch), reference :440
        stats = train_one_epoch(engine, dino_loss, data_loader, lr_schedule, wd_schedule, momentum_schedule, epoch, args,
                                frontend=frontend)
        save_dict = {
            'student': {"module." + k: v for k, v in student.state_dict().items()},     # DDP-style prefix, as the reference saves
            'teacher': teacher.state_dict(),
            'optimizer': engine.optimizer_state_dict(),
            'epoch': epoch + 1,
            'args': args,
            'dino_loss': dino_loss.state_dict(),
            # the reference stores its GradScaler's state under this key when it trains in fp16 (lafs_train.py:451-460) and hands the
            # key to restart_from_checkpoint(fp16_scaler=...).  This path computes in bf16 (fp32 range: nothing to scale), so the entry
            # is a fresh torch.cuda.amp.GradScaler's state_dict: a reference-side resume finds the key and loads a valid state
            'fp16_scaler': FP16_SCALER_STATE,
        }
        utils.save_on_master(save_dict, ckpt)
        if args.saveckp_freq and epoch % args.saveckp_freq == 0:
            utils.save_on_master(save_dict, os.path.join(args.output_dir, f'checkpoint{epoch:04}.pth'))
        if utils.is_main_process():
            with (Path(args.output_dir) / "log.txt").open("a") as f:
                f.write(json.dumps({**{f'train_{k}': v for k, v in stats.items()}, 'epoch': epoch}) + "\n")
    print('Training time {}'.format(str(datetime.timedelta(seconds=int(time.time() - start)))))
    if dist.is_initialized():
        dist.destroy_process_group()


def _load_checkpoint(path, student, teacher, dino_loss, engine, run_variables):
    print("Found checkpoint at {}".format(path))
    ck = torch.load(path, map_location="cpu", weights_only=False)
    strip = lambda sd: {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
    print("=> student:", student.load_state_dict(strip(ck["student"]), strict=False))
    print("=> teacher:", teacher.load_state_dict(strip(ck["teacher"]), strict=False))
    if "dino_loss" in ck:
        dino_loss.load_state_dict(ck["dino_loss"])
    if "optimizer" in ck:
        engine.load_optimizer_state_dict(ck["optimizer"])
    engine.sa.refresh_shadows(); engine.ta.refresh_shadows()
    run_variables["epoch"] = ck.get("epoch", 0)


class _Pipelined:
    """Software pipeline over the view loader: batch i+1 is fetched (and its `produced` event recorded) BEFORE step i is
    launched and handed to the landmark front-end right AFTER, so the frozen CNN + gathers of the next batch run on the
    front-end stream underneath the training step."""

    def __init__(self, data_loader, frontend, engine):
        self.loader, self.fe, self.engine = data_loader, frontend, engine

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        it = iter(self.loader)
        cur = torch.cuda.current_stream()
        try:
            views, _ = next(it)
        except StopIteration:
            return
        self.fe.prefetch(views)
        while True:
            try:
                nxt, _ = next(it)
                ev = cur.record_event()
            except StopIteration:
                nxt = None
            self.fe.commit(self.engine)                    # staged mosaics -> the engine's (graph-captured) input buffers
            yield None, nxt is not None and (lambda n=nxt, e=ev: self.fe.prefetch(n, produced=e))
            if nxt is None:
                return


def train_one_epoch(engine, dino_loss, data_loader, lr_schedule, wd_schedule, momentum_schedule, epoch, args, frontend=None):
    metric_logger = utils.MetricLogger(delimiter="  ")
    header = 'Epoch: [{}/{}]'.format(epoch, args.epochs)
    tt = float(dino_loss.teacher_temp_schedule[epoch])
    n = len(data_loader)
    source = _Pipelined(data_loader, frontend, engine) if frontend is not None else data_loader
    # The reference stops at the first non-finite loss (lafs_train.py:585-587, a host sync per step).  Here the host polls every 20
    # steps; in between, every step's loss is added to a device-side accumulator (NaN / Inf are sticky in a sum), which is checked at
    # each poll and once more at the end of the epoch -- BEFORE the caller writes the checkpoint: no poisoned step can be saved.
    loss_acc = torch.zeros(1, device=engine.device)

    def stop_if_poisoned(value):
        if not math.isfinite(value):
            print("Loss is {}, stopping training".format(value), force=True)
            sys.exit(1)
    for it, (images, after) in enumerate(metric_logger.log_every(source, 100, header)):
        it = n * epoch + it
        loss = engine.step(images, lr=float(lr_schedule[it]), wd=float(wd_schedule[it]), momentum=float(momentum_schedule[it]),
                           teacher_temp=tt, epoch=epoch)
        if callable(after):
            after()                                         # front-end of the next batch, overlapping this step
        loss_acc += loss.view(1)
        if it % 20 == 0:                                   # the only host sync inside the epoch
            lv = float(loss.item())
            stop_if_poisoned(lv)
            stop_if_poisoned(float(loss_acc.item()))
            metric_logger.update(loss=lv)
        metric_logger.update(lr=float(lr_schedule[it]))
        metric_logger.update(wd=float(wd_schedule[it]))
    stop_if_poisoned(float(loss_acc.item()))                # every step of the epoch was finite: the checkpoint may be written
    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}
