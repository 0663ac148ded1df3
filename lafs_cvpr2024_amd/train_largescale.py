"""Supervised Part-fViT + CosFace / ArcFace / PartialFC fine-tuning loop (reference train_largescale.py:317-963) on the HIP
fine-tune engine.  Model configuration as the reference builds it (:432, 542-557): with_land=True (trainable MobileNetV3
landmark branch), dropout = emb_dropout = 0.1, DropPath 0.1, CosFace(s=64, m=0.4) over all classes on every rank.
`--head PartialFC` / `--head ArcFace` select the class-sharded head of config C5 (parity unpinned, see partial_fc.py).

Kept from the reference: flags that define the step (batch size, epochs, the lr rescale of :472
`acc_step/480 * lr * sqrt(world*bs/336) * 336`, weight decay 0.1 on >= 2-D tensors (:618-627; `--weight-decay` is parsed by the
reference but never reaches its optimizer), mixup alpha/prob, acc_step=3 from supervised_config.py:37, warm-up(5 epochs)+cosine(eta_min 1e-6) LR, loading
`ckpt['teacher']` of an SSL checkpoint with the 'encoder.|backbone.|module.' prefixes stripped and strict=False).
Out of scope here: MXNet recordio datasets, the torchvision tensor transforms of FaceDataset
(RandomResizedCrop / ColorJitter / RandomErasing on uint8 tensors: torchvision is absent, unpinnable), LFW/CFP/AgeDB evaluation,
tensorboard; `--data synthetic` feeds uint8 batches of the right shape.
"""
import argparse
import math
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import utils
from .face_pre_pro.ViT_face import ViT_face_landmark_patch8
from .finetune_engine import FinetuneEngine


def get_config(args):
    """The few supervised_config.py entries that define the step."""
    return dict(acc_step=3, SEED=1337, INPUT_SIZE=[112, 112], EMBEDDING_SIZE=768, WARMUP_EPOCH=5)


def get_args_parser():
    p = argparse.ArgumentParser("Part-fViT fine-tuning", add_help=False)
    p.add_argument("--batch_size", "-b", default=128, type=int)
    p.add_argument("--epochs", "-e", default=34, type=int)
    p.add_argument("--lr", default=1e-3, type=float, help="base rate before the reference's rescale (train_largescale.py:355,472)")
    p.add_argument("--weight_decay", default=0.1, type=float)
    p.add_argument("--head", default="CosFace", type=str, choices=["CosFace", "PartialFC", "ArcFace"],
                   help="CosFace: the reference's dense head; ArcFace: dense head with the additive angular margin (m=0.5); "
                        "PartialFC: class centres sharded over the ranks (config C5)")
    p.add_argument("--partial_margin", default="CosFace", type=str, choices=["CosFace", "ArcFace"])
    p.add_argument("--sample_rate", default=1.0, type=float, help="PartialFC negative-class sampling rate")
    p.add_argument("--with_land", default=True, type=utils.bool_flag, help="trainable landmark branch (reference :432)")
    p.add_argument("--dropout", default=0.1, type=float, help="dropout = emb_dropout of the reference (:552-555)")
    p.add_argument("--landmark_ckpt", default="", type=str, help="stn./output_layer. weights (reference :659-661)")
    p.add_argument("--num_class", default=205990, type=int)
    p.add_argument("--mixup", default=0.2, type=float)
    p.add_argument("--mixup-prob", dest="mixup_prob", default=0.1, type=float)
    p.add_argument("--drop_path", default=0.1, type=float)
    p.add_argument("--model_dir", default="", type=str, help="LAFS checkpoint whose ['teacher'] weights initialise the backbone")
    p.add_argument("--pretrain_path", default="", type=str, help="stage-1 checkpoint with the landmark CNN (alias of --landmark_ckpt)")
    p.add_argument("--data", default="synthetic", type=str)
    p.add_argument("--rand_au", default=False, type=utils.bool_flag,
                   help="RandAugment of the reference's FaceDataset (rand_au=True, train_largescale.py:506) on the device")
    p.add_argument("--rand_au_config", default="rand-m1-mstd0.5-inc1", type=str, help="config_str of train_largescale.py:506")
    p.add_argument("--rand_mirror", default=False, type=utils.bool_flag, help="FaceDataset's random horizontal flip (image_iter.py:308-311)")
    p.add_argument("--steps_per_epoch", default=100, type=int)
    p.add_argument("--outdir", "-o", default=".", type=str)
    p.add_argument("--dist_url", default="env://", type=str)
    p.add_argument("--local_rank", default=0, type=int)
    return p


def warmup_cosine(base_lr, epoch_float, warmup_epochs, total_epochs, eta_min=1e-6):
    """GradualWarmupScheduler(multiplier=1) + CosineAnnealingLR (train_largescale.py:728-733); the `warmup_scheduler` package is
    not vendored by the reference, so this follows its documented behaviour (parity unpinned)."""
    if epoch_float < warmup_epochs:
        return base_lr * epoch_float / warmup_epochs
    t, T = epoch_float - warmup_epochs, max(total_epochs - warmup_epochs, 1)
    return eta_min + 0.5 * (base_lr - eta_min) * (1 + math.cos(math.pi * t / T))


def load_ssl_teacher(backbone, path, min_matched=0.9):
    """Initialise from ckpt['teacher'] with the 'encoder.' / 'backbone.' / 'module.' prefixes removed, strict=False
    (train_largescale.py:639-657).  Unlike the reference this refuses to continue silently from random weights: a tensor whose
    shape does not fit (e.g. a `loss.weight` of another class count) is dropped with a message, and fewer than `min_matched` of
    the backbone's trunk tensors being initialised is an error -- that is what a checkpoint of the wrong architecture (a DINO
    ViT teacher: keys blocks.N.attn.qkv..., pos_embed) looks like under strict=False."""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    sd = ck.get("teacher", ck)
    own = backbone.state_dict()
    clean, dropped = {}, []
    for k, v in sd.items():
        if 'dummy_orthogonal_classifier' not in k:
            k = k.replace('encoder.', '').replace('backbone.', '').replace('module.', '')
        if k in own and tuple(own[k].shape) != tuple(v.shape):
            dropped.append((k, tuple(v.shape), tuple(own[k].shape)))
            continue
        clean[k] = v
    trunk = [k for k in own if not k.startswith(("loss.", "stn.", "output_layer."))]
    hit = [k for k in trunk if k in clean]
    for k, a, b in dropped:
        print(f"=> SSL teacher: skipping {k}: checkpoint {a} vs model {b}")
    if len(hit) < min_matched * len(trunk):
        raise RuntimeError(f"{path}: only {len(hit)} of the backbone's {len(trunk)} trunk tensors are in ckpt['teacher'] "
                           f"(first missing: {[k for k in trunk if k not in clean][:3]}); is this a checkpoint of another architecture? "
                           "LAFS pre-training must use --arch mynet for its teacher to initialise this model")
    print(f"=> loaded SSL teacher: {len(hit)}/{len(trunk)} trunk tensors;", backbone.load_state_dict(clean, strict=False))


def load_landmark_branch(backbone, path):
    """load_part_checkpoint_landmark(pretrain_name=['stn', 'output']) (train_largescale.py:659-661): copy the tensors whose
    key starts with stn. / output_layer. from a stage-1 checkpoint."""
    sd = torch.load(path, map_location="cpu", weights_only=False)
    sd = sd.get("model", sd)
    part = {}
    for k, v in sd.items():
        k = k[len("module."):] if k.startswith("module.") else k
        if k.startswith(("stn.", "output_layer.")):
            part[k] = v
    print("=> landmark branch:", len(part), "tensors;", backbone.load_state_dict(part, strict=False))


def main(args):
    utils.init_distributed_mode(args)
    cfg = get_config(args)
    utils.fix_random_seeds(cfg["SEED"])
    device = torch.device("cuda", args.gpu)
    world = utils.get_world_size()
    sharded = args.head == "PartialFC"
    arc = (args.partial_margin if sharded else args.head) == "ArcFace"
    # the dense ArcFace head lives in the same `loss.weight` tensor as CosFace (the reference names an ArcFace class it never
    # defines, ViT_face.py:654-655); the margin is applied by the fused kernel
    backbone = ViT_face_landmark_patch8(loss_type="None" if sharded else "CosFace", GPU_ID=None, num_class=args.num_class,
                                        image_size=112, patch_size=8, dim=768, depth=12, heads=11, mlp_dim=2048,
                                        dropout=args.dropout, emb_dropout=args.dropout, with_land=args.with_land,
                                        drop_path_rate=args.drop_path)
    if args.model_dir:
        load_ssl_teacher(backbone, args.model_dir)
    if args.landmark_ckpt or args.pretrain_path:
        load_landmark_branch(backbone, args.landmark_ckpt or args.pretrain_path)
    head = None
    if sharded:
        from .partial_fc import PartialFC
        head = PartialFC(768, args.num_class, args.batch_size, sample_rate=args.sample_rate, s=64.0, m=0.5 if arc else 0.4,
                         margin_type=1 if arc else 0, device=device, seed=cfg["SEED"])
    engine = FinetuneEngine(backbone, args.batch_size, acc_step=cfg["acc_step"], mixup_alpha=args.mixup, mixup_prob=args.mixup_prob,
                            s=64.0, m=0.5 if arc else 0.4, margin_type=1 if arc else 0, device=device, sharded_head=head)
    # train_largescale.py:472:  lr = acc_step / 480 * lr * sqrt(world * BATCH_SIZE / 336) * 336
    base_lr = cfg["acc_step"] / 480.0 * args.lr * math.sqrt(world * args.batch_size / 336.0) * 336
    n_it = args.steps_per_epoch
    gen = torch.Generator(device=device).manual_seed(cfg["SEED"] + utils.get_rank())
    rand_au = None
    if args.rand_au:                    # the loader's per-sample PIL RandAugment as ONE launch per batch (randaug.py / csrc/randaug.hip)
        from .randaug import DeviceRandAugment
        rand_au = DeviceRandAugment(args.rand_au_config, {"translate_const": 117}, seed=cfg["SEED"] + utils.get_rank())
    t0 = time.time()
    for epoch in range(args.epochs):
        for it in range(n_it):
            x = torch.randint(0, 256, (args.batch_size, 3, 112, 112), device=device, dtype=torch.uint8, generator=gen)
            y = torch.randint(0, args.num_class, (args.batch_size,), device=device, generator=gen)
            if args.rand_mirror:        # _rd = random.randint(0, 1) per sample, flip along the width
                flip = torch.randint(0, 2, (args.batch_size, 1, 1, 1), device=device, generator=gen).bool()
                x = torch.where(flip, x.flip(3), x)
            if rand_au is not None:
                x = rand_au(x)
            lr = warmup_cosine(base_lr, epoch + it / n_it, cfg["WARMUP_EPOCH"], args.epochs)
            loss = engine.step(x, y, lr=lr, weight_decay=args.weight_decay)
            if it % 50 == 0:
                print(f"Epoch {epoch} it {it}/{n_it} loss {float(loss.item()):.4f} lr {lr:.3e} "
                      f"{(epoch * n_it + it + 1) * args.batch_size * world / (time.time() - t0):.1f} samples/s")
        if utils.is_main_process():
            torch.save({"module." + k: v for k, v in backbone.state_dict().items()},
                       os.path.join(args.outdir, f"Backbone_VIT_Epoch_{epoch + 1}.pth"))    # IJB loader expects 'module.' (IJB_evaluation.py:126)
    if dist.is_initialized():
        dist.destroy_process_group()
