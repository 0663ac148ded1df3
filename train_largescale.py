#!/usr/bin/env python3
"""Entry point with the reference's name: Part-fViT + CosFace fine-tuning on MI355X (lafs_cvpr2024_amd/train_largescale.py).
Flags of the reference that only concern its data pipeline / evaluation / timm scheduler zoo are accepted and ignored."""
import argparse

from lafs_cvpr2024_amd.train_largescale import get_args_parser, main

if __name__ == "__main__":
    args, ignored = argparse.ArgumentParser("train_largescale", parents=[get_args_parser()]).parse_known_args()
    if ignored:
        print("ignored reference flags (outside the hot path):", " ".join(ignored))
    main(args)
