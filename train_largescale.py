#!/usr/bin/env python3
"""Entry point with the reference's name: Part-fViT + CosFace fine-tuning on MI355X (lafs_cvpr2024_amd/train_largescale.py)."""
import argparse

from lafs_cvpr2024_amd.train_largescale import get_args_parser, main

if __name__ == "__main__":
    main(argparse.ArgumentParser("train_largescale", parents=[get_args_parser()]).parse_args())
