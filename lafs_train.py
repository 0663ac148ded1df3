#!/usr/bin/env python3
"""Entry point with the reference's name: LAFS pre-training on MI355X (see lafs_cvpr2024_amd/lafs_train.py)."""
import argparse
from pathlib import Path

from lafs_cvpr2024_amd.lafs_train import get_args_parser, train_lafs

if __name__ == '__main__':
    parser = argparse.ArgumentParser('LAFS', parents=[get_args_parser()])
    args = parser.parse_args()
    Path(args.output_dir).mkdir(parents=True, exist_ok=True)
    train_lafs(args)
