"""Train/eval entry point of DeepFM; flags and defaults of the reference's DeepFM/main.py (argv is honoured)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import recommendation_amd as ra  # noqa: E402
from recommendation_amd import train_loop as tl  # noqa: E402

ra.enable_zoo_imports()
from DeepFM import DeepFM  # noqa: E402


def make_parser():
    p = tl.common_flags(argparse.ArgumentParser(), 'DeepFM')
    p.add_argument('--deep_layers', type=tl.int_list, default='[256, 128, 64, 32]')
    p.add_argument('--dropout_fm', type=float, default=0.5)
    p.add_argument('--dropout_deep', type=float, default=0.2)
    return p


if __name__ == '__main__':
    args = make_parser().parse_args()
    if isinstance(args.deep_layers, str):
        args.deep_layers = tl.int_list(args.deep_layers)
    tl.fit(DeepFM.DeepFM, args)
