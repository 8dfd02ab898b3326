"""Train/eval entry point of AFM; flags and defaults of the reference's AFM/main.py (argv is honoured)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import recommendation_amd as ra  # noqa: E402
from recommendation_amd import train_loop as tl  # noqa: E402

ra.enable_zoo_imports()
from AFM import AFM  # noqa: E402


def make_parser():
    p = tl.common_flags(argparse.ArgumentParser(), 'AFM')
    p.add_argument('--attention_size', type=int, default=10)
    p.add_argument('--dropout', type=float, default=0.2)
    return p


if __name__ == '__main__':
    args = make_parser().parse_args()
    tl.fit(AFM.AFM, args)
