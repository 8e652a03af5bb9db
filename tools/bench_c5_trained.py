#!/usr/bin/env python3
"""bench.py's `config5_everything_trained_256px` series on its own (kernel work on the fp16 gradient path, profiling):
    python tools/bench_c5_trained.py [--steps 3] [--warmup 1] [--no-fp32]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--no-fp32', action='store_true', help='skip the fp32-path comparison run')
    ap.add_argument('--layers', default='', help='write the per-launch probe records (kind, ms, flops, bytes) to this JSON file')
    ap.add_argument('--patch', type=int, default=256)
    a = ap.parse_args()
    import torch
    from gridnext_amd import distributed as gdist
    rank, world, device = gdist.init_from_env(None)
    out = bench.config5_trained_series(argparse.Namespace(), device, rank, world, steps=a.steps, warmup=a.warmup,
                                       P=a.patch, probe_dump=a.layers or None, compare_fp32=not a.no_fp32)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
