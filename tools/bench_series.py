#!/usr/bin/env python3
"""One series of bench.py on its own:  python tools/bench_series.py {patch224|c5|c5trained} [--steps N] [--warmup W]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('which', choices=['patch224', 'c5', 'c5trained'])
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    a = ap.parse_args()
    from gridnext_amd import distributed as gdist
    rank, world, device = gdist.init_from_env(None)
    ns = argparse.Namespace(no_cpu_baseline=True)
    fn = {'patch224': bench.patch224_series, 'c5': bench.config5_series, 'c5trained': bench.config5_trained_series}[a.which]
    print(json.dumps(fn(ns, device, rank, world, steps=a.steps, warmup=a.warmup)))


if __name__ == '__main__':
    main()
