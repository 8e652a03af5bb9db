#!/usr/bin/env python3
"""The `headline_split_128px` series of bench.py alone (the headline's step with DenseNet.split_conv1 / split_conv2).
python tools/bench_split.py"""
import json
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # noqa: E402

if __name__ == '__main__':
    import torch
    args = types.SimpleNamespace(no_cpu_baseline=True)
    out = bench.split_series(args, torch.device('cuda:0'), 0, 1)
    print(json.dumps(out))
