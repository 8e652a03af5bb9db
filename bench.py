#!/usr/bin/env python3
"""Headline benchmark: spots/sec of the multimodal f∘g training step (BASELINE.json configs[3], "C4").

One step = one pass of the hot path over one synthetic 78x64 Visium array per GPU, exactly what one
iteration of `train_gridwise` does in the reference's multimodal tutorial (Tutorial_multimodal.ipynb
cells 23-28): DenseNet-121 image f (frozen, eval) over all 4992 spots of 128-px patches, count-MLP f over
the 2000-gene count grid, concat, 5-layer hex corrector g, foreground-masked CE, backward through g (and
the count MLP, whose parameters still require grad - the reference's GridNetHexMM quirk), gradient
all-reduce over ranks, Adam step on the corrector.  Inputs are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task description) with `roofline` for the dominant kernel
(the 1x1 or the 3x3 conv of the dense layers, whichever took more of the step; FLOP-weighted over its 58
launches per step, timed with HIP events on the
launch stream inside the timed region) and `cpu_baseline` (the CPU oracle on a bounded sample, rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W, GENES, CLASSES = 78, 64, 2000, 8
DENSENET121 = dict(growth_rate=32, block_config=(6, 12, 24, 16), num_init_features=64, bn_size=4, drop_rate=0,
                   small_inputs=False)
PEAK_F32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E
PMC_TRAFFIC_FILE = 'r01l_pmc_traffic.json'


def conv3x3_flops_per_spot(patch):
    """Algorithmic FLOPs of all dense-layer 3x3 convs per spot (2*K*N MAC-flops per output position)."""
    s = ((patch + 6 - 7) // 2 + 1 + 2 - 3) // 2 + 1
    total = 0
    for n_layers in DENSENET121['block_config']:
        total += n_layers * s * s * 2 * (9 * 128) * 32
        s //= 2
    return total


def conv3x3_executed_flops_per_spot(patch):
    """Matrix FLOPs the conv2 launches actually execute: power-of-two maps of 8 x 8 and up run Winograd F(2,3) along x -
    12 instead of 18 multiply-accumulate "taps" per output pair, i.e. 2/3 of the direct count."""
    s = ((patch + 6 - 7) // 2 + 1 + 2 - 3) // 2 + 1
    total = 0
    for n_layers in DENSENET121['block_config']:
        f = n_layers * s * s * 2 * (9 * 128) * 32
        total += f * 2 // 3 if (s >= 8 and (s & (s - 1)) == 0) else f
        s //= 2
    return total


def conv3x3_bytes_per_spot(patch):
    """Algorithmic HBM bytes of those launches per spot: the 128-channel bottleneck in, 32 new channels out."""
    s = ((patch + 6 - 7) // 2 + 1 + 2 - 3) // 2 + 1
    total = 0
    for n_layers in DENSENET121['block_config']:
        total += n_layers * s * s * 4 * (128 + 32)
        s //= 2
    return total


def _conv1x1_layers(patch):
    s = ((patch + 6 - 7) // 2 + 1 + 2 - 3) // 2 + 1
    c = DENSENET121['num_init_features']
    for n_layers in DENSENET121['block_config']:
        for l in range(n_layers):
            yield s, c + l * DENSENET121['growth_rate']
        c = (c + n_layers * DENSENET121['growth_rate']) // 2
        s //= 2


def conv1x1_flops_per_spot(patch):
    """Algorithmic FLOPs of the dense layers' 1x1 bottleneck convs per spot (K = channels so far, N = 128)."""
    return sum(s * s * 2 * k * 128 for s, k in _conv1x1_layers(patch))


def conv1x1_bytes_per_spot(patch):
    """Algorithmic HBM bytes of those launches per spot: K channels of the block buffer in, 128 out."""
    return sum(s * s * 4 * (k + 128) for s, k in _conv1x1_layers(patch))


def build_model(device, patch=128):
    import gridnext_amd as ga
    from gridnext_amd.synthetic import count_mlp
    torch.manual_seed(0)
    f_img = ga.DenseNet(num_classes=CLASSES, **DENSENET121)
    f_cnt = count_mlp(GENES, CLASSES)
    return ga.GridNetHexMM(f_img, f_cnt, (3, patch, patch), (GENES,), (H, W), CLASSES).to(device)


def cpu_baseline(patch, seed=0):
    """The CPU oracle (oracle/, kind 'port') on a bounded sample of the same step: a 26x16 = 416-spot
    sub-grid, same model family, same loss/backward/optimizer work, host cores of this box."""
    from oracle import densenet as odn, gridnet as ogn, masked_ce as oce
    from gridnext_amd.synthetic import count_mlp
    hs, ws = 26, 16
    torch.manual_seed(seed)
    f_img = odn.DenseNet(num_classes=CLASSES, **{k: v for k, v in DENSENET121.items()})
    g = ogn.GridNetHexMM(f_img, count_mlp(GENES, CLASSES), (3, patch, patch), (GENES,), (hs, ws), CLASSES)
    for p in g.patch_classifier.parameters():
        p.requires_grad = False
    opt = torch.optim.Adam(g.corrector.parameters(), lr=1e-3)
    gen = torch.Generator().manual_seed(seed)
    x_img = torch.rand((1, hs, ws, 3, patch, patch), generator=gen)
    x_cnt = torch.randint(0, 10, (1, GENES, hs, ws), generator=gen).float()
    y = torch.randint(0, CLASSES + 1, (1, hs, ws), generator=gen)
    g.train()
    g.patch_classifier.eval()
    t0 = time.time()
    out = g([x_img, x_cnt])
    loss, _, _ = oce.masked_ce(out, y, 1)
    loss.backward()
    opt.step()
    opt.zero_grad()
    dt = time.time() - t0
    n = hs * ws
    return {"value": n / dt, "unit": "spots/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "1 step on a %dx%d sub-grid (%d spots of %d px, 2000 genes), %.1f s" % (hs, ws, n, patch, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--patch', type=int, default=128)
    ap.add_argument('--arrays', type=int, default=2, help='distinct synthetic arrays resident per GPU')
    ap.add_argument('--train-f', action='store_true',
                    help='second series (SURVEY 8d): both classifiers trained through f_opt, DenseNet backward included')
    ap.add_argument('--mfma', default='f32', choices=['f32', 'f16'],
                    help="matrix-core operand type of the DenseNet convs; 'f16' = BASELINE config 5's fp16 MFMA path "
                         "(fp32 accumulate; NOT the headline, reported as dtype f16)")
    ap.add_argument('--backend', default=None, help='torch.distributed backend (default nccl = RCCL)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    args = ap.parse_args()

    from gridnext_amd import distributed as gdist
    from gridnext_amd import training as gtrain
    from gridnext_amd.synthetic import visium_array
    rank, world, device = gdist.init_from_env(args.backend)
    assert torch.cuda.is_available(), "bench.py measures the HIP path; no HIP device visible"
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    model = build_model(device, args.patch)
    gdist.broadcast_module(model)
    optimizer = torch.optim.Adam(model.corrector.parameters(), lr=1e-3)
    f_opt = None
    if args.train_f:
        f_opt = torch.optim.Adam(list(model.image_classifier.parameters()) +
                                 list(model.count_classifier.parameters()), lr=1e-4)
    else:
        for p in model.patch_classifier.parameters():           # Tutorial_multimodal.ipynb cell 27
            p.requires_grad = False
    criterion = nn.CrossEntropyLoss()
    stepped = gdist.optimizer_params(optimizer, f_opt)

    # synthetic arrays, resident in HBM (different per rank: weak scaling, 1 array per GPU per step)
    arrays = []
    for a in range(args.arrays):
        x_img, x_cnt, y = visium_array(1000 * rank + a, GENES, CLASSES, args.patch, device=device)
        arrays.append(([x_img.unsqueeze(0), x_cnt.unsqueeze(0)], y.unsqueeze(0)))

    f_img = model.image_classifier
    f_img.mfma = args.mfma
    if args.mfma == 'f16':
        # A freshly initialised DenseNet-121 with untouched running statistics (mean 0, var 1) does not normalise anything:
        # its activations grow to ~1e6 by the last block - fine in fp32, overflow in fp16.  Config 5 therefore runs with
        # running statistics calibrated on one batch of the synthetic patches (one train-mode forward, momentum 1), as any
        # network that has seen data has them; weights stay the random initialisation.
        bns = [m for m in f_img.modules() if isinstance(m, nn.BatchNorm2d)]
        moms = [m.momentum for m in bns]
        for m in bns:
            m.momentum = 1.0
        f_img.train()
        with torch.no_grad():
            f_img(arrays[0][0][0].reshape(-1, 3, args.patch, args.patch)[:64])
        for m, mo in zip(bns, moms):
            m.momentum = mo
        f_img.eval()
    model.train()
    model.patch_classifier.eval()

    def step(i):
        inputs, labels = arrays[i % len(arrays)]
        loss, correct, n_fg = gtrain._grid_loss(model, inputs, labels, criterion, 1, True)
        loss.backward()
        gdist.allreduce_gradients(stepped)
        optimizer.step()
        optimizer.zero_grad()
        if f_opt is not None:
            f_opt.step()
            f_opt.zero_grad()
        return loss

    for i in range(args.warmup):
        step(i)
    if not args.no_kernel_timing:
        f_img._probe = []                                       # (kind, start_event, end_event) per timed launch
    if gdist.is_active():
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = step(args.warmup + i)
    torch.cuda.synchronize()
    if gdist.is_active():
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if gdist.is_active():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    spots = H * W * world * args.steps
    result = {
        "metric": "spots/sec training throughput (multimodal f+g)",
        "value": spots / elapsed, "unit": "spots/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.mfma, "data": "synthetic",
        "config": {"workload": "C4: multimodal f(DenseNet-121 @%dpx + count-MLP 2000 genes) + hex g on 78x64 Visium "
                               "grids, 1 array (4992 spots) per GPU per step, %s, "
                               "g trained with Adam, masked CE" % (args.patch, "f AND g trained (f_opt; DenseNet "
                               "forward+backward, eval-mode BN as training.py:126)" if args.train_f else
                               "f frozen/eval (tutorial mode)"),
                   "arrays_per_gpu_per_step": 1, "spots_per_array": H * W, "parallelism": "dp%d" % world,
                   "final_loss": float(last.item())},
    }
    probe = getattr(f_img, '_probe', None)
    if probe:
        # the two matrix kernels of the dense layers, each timed per launch with HIP events on the launch stream inside
        # the timed region; the roofline object describes whichever took more of the step
        kern = {}
        for kind, flops_per_spot, bytes_per_spot, name in (
                ('conv1x1', conv1x1_flops_per_spot, conv1x1_bytes_per_spot, 'conv1x1_ws_kernel'),
                ('conv3x3', conv3x3_flops_per_spot, conv3x3_bytes_per_spot, 'conv3x3_dma_kernel')):
            ms = sum(s.elapsed_time(e) for k, s, e in probe if k == kind)
            n_launch = sum(1 for k, _, _ in probe if k == kind)
            flops = flops_per_spot(args.patch) * H * W * args.steps
            achieved = flops / (ms * 1e-3) / 1e12
            kern[kind] = {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": PEAK_F32_MATRIX_TFLOPS,
                          "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MATRIX_TFLOPS, "traffic": None,
                          "launches": n_launch, "avg_launch_ms": ms / max(n_launch, 1),
                          "flops_per_launch_avg": flops / max(n_launch, 1),
                          "algorithmic_bytes_per_launch_avg": bytes_per_spot(args.patch) * H * W * args.steps
                          / max(n_launch, 1), "ms_per_step": ms / args.steps}
        # HBM bytes per launch from rocprofv3 PMC passes of this same command (FETCH_SIZE and WRITE_SIZE in separate
        # runs, KiB units, FETCH doubled for 16-B/lane loads as MI355X_MICROARCH.md prescribes): tools/pmc_traffic.py,
        # profiles/README.md
        tfile = os.path.join(ROOT, 'profiles', PMC_TRAFFIC_FILE)
        if os.path.exists(tfile) and args.patch == 128 and args.mfma == 'f32':
            with open(tfile) as fh:
                tr = json.load(fh)
            for kind in kern:
                if kind in tr:
                    kern[kind]["traffic"] = tr[kind]["hbm_bytes_per_launch"]
                    kern[kind]["traffic_source"] = "profiles/%s (PMC, separate passes)" % PMC_TRAFFIC_FILE
        if getattr(f_img, 'winograd', False) and args.mfma == 'f32':
            # `achieved` above counts direct-convolution FLOPs (the algorithmic figure of SURVEY 8d); the Winograd launches
            # execute fewer: report the executed rate next to it
            k3 = kern['conv3x3']
            ex = conv3x3_executed_flops_per_spot(args.patch) * H * W * args.steps
            k3["kernel"] = "conv3x3_wino_kernel (S >= 8) + conv3x3_dma_kernel (S = 4)"
            k3["algorithm"] = ("Winograd F(2,3) along x for maps of 8 x 8 and up: 2/3 of the direct multiply-adds; `achieved` and "
                               "`frac` count direct-convolution FLOPs, `executed_*` the matrix FLOPs actually issued")
            k3["executed_achieved"] = ex / (k3["ms_per_step"] * args.steps * 1e-3) / 1e12
            k3["executed_frac"] = k3["executed_achieved"] / PEAK_F32_MATRIX_TFLOPS
        if args.mfma == 'f16':
            # config 5's kernels multiply 16x faster than they can be fed: they are priced against HBM (algorithmic bytes
            # per launch / launch time); the fp32-FLOP figure stays in `matrix_tflops` for reference
            h16 = bool(getattr(f_img, '_used_f16_buffers', False))     # block buffers in fp16: every operand is 2 B
            for kind, name in (('conv1x1', 'conv1x1_h16_kernel' if h16 else 'conv1x1_f16_kernel'),
                               ('conv3x3', 'conv3x3_dma_kernel<H16, O16>' if h16 else 'conv3x3_dma_kernel<H16> / conv3x3_f16_kernel')):
                kk = kern[kind]
                if h16:
                    kk["algorithmic_bytes_per_launch_avg"] *= 0.5
                kk["fp16_block_buffers"] = h16
                gbs = kk["algorithmic_bytes_per_launch_avg"] / (kk["avg_launch_ms"] * 1e-3) / 1e9
                kk.update({"bound": "hbm", "kernel": name, "matrix_tflops": kk["achieved"], "achieved": gbs,
                           "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS})
        dom = max(kern, key=lambda k: kern[k]["ms_per_step"])
        other = [k for k in kern if k != dom][0]
        result["roofline"] = dict(kern[dom])
        result["roofline"]["second_kernel"] = kern[other]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.patch)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if gdist.is_active():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
