#!/usr/bin/env python3
"""Headline benchmark: spots/sec of the multimodal f∘g training step (BASELINE.json configs[3], "C4").

One step = one pass of the hot path over one synthetic 78x64 Visium array per GPU, exactly what one
iteration of `train_gridwise` does in the reference's multimodal tutorial (Tutorial_multimodal.ipynb
cells 23-28): DenseNet-121 image f (frozen, eval) over all 4992 spots of 128-px patches, count-MLP f over
the 2000-gene count grid, concat, 5-layer hex corrector g, foreground-masked CE, backward through g (and
the count MLP, whose parameters still require grad - the reference's GridNetHexMM quirk), gradient
all-reduce over ranks, Adam step on the corrector.  Inputs are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W          # self-launching: N > 1 spawns one worker process per GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             # torchrun: every process is a worker (RANK/WORLD_SIZE set)

The launching parent never touches the GPU (it does not even import torch): it starts N fresh children with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, waits, and exits non-zero if any of them failed.

Rank 0 prints ONE JSON line (contract in the task description) with
  * `roofline`      the dominant kernel (the 1x1 or the 3x3 conv of the dense layers, whichever took more of the step;
                    FLOP-weighted over its 58 launches per step, timed with HIP events on the launch stream inside the
                    timed region);
  * `cpu_baseline`  the CPU oracle on a bounded sample (rank 0, N = 1 only): 1 warm-up + 2 timed steps on a 26x16
                    sub-grid, physical cores, CPU model string, the SAME weights as the GPU leg;
  * `ce_vs_ref`     BASELINE.json's "CE vs ref": the masked CE of that sub-grid through the HIP path and through the
                    oracle, same weights and inputs (|dCE| <= 1e-4 asserted for f32), argmax agreement on decided spots;
  * `series_summary` / `series_file`   further series on the same box (value and ms/step in the line, the full objects
                    with their rooflines in the file): `from_host` = the same step fed from pageable host memory (uint8
                    patches, DataLoader, pinned double-buffered H2D prefetcher; PCIe inside the timed region - never the
                    headline `value`); `config5_f16_256px` = BASELINE config 5's step (256-px patches, fp16 MFMA conv
                    path) on this GPU, `headline_geometry_f16_128px` = the same path on the headline's 128-px patches
                    (dtype f16: its CE is reported, not gated at 1e-4); `train_f` = the second series of SURVEY 8d, both classifiers trained through f_opt
                    (DenseNet forward + backward), with its own roofline object; `other_configs` = BASELINE configs 1-3
                    through the product's training loops (N = 1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W, GENES, CLASSES = 78, 64, 2000, 8
DENSENET121 = dict(growth_rate=32, block_config=(6, 12, 24, 16), num_init_features=64, bn_size=4, drop_rate=0,
                   small_inputs=False)
PEAK_F32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E
PEAK_F16_TFLOPS = 2500.0            # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA peak
SUB_H, SUB_W = 26, 16               # the sub-grid of the CPU leg (416 spots)


# ------------------------------------------------------------------------------------------ algorithmic work per spot
def _first_map(patch):
    return ((patch + 6 - 7) // 2 + 1 + 2 - 3) // 2 + 1


def conv3x3_flops_per_spot(patch):
    """Algorithmic FLOPs of all dense-layer 3x3 convs per spot (2*K*N MAC-flops per output position)."""
    s = _first_map(patch)
    total = 0
    for n_layers in DENSENET121['block_config']:
        total += n_layers * s * s * 2 * (9 * 128) * 32
        s //= 2
    return total


def conv3x3_executed_flops_per_spot(patch):
    """Matrix FLOPs the conv2 launches actually execute: power-of-two maps of 8 x 8 and up run Winograd F(2,3) along x -
    12 instead of 18 multiply-accumulate "taps" per output pair, i.e. 2/3 of the direct count."""
    s = _first_map(patch)
    total = 0
    for n_layers in DENSENET121['block_config']:
        f = n_layers * s * s * 2 * (9 * 128) * 32
        total += f * 2 // 3 if (s >= 8 and (s & (s - 1)) == 0) else f
        s //= 2
    return total


def conv3x3_bytes_per_spot(patch):
    """Algorithmic HBM bytes of those launches per spot: the 128-channel bottleneck in, 32 new channels out."""
    s = _first_map(patch)
    total = 0
    for n_layers in DENSENET121['block_config']:
        total += n_layers * s * s * 4 * (128 + 32)
        s //= 2
    return total


def _conv1x1_layers(patch):
    s = _first_map(patch)
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


def dgrad1x1_bytes_per_spot(patch):
    """conv1 data gradient fused with norm1/relu1's adjoint: the 128-wide bottleneck gradient in, then three passes over
    [M][cin] (layer input for mask and x_hat, block gradient read, block gradient written back)."""
    return sum(s * s * 4 * (128 + 3 * k) for s, k in _conv1x1_layers(patch))


def dense_layer_bytes_per_spot(patch):
    """The fused dense layer (gnx_dense_layer_f16): K channels of the block buffer in, the layer's 32 new channels out - the
    128-channel bottleneck stays in LDS.  In 4-byte units like the other tables (the fp16 series halves them): 26.6 MB per
    spot at 256 px in fp16."""
    return sum(s * s * 4 * (k + 32) for s, k in _conv1x1_layers(patch))


# kind -> (kernel name, FLOPs per spot, algorithmic bytes per spot)           (backward kinds: the f-trained series)
KINDS = {
    'dense_layer': ('dense_layer_f16_kernel / dense_layer_f16_s64_kernel', lambda patch: conv1x1_flops_per_spot(patch) + conv3x3_flops_per_spot(patch),
                    dense_layer_bytes_per_spot),
    'conv1x1': ('conv1x1_ws_kernel', conv1x1_flops_per_spot, conv1x1_bytes_per_spot),
    'conv3x3': ('conv3x3_dma_kernel', conv3x3_flops_per_spot, conv3x3_bytes_per_spot),
    'wgrad3x3': ('wgrad 3x3 (gnx_wgrad_bnrelu, taps = 9: slab kernel + fixed-order reduce)', conv3x3_flops_per_spot,
                 conv3x3_bytes_per_spot),
    'wgrad1x1': ('wgrad 1x1 (gnx_wgrad_bnrelu, taps = 1: slab kernel + fixed-order reduce)', conv1x1_flops_per_spot,
                 conv1x1_bytes_per_spot),
    'dgrad3x3': ('conv3x3_dma_kernel, data-gradient shape (K = 32, N = 128)', conv3x3_flops_per_spot,
                 conv3x3_bytes_per_spot),
    'dgrad3x3_bn2': ('conv3x3_dma_kernel<..., data-gradient shape + norm2/relu2 adjoint>', conv3x3_flops_per_spot,
                     lambda patch: conv3x3_bytes_per_spot(patch) + sum(
                         n_l * (_first_map(patch) >> b) ** 2 * 4 * 128 for b, n_l in enumerate(DENSENET121['block_config']))),
    'dgrad1x1_bn1': ('conv1x1_ws_kernel<..., dgrad + norm1/relu1 adjoint>', conv1x1_flops_per_spot,
                     dgrad1x1_bytes_per_spot),
    'dgrad_wgrad1x1_bn1': ('dgrad_wgrad1x1_f32_kernel (conv1: data gradient + norm1/relu1 adjoint + weight gradient, one pass)',
                           lambda patch: 2 * conv1x1_flops_per_spot(patch), dgrad1x1_bytes_per_spot),
}


def kernel_table(probe, patch, steps):
    """{kind: roofline dict} from the (kind, start_event, end_event, flops, bytes) records of `steps` timed steps.  Every
    kind is credited with the ALGORITHMIC work of exactly the launches that were timed under its name (the launch sites in
    gridnext_amd/densenet.py / densenet_train.py state each launch's FLOPs and bytes from its own M, K, N): a series that
    launches a kernel twice per step (recompute) or splits a role over two kinds is priced launch by launch.  The closed
    forms above (`*_per_spot`) are the same sums over one pass of the network and serve DESIGN.md's tables and the tests."""
    kern = {}
    for kind in sorted({r[0] for r in probe}):
        if kind not in KINDS:
            continue
        name, flops_per_spot, bytes_per_spot = KINDS[kind]
        recs = [r for r in probe if r[0] == kind]
        ms = sum(r[1].elapsed_time(r[2]) for r in recs)
        n_launch = len(recs)
        if all(len(r) >= 5 and r[3] is not None for r in recs):
            flops, nbytes = sum(r[3] for r in recs), sum(r[4] for r in recs)
        else:                                                   # records without their own work: one pass of the network
            flops, nbytes = flops_per_spot(patch) * H * W * steps, bytes_per_spot(patch) * H * W * steps
        achieved = flops / (ms * 1e-3) / 1e12
        kern[kind] = {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": PEAK_F32_MATRIX_TFLOPS,
                      "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MATRIX_TFLOPS, "traffic": None,
                      "launches": n_launch, "avg_launch_ms": ms / max(n_launch, 1),
                      "flops_per_launch_avg": flops / max(n_launch, 1),
                      "algorithmic_bytes_per_launch_avg": nbytes / max(n_launch, 1),
                      "algorithmic_gbs": nbytes / (ms * 1e-3) / 1e9,
                      "ms_per_step": ms / steps}
    return kern


def pmc_traffic_table(suffix=''):
    """{kind: {hbm_bytes_per_launch, ...}} from the newest profiles/r*_pmc_traffic<suffix>.json, with its file name; HBM
    bytes per launch from rocprofv3 PMC passes of the same command (FETCH_SIZE and WRITE_SIZE in separate runs, KiB units,
    FETCH doubled for 16-B/lane loads as MI355X_MICROARCH.md prescribes): tools/pmc_traffic.py, profiles/README.md."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic%s.json' % suffix)), reverse=True)
    for f in files:
        try:
            with open(f) as fh:
                return json.load(fh), os.path.basename(f)
        except (OSError, ValueError):
            continue
    return {}, None


# which sources a kernel kind is built from: its PMC figure is only attached while they still hash to what the passes saw
KIND_SOURCES = {
    'dense_layer': ('dense_layer_f16.hip', 'fwd_common.h', 'common.h'),
    'dense_layer_tape': ('dense_layer_f16.hip', 'fwd_common.h', 'common.h'),
    'conv1x1': ('conv1x1.hip', 'densenet_f16.hip', 'fwd_common.h', 'common.h'),
    'transition': ('conv1x1.hip', 'densenet_f16.hip', 'transition_f16.hip', 'fwd_common.h', 'common.h'),
    'conv3x3': ('conv3x3.hip', 'fwd_common.h', 'common.h'),
    'stem': ('stem_pool.hip', 'fwd_common.h', 'common.h'),
    'dgrad1x1_bn1': ('conv1x1.hip', 'fwd_common.h', 'common.h'),
    'dgrad_wgrad1x1_bn1': ('dgrad_wgrad_f32.hip', 'common.h'),
    'dgrad3x3_bn2': ('conv3x3.hip', 'fwd_common.h', 'common.h'),
    'wgrad1x1': ('densenet_bwd.hip', 'common.h'),
    'wgrad3x3': ('densenet_bwd.hip', 'common.h'),
    'dgrad_wgrad1x1_bn1_f16': ('dense_bwd_f16.hip', 'common.h'),
    'dgrad1x1_bn1_f16': ('dense_bwd_f16.hip', 'common.h'),
    'dgrad3x3_bn2_f16': ('dense_bwd_f16.hip', 'common.h'),
    'wgrad3x3_f16': ('dense_bwd_f16.hip', 'common.h'),
    'conv3x3_bwd_f16': ('dense_bwd_f16.hip', 'common.h'),
    'wgrad1x1_f16': ('dense_bwd_f16.hip', 'common.h'),
    'stem_bwd_f16': ('stem_bwd_f16.hip', 'common.h'),
    'conv1x1_split': ('conv1x1_split.hip', 'fwd_common.h', 'common.h'),
    'conv3x3_split': ('conv3x3_split.hip', 'fwd_common.h', 'common.h'),
}


def _source_hash(name):
    import hashlib
    try:
        with open(os.path.join(ROOT, 'gridnext_amd', 'csrc', name), 'rb') as fh:
            return hashlib.sha1(fh.read()).hexdigest()
    except OSError:
        return None


def attach_traffic(kern, suffix):
    """PMC traffic per launch from the newest profiles/r*_pmc_traffic<suffix>.json - but only for kernels whose sources are
    the ones those passes profiled (the file records their sha1): after a kernel change without a new pass the field stays
    null and says why."""
    table, name = pmc_traffic_table(suffix)
    recorded = table.get('_sources')
    if 'dense_layer' in table:                  # (tools/pmc_traffic.py groups by kernel name: the taped launches are that kernel)
        table.setdefault('dense_layer_tape', table['dense_layer'])
    for kind in kern:
        if kind not in table:
            continue
        if recorded is None:
            kern[kind]["traffic_source"] = "profiles/%s predates source hashing: not attached" % name
            continue
        stale = [f for f in KIND_SOURCES.get(kind, ()) if recorded.get(f) != _source_hash(f)]
        if stale:
            kern[kind]["traffic_source"] = "profiles/%s is older than %s: not attached (re-run the PMC passes)" % (name, ", ".join(stale))
            continue
        kern[kind]["traffic"] = table[kind]["hbm_bytes_per_launch"]
        kern[kind]["traffic_source"] = "profiles/%s (PMC, separate passes; sources unchanged since)" % name


def winograd_credit(k3, patch, steps):
    """The Winograd launches execute 2/3 of the direct-convolution multiply-adds: `achieved` / `frac` become the EXECUTED
    matrix FLOPs against the matrix peak; the direct-convolution credit (SURVEY 8d's algorithmic figure) moves to
    `direct_conv_*`."""
    # (whole passes of the network were timed, so the executed : direct ratio of one pass applies to the recorded total)
    ex = k3["flops_per_launch_avg"] * k3["launches"] * conv3x3_executed_flops_per_spot(patch) / conv3x3_flops_per_spot(patch)
    k3["kernel"] = "conv3x3_wino_kernel (S >= 8) + conv3x3_dma_kernel (S = 4)"
    k3["algorithm"] = ("Winograd F(2,3) along x for maps of 8 x 8 and up: 2/3 of the direct multiply-adds. "
                       "`achieved`/`frac` = executed matrix FLOPs; `direct_conv_*` = the same time credited with "
                       "direct-convolution FLOPs (can exceed the peak, not a fraction of it)")
    k3["direct_conv_tflops"] = k3["achieved"]
    k3["direct_conv_flops_per_launch_avg"] = k3["flops_per_launch_avg"]
    k3["achieved"] = ex / (k3["ms_per_step"] * steps * 1e-3) / 1e12
    k3["frac"] = k3["achieved"] / PEAK_F32_MATRIX_TFLOPS
    k3["flops_per_launch_avg"] = ex / max(k3["launches"], 1)


class _RankDiag:
    """What a multi-GPU run needs in its line to explain a bad scaling point: every rank's own step time (before the MAX that
    the headline takes), the time inside the gradient all-reduce (HIP events on the launch stream around
    `distributed.allreduce_gradients`), the RCCL version, and - for host-fed series - every rank's H2D rate."""

    def __init__(self, device):
        import torch
        from gridnext_amd import distributed as gdist
        self.torch, self.gdist, self.device = torch, gdist, device
        self.events = []

    def allreduce(self, params):
        if not self.gdist.is_active():
            return
        t = self.torch
        e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        e0.record()
        self.gdist.allreduce_gradients(params)
        e1.record()
        self.events.append((e0, e1))

    def reset(self):
        self.events = []

    def report(self, elapsed_own, steps, extra=None):
        """Call after torch.cuda.synchronize(); elapsed_own = this rank's own wall time of the timed steps (s)."""
        t, gdist = self.torch, self.gdist
        world = gdist.world_size()
        ar_ms = sum(a.elapsed_time(b) for a, b in self.events) / max(steps, 1) if self.events else 0.0
        vals = [1e3 * elapsed_own / max(steps, 1), ar_ms] + list(extra or [])
        table = [vals]
        if gdist.is_active():
            mine = t.tensor(vals, dtype=t.float64, device=self.device)
            allv = [t.zeros_like(mine) for _ in range(world)]
            t.distributed.all_gather(allv, mine)
            table = [v.tolist() for v in allv]
        per_rank = [r[0] for r in table]
        out = {"per_rank_ms_per_step": {"min": min(per_rank), "max": max(per_rank), "all": per_rank},
               "allreduce_ms_per_step": {"max": max(r[1] for r in table), "all": [r[1] for r in table],
                                         "timed_with": "HIP events around distributed.allreduce_gradients" if gdist.is_active()
                                         else "no collective (1 process)"}}
        if extra:
            out["extra_per_rank"] = [r[2:] for r in table]
        try:
            out["rccl_version"] = ".".join(str(v) for v in t.cuda.nccl.version()) if gdist.is_active() and \
                t.distributed.get_backend() == 'nccl' else None
        except Exception:                                          # noqa: BLE001
            out["rccl_version"] = None
        return out


def build_model(device, patch=128):
    import torch
    import gridnext_amd as ga
    from gridnext_amd.synthetic import count_mlp
    torch.manual_seed(0)
    f_img = ga.DenseNet(num_classes=CLASSES, **DENSENET121)
    if os.environ.get('GNX_BENCH_SPLIT') == '1':               # (diagnosis only: every series of the run on the opt-in split-operand convs)
        f_img.split_conv1 = f_img.split_conv2 = f_img.split_wgrad = True
    f_cnt = count_mlp(GENES, CLASSES)
    return ga.GridNetHexMM(f_img, f_cnt, (3, patch, patch), (GENES,), (H, W), CLASSES).to(device)


# ------------------------------------------------------------------------------------------ CPU leg + CE vs reference
def _host_cpu():
    """(physical cores this process may use, CPU model string)."""
    model, cores = 'unknown', set()
    try:
        phys = core = None
        with open('/proc/cpuinfo') as fh:
            for line in fh:
                if line.startswith('model name') and model == 'unknown':
                    model = line.split(':', 1)[1].strip()
                elif line.startswith('physical id'):
                    phys = line.split(':', 1)[1].strip()
                elif line.startswith('core id'):
                    core = line.split(':', 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    n_phys = len(cores) or (os.cpu_count() or 1)
    try:
        n_phys = min(n_phys, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    return max(1, n_phys), model


def _chunked_frozen_f(f, chunk):
    """The oracle's image classifier over a whole grid in chunks (a frozen, eval-mode f treats spots independently, so this is
    the same function as one call - it only bounds the oracle's memory and keeps its convolutions cache-sized)."""
    import torch

    class ChunkedFrozenF(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.f = f

        def forward(self, x):
            assert not self.f.training
            with torch.no_grad():
                return torch.cat([self.f(x[i:i + chunk]) for i in range(0, len(x), chunk)], 0)

    return ChunkedFrozenF()


def full_grid_ce(model, patch, mfma, device, inputs=None, chunk=64):
    """BASELINE's "CE vs ref" at the benchmark's OWN size: ONE full 78 x 64 array (4992 spots of `patch`-px patches + 2000
    genes) through GridNetHexMM -> masked CE on the HIP path and through the CPU oracle (forward + CE only), same state_dict,
    same inputs, tutorial mode - g's train-mode BatchNorm statistics and n_fg are those of the timed step, not of a sub-grid
    (/root/reference/gridnext/training.py:146-160, gridnet_models.py:226-235).  The oracle's frozen image f runs in chunks."""
    _, ce = cpu_leg(model, patch, mfma, device, timed_steps=0, sub_hw=(H, W), inputs=inputs, oracle_chunk=chunk)
    return ce


def cpu_leg(model, patch, mfma, device, timed_steps=2, sub_hw=None, u8=False, inputs=None, oracle_chunk=None):
    """The CPU oracle (oracle/, kind 'port') and the HIP path on the SAME bounded sample and the SAME weights: a 26x16
    = 416-spot sub-grid of synthetic array 0 (`sub_hw`: another sub-grid), tutorial mode.  Returns (cpu_baseline, ce_vs_ref).
    The HIP side runs first (forward + masked CE), on copies of the trainable parts so the benchmark model is untouched;
    the oracle then does 1 warm-up + `timed_steps` timed training steps (forward, CE, backward, Adam on the corrector);
    timed_steps = 0: the comparison only (cpu_baseline None).  u8: the HIP side gets uint8 patches (ToTensor inside the stem
    kernel), the oracle the same bytes / 255."""
    import copy
    import torch
    import gridnext_amd as ga
    from gridnext_amd import functional as GF
    from gridnext_amd.synthetic import count_mlp
    from oracle import densenet as odn, gridnet as ogn, masked_ce as oce

    host_cores, cpu_model = _host_cpu()
    # the oracle's torch convolutions stop scaling beyond a few dozen threads (measured on the GPU box's EPYC 9575F,
    # tools/diag/oracle_threads.py: 106 spots/s forward on 16 or 32 threads, 53 on 64, 19-22 on 128): 32 at most
    cores = min(host_cores, 32)
    torch.set_num_threads(cores)
    SUB_H, SUB_W = sub_hw or (globals()['SUB_H'], globals()['SUB_W'])
    gen = torch.Generator().manual_seed(12345)
    if inputs is not None:                                       # (x_img (1,h,w,3,P,P) float, x_cnt (1,G,h,w), y (1,h,w)) on the CPU
        x_img, x_cnt, y = inputs
        x_hip = x_img
    else:
        if u8:
            x_img8 = torch.randint(0, 256, (1, SUB_H, SUB_W, 3, patch, patch), generator=gen, dtype=torch.uint8)
            x_img = x_img8.float() / 255
        else:
            x_img = torch.rand((1, SUB_H, SUB_W, 3, patch, patch), generator=gen)
        x_cnt = torch.randint(0, 10, (1, GENES, SUB_H, SUB_W), generator=gen).float()
        y = torch.randint(0, CLASSES + 1, (1, SUB_H, SUB_W), generator=gen)
        x_img *= (y > 0).float().view(1, SUB_H, SUB_W, 1, 1, 1)
        x_cnt *= (y > 0).float().view(1, 1, SUB_H, SUB_W)
        x_hip = (x_img8 * (y > 0).to(torch.uint8).view(1, SUB_H, SUB_W, 1, 1, 1)) if u8 else x_img

    # ---- HIP path on the sub-grid: the benchmark model's own image f (frozen: shared), copies of count f and corrector
    sub = ga.GridNetHexMM(model.image_classifier, copy.deepcopy(model.count_classifier), (3, patch, patch), (GENES,),
                          (SUB_H, SUB_W), CLASSES).to(device)
    sub.corrector.load_state_dict(model.corrector.state_dict())
    sub.train()
    sub.patch_classifier.eval()
    state = {k: v.detach().cpu().clone() for k, v in sub.state_dict().items()}
    with torch.no_grad():
        logits = sub.forward_nhwc([x_hip.to(device), x_cnt.to(device)])
        loss_hip, stats, preds = GF.masked_cross_entropy(logits.reshape(-1, CLASSES), y.to(device), 1)
    torch.cuda.synchronize()
    loss_hip, preds_hip = float(loss_hip.item()), preds.cpu()
    # the same forward with the image f's conv1 / conv2 on split bf16 operands (opt-in form of the fp32 path): held to the same
    # oracle result below, at no extra CPU cost
    split_hip = None
    f_hip = model.image_classifier
    if mfma == 'f32' and hasattr(f_hip, 'split_conv1') and getattr(f_hip, 'mfma', None) == 'f32':
        keep = (f_hip.split_conv1, f_hip.split_conv2)
        f_hip.split_conv1 = f_hip.split_conv2 = True
        try:
            with torch.no_grad():
                logits2 = sub.forward_nhwc([x_hip.to(device), x_cnt.to(device)])
                loss2, _, preds2 = GF.masked_cross_entropy(logits2.reshape(-1, CLASSES), y.to(device), 1)
            torch.cuda.synchronize()
            split_hip = (float(loss2.item()), preds2.cpu())
        finally:
            f_hip.split_conv1, f_hip.split_conv2 = keep
    del sub

    # ---- oracle with the same weights
    f_img = odn.DenseNet(num_classes=CLASSES, **DENSENET121)
    f_img.load_named_state({k[len('image_classifier.'):]: v for k, v in state.items()
                            if k.startswith('image_classifier.')})
    f_cnt = count_mlp(GENES, CLASSES)
    f_cnt.load_state_dict({k[len('count_classifier.'):]: v for k, v in state.items()
                           if k.startswith('count_classifier.')})
    if oracle_chunk:
        f_img = _chunked_frozen_f(f_img, oracle_chunk)
    g = ogn.GridNetHexMM(f_img, f_cnt, (3, patch, patch), (GENES,), (SUB_H, SUB_W), CLASSES)
    g.corrector.load_state_dict({k[len('corrector.'):]: v for k, v in state.items() if k.startswith('corrector.')})
    for p in g.patch_classifier.parameters():
        p.requires_grad = False
    opt = torch.optim.Adam(g.corrector.parameters(), lr=1e-3)
    g.train()
    g.patch_classifier.eval()

    def step():
        out = g([x_img, x_cnt])
        loss, _, _ = oce.masked_ce(out, y, 1)
        loss.backward()
        opt.step()
        opt.zero_grad()
        return out.detach(), float(loss.item())

    t0 = time.time()
    out_ref, loss_ref = step()                                   # warm-up step; its forward is the CE comparison
    warm = time.time() - t0
    t0 = time.time()
    for _ in range(timed_steps):
        step()
    dt = (time.time() - t0) / max(timed_steps, 1)
    n = SUB_H * SUB_W

    rows = out_ref.permute(0, 2, 3, 1).reshape(-1, CLASSES)
    fg = y.reshape(-1) > 0
    top = rows.topk(2, dim=1).values
    decided = fg & ((top[:, 0] - top[:, 1]) > 1e-3)
    agree = int((preds_hip[decided] == rows.argmax(1)[decided]).sum())
    ce = {"hip": loss_hip, "oracle": loss_ref, "abs_diff": abs(loss_hip - loss_ref),
          "argmax_agree": agree, "argmax_compared": int(decided.sum()),
          "near_ties": int((fg & ~decided).sum()), "foreground_spots": int(fg.sum()),
          "sample": "%dx%d sub-grid (%d spots of %d px, %d genes), same state_dict and inputs on both sides; "
                    "forward + masked CE in tutorial mode (image f eval, count f and g train-mode BN)"
                    % (SUB_H, SUB_W, n, patch, GENES),
          "dtype": mfma, "gate": "abs_diff <= 1e-4 (f32)" if mfma == 'f32' else "reported only (fp16 operands)"}
    if decided.any():
        ce["agreement_rate"] = agree / int(decided.sum())
    if split_hip is not None:
        ce["split_operands"] = {"hip": split_hip[0], "abs_diff": abs(split_hip[0] - loss_ref),
                                "abs_diff_vs_fp32_instruction_path": abs(split_hip[0] - loss_hip),
                                "argmax_agree": int((split_hip[1][decided] == rows.argmax(1)[decided]).sum()),
                                "argmax_equal_to_fp32_instruction_path": int((split_hip[1] == preds_hip).sum()),
                                "what": "DenseNet.split_conv1 / split_conv2 on the same inputs and weights, against the same oracle result"}
    if timed_steps == 0:
        return None, ce
    base = {"value": n / dt, "unit": "spots/s", "cores": cores, "host_cores": host_cores, "cpu_model": cpu_model, "kind": "port",
            "sample": "1 warm-up (%.1f s) + %d timed training steps (%.1f s each) on a %dx%d sub-grid (%d spots of %d px, "
                      "%d genes), %d torch CPU threads (of %d physical cores: the oracle is 5x slower on 128 threads than on 32)"
                      % (warm, timed_steps, dt, SUB_H, SUB_W, n, patch, GENES, cores, host_cores)}
    return base, ce


# ------------------------------------------------------------------------------------------ tape size (f trained)
def tape_bytes(patch, spots):
    """HBM the f-trained step holds at its peak, in bytes: the raw tensors of the tape (stem map, block buffers, one
    bottleneck per layer) plus the backward's block-gradient buffers and scratch.  No recompute."""
    hs = (patch + 6 - 7) // 2 + 1
    s = _first_map(patch)
    c = DENSENET121['num_init_features']
    total = hs * hs * c                                      # conv0 map
    biggest = 0
    for n_layers in DENSENET121['block_config']:
        c_total = c + n_layers * DENSENET121['growth_rate']
        block = s * s * c_total
        total += block + n_layers * s * s * 128              # block buffer + every layer's bottleneck
        biggest = max(biggest, 2 * block + 2 * s * s * 128)  # its gradient, tC scratch, tA, tB
        c = c_total // 2
        s //= 2
    return 4 * spots * (total + biggest + 3 * patch * patch)


# ------------------------------------------------------------------------------------------ feed from host memory
def from_host_series(args, model, optimizer, criterion, device, rank, world):
    """The same tutorial-mode step fed the way `train_gridwise` is fed in real use: a Dataset of arrays living in pageable
    host memory (uint8 patches: 245 MB per 128-px array; float counts; labels) -> DataLoader(batch_size=1) ->
    prefetch.DevicePrefetcher (pinned staging ring, non_blocking H2D on a side stream one batch ahead, event-ordered) ->
    the step.  ToTensor's / 255 runs inside the stem kernel."""
    import torch
    from torch.utils.data import DataLoader, Dataset
    from gridnext_amd import distributed as gdist
    from gridnext_amd import prefetch
    from gridnext_amd import training as gtrain
    steps, warmup = args.steps, max(args.warmup, 3)
    host = []
    for a in range(args.arrays):
        gen = torch.Generator().manual_seed(5000 + 1000 * rank + a)
        y = torch.randint(0, CLASSES + 1, (H, W), generator=gen)
        x8 = torch.randint(0, 256, (H, W, 3, args.patch, args.patch), generator=gen, dtype=torch.uint8)
        x8 *= (y > 0).to(torch.uint8).view(H, W, 1, 1, 1)
        xc = torch.randint(0, 10, (GENES, H, W), generator=gen).float() * (y > 0).float().unsqueeze(0)
        host.append(((x8, xc), y))

    class Arrays(Dataset):
        def __len__(self):
            return steps + warmup

        def __getitem__(self, i):
            return host[i % len(host)]

    for p in model.patch_classifier.parameters():
        p.requires_grad = False
    stepped = gdist.optimizer_params(optimizer)
    model.train()
    model.patch_classifier.eval()
    pf = prefetch.DevicePrefetcher(DataLoader(Arrays(), batch_size=1), device)
    it = iter(pf)
    diag = _RankDiag(device)

    def step():
        inputs, labels = next(it)
        loss, _, _ = gtrain._grid_loss(model, inputs, labels, criterion, 1, True)
        loss.backward()
        diag.allreduce(stepped)
        optimizer.step()
        optimizer.zero_grad()
        return loss

    for _ in range(warmup):
        step()
    diag.reset()
    if gdist.is_active():
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize()
    own = time.perf_counter() - t0
    if gdist.is_active():
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    # every rank's own feed rate: bytes its prefetcher moved over PCIe during its own timed steps
    # (the producer thread runs batches ahead of the consumer, so bytes are counted per batch, not per interval)
    per_batch = pf.bytes_moved / float(steps + warmup)
    rank_diag = diag.report(own, steps, extra=[per_batch * steps / max(own, 1e-9) / 1e9])
    rank_diag["h2d_gb_per_s_per_rank"] = [e[0] for e in rank_diag.pop("extra_per_rank")]
    if gdist.is_active():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    for _ in it:                                                  # exhaust (nothing left) so the producer thread ends
        pass
    return {"value": H * W * world * steps / elapsed, "unit": "spots/s", "ms_per_step": 1e3 * elapsed / steps, "steps": steps,
            "warmup": warmup, "final_loss": float(last.item()), "ranks": rank_diag,
            "h2d_bytes_per_step": pf.bytes_moved // (steps + warmup),
            "workload": "the headline step fed from pageable host memory: Dataset -> DataLoader(batch_size=1) -> pinned "
                        "double-buffered H2D prefetcher (side stream, one batch ahead); uint8 patches (ToTensor fused into the "
                        "stem kernel), float32 counts, int64 labels; timed region includes collate, staging and PCIe"}


# ------------------------------------------------------------------------------------------ config 5 (fp16, 256 px)
def split_series(args, device, rank, world, steps=4, warmup=2, P=128):
    """The headline's step (config 4: f frozen, fp32 tensors) with conv1 and conv2 of every dense layer on SPLIT bf16 operands
    (`DenseNet.split_conv1 = split_conv2 = True`: every fp32 operand = hi + lo in bf16, three 16-bit matrix instructions per
    product, fp32 accumulation; csrc/conv1x1_split.hip, conv3x3_split.hip) - opt-in, NOT the headline.  Reported with the loss and the f logits of the same
    array on the fp32-instruction path, so that what the split costs in accuracy stands next to what it buys."""
    import torch
    import torch.nn as nn
    from gridnext_amd import distributed as gdist
    from gridnext_amd import training as gtrain
    model = build_model(device, P)
    gdist.broadcast_module(model)
    for p in model.patch_classifier.parameters():
        p.requires_grad = False
    opt = torch.optim.Adam(model.corrector.parameters(), lr=1e-3)
    crit = nn.CrossEntropyLoss()
    gen = torch.Generator(device=device).manual_seed(700 + rank)
    y = torch.randint(0, CLASSES + 1, (1, H, W), device=device, generator=gen)
    x = torch.rand((1, H, W, 3, P, P), device=device, generator=gen)
    xc = torch.randint(0, 10, (1, GENES, H, W), device=device, generator=gen).float()
    f_img = model.image_classifier
    model.train()
    model.patch_classifier.eval()
    stepped = gdist.optimizer_params(opt)
    cmp = {}
    with torch.no_grad():
        for name, flag in (("fp32", False), ("split", True)):
            f_img.split_conv1 = f_img.split_conv2 = flag
            cmp[name] = (float(gtrain._grid_loss(model, [x, xc], y, crit, 1, True)[0].item()),
                         f_img(x.reshape(-1, 3, P, P)).double())
    d = (cmp["split"][1] - cmp["fp32"][1]).abs().max().item()
    rng = cmp["fp32"][1].abs().max().item()
    same = int((cmp["split"][1].argmax(1) == cmp["fp32"][1].argmax(1)).sum().item())
    f_img.split_conv1 = f_img.split_conv2 = True

    def step():
        loss, _, _ = gtrain._grid_loss(model, [x, xc], y, crit, 1, True)
        loss.backward()
        gdist.allreduce_gradients(stepped)
        opt.step()
        opt.zero_grad()
        return loss

    for _ in range(warmup):
        step()
    f_img._probe = []
    if gdist.is_active():
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize()
    if gdist.is_active():
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if gdist.is_active():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    probe, f_img._probe = f_img._probe, None
    kt = kernel_table(probe, P, steps)
    k1 = kt.get('conv1x1')
    if k1:
        k1["kernel"] = "conv1x1_split_kernel (bf16 hi/lo operands, 3 x v_mfma_f32_32x32x16_bf16 per product)"
        k1["bound"], k1["peak"], k1["unit"] = "hbm", PEAK_HBM_GBS, "GB/s"
        k1["achieved"] = k1["algorithmic_gbs"]
        k1["frac"] = k1["achieved"] / PEAK_HBM_GBS
        k1["executed_matrix_tflops"] = 3 * k1["flops_per_launch_avg"] * k1["launches"] / (k1["ms_per_step"] * steps * 1e-3) / 1e12
    k3 = kt.get('conv3x3')
    if k3:
        k3["kernel"] = "conv3x3_split_kernel (nine shifted products of bf16 hi/lo operands)"
        k3["bound"], k3["peak"], k3["unit"] = "hbm", PEAK_HBM_GBS, "GB/s"
        k3["achieved"] = k3["algorithmic_gbs"]
        k3["frac"] = k3["achieved"] / PEAK_HBM_GBS
        k3["executed_matrix_tflops"] = 3 * k3["flops_per_launch_avg"] * k3["launches"] / (k3["ms_per_step"] * steps * 1e-3) / 1e12
    # PMC traffic of the two kernels (tools/profile_round.sh <tag> split), attached while their sources are the profiled ones
    named = {('conv1x1_split' if k == 'conv1x1' else 'conv3x3_split'): kt[k] for k in ('conv1x1', 'conv3x3') if k in kt}
    attach_traffic(named, '_split')
    return {"value": H * W * world * steps / elapsed, "unit": "spots/s", "ms_per_step": 1e3 * elapsed / steps, "steps": steps,
            "warmup": warmup, "dtype": "f32 tensors and accumulation; conv1 / conv2 products as three bf16 matrix instructions (hi/lo split)",
            "final_loss": float(last.item()), "patch": P,
            "vs_fp32_instruction_path": {"ce_fp32": cmp["fp32"][0], "ce_split": cmp["split"][0],
                                         "abs_diff_ce": abs(cmp["split"][0] - cmp["fp32"][0]),
                                         "f_logits_max_abs_diff": d, "f_logits_range": rng, "argmax_equal": same, "spots": H * W},
            "kernels": {k: kt[k] for k in ('conv1x1', 'conv3x3') if k in kt},
            "workload": "the headline's step (config 4, 128-px fp32 patches resident, f frozen, g trained) with DenseNet.split_conv1 / split_conv2"}


def config5_series(args, device, rank, world, steps=4, warmup=2, P=256):
    """BASELINE config 5's step on this GPU: the same multimodal f + g step with 256-px patches and the fp16 MFMA conv path
    (`DenseNet.mfma = 'f16'`: fp16 matrix operands incl. the stem, fp16 block buffers, fp32 accumulate), uint8 patches
    resident in HBM, running statistics calibrated on one batch (a freshly initialised network with untouched statistics
    overflows fp16).  Priced against HBM: its conv kernels multiply 16x faster than they can be fed."""
    import torch
    import torch.nn as nn
    from gridnext_amd import distributed as gdist
    from gridnext_amd import training as gtrain
    model = build_model(device, P)
    gdist.broadcast_module(model)
    for p in model.patch_classifier.parameters():
        p.requires_grad = False
    opt = torch.optim.Adam(model.corrector.parameters(), lr=1e-3)
    crit = nn.CrossEntropyLoss()
    gen = torch.Generator(device=device).manual_seed(900 + rank)
    y = torch.randint(0, CLASSES + 1, (1, H, W), device=device, generator=gen)
    x8 = torch.randint(0, 256, (1, H, W, 3, P, P), device=device, generator=gen, dtype=torch.uint8)
    xc = torch.randint(0, 10, (1, GENES, H, W), device=device, generator=gen).float()
    f_img = model.image_classifier
    f_img.mfma = 'f16'
    bns = [m for m in f_img.modules() if isinstance(m, nn.BatchNorm2d)]
    moms = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0
    f_img.train()
    with torch.no_grad():
        f_img(x8.reshape(-1, 3, P, P)[:32])
    for m, mo in zip(bns, moms):
        m.momentum = mo
    model.train()
    model.patch_classifier.eval()
    stepped = gdist.optimizer_params(opt)

    def step():
        loss, _, _ = gtrain._grid_loss(model, [x8, xc], y, crit, 1, True)
        loss.backward()
        gdist.allreduce_gradients(stepped)
        opt.step()
        opt.zero_grad()
        return loss

    for _ in range(warmup):
        step()
    f_img._probe = []
    if gdist.is_active():
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize()
    if gdist.is_active():
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if gdist.is_active():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    probe, f_img._probe = f_img._probe, None
    out = {"value": H * W * world * steps / elapsed, "unit": "spots/s", "ms_per_step": 1e3 * elapsed / steps, "steps": steps,
           "warmup": warmup, "dtype": "f16", "final_loss": float(last.item()), "fp16_block_buffers": bool(f_img._used_f16_buffers),
           "patch": P,
           "workload": ("BASELINE config 5 on one GPU" if P == 256 else "the headline's geometry on the fp16 MFMA conv path") +
                       ": multimodal f (DenseNet-121 @%d px, fp16 MFMA conv path) + count MLP + hex g, "
                       "1 array (4992 spots) per step, f frozen/eval, g trained; uint8 patches resident in HBM" % P}
    # SURVEY 8d for this config: dCE and agreement rate after g against the fp32 CPU oracle (13 x 8 = 104-spot sub-grid, the
    # same state_dict - calibrated statistics included - uint8 patches on the HIP side); reported, no 1e-4 claim
    if rank == 0 and world == 1 and not getattr(args, 'no_cpu_baseline', False):
        _, out["ce_vs_ref"] = cpu_leg(model, P, 'f16', device, timed_steps=0, sub_hw=(13, 8), u8=True)
        out["fused_dense_layers"] = bool(getattr(f_img, '_used_f16_fused', False))
    if os.environ.get('GNX_PROBE_DUMP'):                         # per-launch records in launch order (tools/diag/c5_layers.py)
        with open(os.environ['GNX_PROBE_DUMP'], 'w') as fh:
            json.dump([{"kind": r[0], "ms": r[1].elapsed_time(r[2]), "flops": r[3], "bytes": r[4]} for r in (probe or [])], fh)
    kt = kernel_table(probe, P, steps)
    for kind in kt:                                              # fp16 operands (2 B, stated per launch); priced against HBM
        kk = kt[kind]
        gbs = kk["algorithmic_bytes_per_launch_avg"] / (kk["avg_launch_ms"] * 1e-3) / 1e9
        kk.update({"bound": "hbm", "matrix_tflops": kk["achieved"], "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                   "frac": gbs / PEAK_HBM_GBS, "kernel": {"conv1x1": "conv1x1_h16_kernel", "conv3x3": "conv3x3_dma_kernel<H16, O16>",
                                                          "dense_layer": "dense_layer_f16_kernel / dense_layer_f16_s64_kernel"}.get(kind, kk["kernel"])})
        kk.pop("algorithmic_gbs", None)
    if P == 256:
        attach_traffic(kt, '_f16_256')                          # (the PMC passes ran this geometry)
    if kt:
        order = sorted(kt, key=lambda k: -kt[k]["ms_per_step"])
        out["roofline"] = dict(kt[order[0]])
        if len(order) > 1:
            out["roofline"]["second_kernel"] = kt[order[1]]
    del model, x8, xc
    torch.cuda.empty_cache()
    return out


def patch224_series(args, device, rank, world, steps=3, warmup=1, P=224):
    """The headline tutorial-mode step at the REFERENCE's own patch geometry, (3, 224, 224) (/root/reference/scripts/
    multimodal_model_test.py:32-36, notebooks/Tutorial_visium_image.ipynb; maps 56 / 28 / 14 / 7; 5.666 GFLOP per spot): fp32,
    f frozen / eval, g trained, uint8 patches resident in HBM, one 78 x 64 array per step."""
    import torch
    import torch.nn as nn
    from gridnext_amd import distributed as gdist
    from gridnext_amd import training as gtrain
    model = build_model(device, P)
    gdist.broadcast_module(model)
    for p in model.patch_classifier.parameters():
        p.requires_grad = False
    opt = torch.optim.Adam(model.corrector.parameters(), lr=1e-3)
    crit = nn.CrossEntropyLoss()
    gen = torch.Generator(device=device).manual_seed(700 + rank)
    y = torch.randint(0, CLASSES + 1, (1, H, W), device=device, generator=gen)
    x8 = torch.randint(0, 256, (1, H, W, 3, P, P), device=device, generator=gen, dtype=torch.uint8)
    xc = torch.randint(0, 10, (1, GENES, H, W), device=device, generator=gen).float()
    f_img = model.image_classifier
    model.train()
    model.patch_classifier.eval()
    stepped = gdist.optimizer_params(opt)

    def step():
        loss, _, _ = gtrain._grid_loss(model, [x8, xc], y, crit, 1, True)
        loss.backward()
        gdist.allreduce_gradients(stepped)
        opt.step()
        opt.zero_grad()
        return loss

    for _ in range(warmup):
        step()
    f_img._probe = []
    if gdist.is_active():
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize()
    if gdist.is_active():
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if gdist.is_active():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    probe, f_img._probe = f_img._probe, None
    flops_per_spot = 5.666e9 if P == 224 else None
    out = {"value": H * W * world * steps / elapsed, "unit": "spots/s", "ms_per_step": 1e3 * elapsed / steps, "steps": steps,
           "warmup": warmup, "dtype": "f32", "patch": P, "final_loss": float(last.item()),
           "workload": "the headline step at the reference's own patch size: multimodal f (DenseNet-121 @%d px, fp32, frozen/eval) + "
                       "count MLP + hex g, 1 array (4992 spots) per step, uint8 patches resident in HBM" % P}
    if flops_per_spot:
        out["algorithmic_tflops"] = flops_per_spot * H * W * world * steps / elapsed / 1e12
        out["frac_of_fp32_matrix_peak"] = out["algorithmic_tflops"] / PEAK_F32_MATRIX_TFLOPS
    kt = kernel_table(probe, P, steps)
    if getattr(f_img, 'winograd', False) and 'conv3x3' in kt:
        kt['conv3x3']["kernel"] = "conv3x3 (Winograd F(2,3) where the map allows, direct otherwise)"
    if kt:
        order = sorted(kt, key=lambda k: -kt[k]["ms_per_step"])
        out["roofline"] = dict(kt[order[0]])
        if len(order) > 1:
            out["roofline"]["second_kernel"] = kt[order[1]]
    del model, x8, xc
    torch.cuda.empty_cache()
    return out


def config5_trained_series(args, device, rank, world, steps=3, warmup=1, P=256, probe_dump=None, compare_fp32=True):
    """SURVEY 8d's "everything trained" column at config 5's geometry (22.21 GFLOP per spot): the multimodal step on one
    256-px array with both classifiers trained through f_opt on the fp16-MFMA path - `DenseNet.mfma = 'f16'` on the gradient
    path: the taped forward is the fused dense-layer kernel on channel-blocked fp16 buffers (tape: block buffers + activated
    bottlenecks, one chunk), fp16-MFMA backward on the same buffers with fp32 accumulation
    and fp32 parameter gradients, power-of-two loss scale per backward (gridnext_amd/densenet_train_f16.py).  Running
    statistics calibrated on one batch as in `config5_series`.  The SAME steps (same initial state, same inputs) then run on
    the fp32 HIP gradient path (recomputed chunks): `ce_vs_fp32_path` is BASELINE's "CE vs ref" for this series - the loss
    after every optimizer step on both paths."""
    import torch
    import torch.nn as nn
    from gridnext_amd import distributed as gdist
    from gridnext_amd import training as gtrain
    from gridnext_amd import densenet_train as dt
    crit = nn.CrossEntropyLoss()
    gen = torch.Generator(device=device).manual_seed(950 + rank)
    y = torch.randint(0, CLASSES + 1, (1, H, W), device=device, generator=gen)
    x8 = torch.randint(0, 256, (1, H, W, 3, P, P), device=device, generator=gen, dtype=torch.uint8)
    xc = torch.randint(0, 10, (1, GENES, H, W), device=device, generator=gen).float()

    def run(mfma, n_warm, n_timed, probe_on):
        model = build_model(device, P)
        gdist.broadcast_module(model)
        f_img = model.image_classifier
        bns = [m for m in f_img.modules() if isinstance(m, nn.BatchNorm2d)]
        moms = [m.momentum for m in bns]
        for m in bns:
            m.momentum = 1.0
        f_img.train()
        with torch.no_grad():
            f_img(x8.reshape(-1, 3, P, P)[:32])
        for m, mo in zip(bns, moms):
            m.momentum = mo
        f_img.mfma = mfma
        free, _ = torch.cuda.mem_get_info(device)
        f_img.tape_budget = min(f_img.tape_budget, int(0.55 * free))
        opt = torch.optim.Adam(model.corrector.parameters(), lr=1e-3)
        f_opt = torch.optim.Adam(list(model.image_classifier.parameters()) + list(model.count_classifier.parameters()), lr=1e-4)
        model.train()
        model.patch_classifier.eval()
        stepped = gdist.optimizer_params(opt, f_opt)
        losses = []

        def step():
            loss, _, _ = gtrain._grid_loss(model, [x8, xc], y, crit, 1, True)
            loss.backward()
            gdist.allreduce_gradients(stepped)
            opt.step()
            opt.zero_grad()
            f_opt.step()
            f_opt.zero_grad()
            losses.append(loss.detach())
            return loss

        torch.cuda.reset_peak_memory_stats(device)
        for _ in range(n_warm):
            step()
        if probe_on:
            f_img._probe = []
        if gdist.is_active():
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_timed):
            step()
        torch.cuda.synchronize()
        if gdist.is_active():
            torch.distributed.barrier()
        elapsed = time.perf_counter() - t0
        if gdist.is_active():
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            elapsed = float(t.item())
        probe, f_img._probe = f_img._probe, None
        if mfma == 'f16':
            from gridnext_amd import densenet_train_f16 as dt16
            per = dt16.tape_bytes_per_spot(f_img, P)
        else:
            per = dt.tape_bytes_per_spot(f_img, P)
        chunk = H * W if per * H * W <= f_img.tape_budget else max(8, int(f_img.tape_budget // per) // 8 * 8)
        info = {"elapsed": elapsed, "losses": [float(v.item()) for v in losses], "probe": probe,
                "peak_hbm_gb": torch.cuda.max_memory_allocated(device) / 1e9, "chunks": -(-H * W // chunk), "chunk_spots": chunk,
                "overflow": int(f_img.f16_grad_overflow.item()) if 'f16_grad_overflow' in f_img.__dict__ else None,
                "loss_scale": float(f_img.f16_grad_scale[0].item()) if 'f16_grad_scale' in f_img.__dict__ else None}
        del model, opt, f_opt, f_img
        torch.cuda.empty_cache()
        return info

    r16 = run('f16', warmup, steps, True)
    elapsed = r16["elapsed"]
    out = {"value": H * W * world * steps / elapsed, "unit": "spots/s", "ms_per_step": 1e3 * elapsed / steps, "steps": steps,
           "warmup": warmup, "dtype": "f16", "final_loss": r16["losses"][-1], "peak_hbm_gb": r16["peak_hbm_gb"],
           "recomputed_chunks": 0 if r16["chunks"] == 1 else r16["chunks"], "chunk_spots": r16["chunk_spots"],
           "loss_scale": r16["loss_scale"], "grad_overflow_flag": r16["overflow"],
           "algorithmic_tflops": (22.21e9 if P == 256 else 5.557e9) * H * W * world * steps / elapsed / 1e12, "patch": P,
           "workload": "%s, everything trained: multimodal f (DenseNet-121 @%d px, fp16-MFMA forward AND "
                       "backward, fp32 accumulate / parameter gradients) + count MLP + hex g, 1 array (4992 spots) per step, f and "
                       "g trained (f_opt), eval-mode BN (calibrated statistics), uint8 patches resident in HBM"
                       % ("BASELINE config 5's geometry" if P == 256 else "config 4's geometry on config 5's fp16 path", P)}
    if world == 1 and compare_fp32:
        r32 = run('f32', 1, warmup + steps - 1, False)           # (its first step - allocator growth - is not in its timing)
        out["ce_vs_fp32_path"] = {"f16": r16["losses"], "f32": r32["losses"],
                                  "abs_diff": [abs(a - b) for a, b in zip(r16["losses"], r32["losses"])],
                                  "what": "masked CE of the array before optimizer step k = 0, 1, ... on the fp16-MFMA path and on "
                                          "the fp32 HIP gradient path; same initial state_dict, inputs and optimizers",
                                  "fp32_path_ms_per_step": 1e3 * r32["elapsed"] / max(warmup + steps - 1, 1),
                                  "fp32_path_recomputed_chunks": r32["chunks"], "fp32_path_peak_hbm_gb": r32["peak_hbm_gb"]}
    probe = r16["probe"]
    if probe_dump:                                           # per-launch records, in launch order (tools/bench_c5_trained.py --layers)
        with open(probe_dump, 'w') as fh:
            json.dump([{"kind": r[0], "ms": r[1].elapsed_time(r[2]), "flops": r[3], "bytes": r[4]} for r in (probe or [])], fh)
    KINDS16 = {'dense_layer_tape': 'dense_layer_f16_kernel / dense_layer_f16_s64_kernel (taped forward: norm1 .. conv2 in ONE kernel per '
                                   'dense layer on channel-blocked buffers; also copies the activated bottleneck tile out of the LDS)',
               'conv3x3_bwd_f16': 'conv3x3_bwd_f16_kernel (conv2: data gradient + norm2 adjoint + weight gradient, one pass over dY and A)',
               'wgrad3x3_f16': 'wgrad3x3_f16_kernel', 'dgrad3x3_bn2_f16': 'dgrad3x3_bn_f16_kernel',
               'wgrad1x1_f16': 'wgrad1x1_f16_kernel', 'dgrad1x1_bn1_f16': 'dgrad1x1_bn_f16_kernel<false>',
               'dgrad_wgrad1x1_bn1_f16': 'dgrad1x1_bn_f16_kernel<true> (conv1 data gradient + norm1 adjoint + conv1 weight gradient)',
               'stem_bwd_f16': 'stem_bwd_f16_kernel (conv0 rows recomputed + pool0 / norm0 adjoint + conv0 weight gradient, + its '
                               'two slab reductions)'}
    MFMA_BOUND = ('stem_bwd_f16',)        # 3 kB of patch per spot-row against 2 x 1.5 x 64 x 147 x 2 flops per conv0 position
    kt = {}
    for kind, name in KINDS16.items():
        recs = [r for r in (probe or []) if r[0] == kind]
        if not recs:
            continue
        ms = sum(r[1].elapsed_time(r[2]) for r in recs)
        flops, nbytes = sum(r[3] for r in recs), sum(r[4] for r in recs)
        gbs = nbytes / (ms * 1e-3) / 1e9
        tfl = flops / (ms * 1e-3) / 1e12
        kt[kind] = {"bound": "hbm", "kernel": name, "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                    "traffic": None, "launches": len(recs), "avg_launch_ms": ms / len(recs),
                    "algorithmic_bytes_per_launch_avg": nbytes / len(recs), "matrix_tflops": tfl, "ms_per_step": ms / steps}
        if kind in MFMA_BOUND:
            kt[kind].update({"bound": "mfma", "achieved": tfl, "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                             "frac": tfl / PEAK_F16_TFLOPS})
    if P == 256:
        attach_traffic(kt, '_c5trained')
    if kt:
        order = sorted(kt, key=lambda k: -kt[k]["ms_per_step"])
        out["roofline"] = dict(kt[order[0]])
        out["roofline"]["other_kernels"] = {k: {f: kt[k][f] for f in ("kernel", "achieved", "frac", "ms_per_step", "launches")}
                                            for k in order[1:]}
    del x8, xc
    torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------ configs 1-3
def other_configs(device):
    """BASELINE configs 1-3 driven through train_spotwise / train_gridwise on synthetic, device-resident data
    (tools/bench_configs.py): C1 count-MLP spot loop (batch 128), C2 DenseNet-121 @128 px spot loop (batch 32 as in the tutorial, and 256;
    train-mode BatchNorm, forward + backward + Adam), C3 count f (frozen) + hex g grid loop (batch 1).  Loop output is swallowed."""
    import contextlib
    import io
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import bench_configs as bc
    out = {}
    with contextlib.redirect_stdout(io.StringIO()):
        for key, fn in (("config1_count_mlp_spotwise_b128", lambda: bc.c1(3)),
                        ("config3_count_f_hex_g_gridwise_b1", lambda: bc.c3(40)),
                        # (the same loop with the user's optimizer built as torch.optim.Adam(fused=True): the plain one's host
                        #  side, ~170 us per step, bounds a count-only step whose device work is ~250 us)
                        ("config3_count_f_hex_g_gridwise_b1_fused_adam", lambda: bc.c3(40, fused_adam=True)),
                        ("config2_densenet121_spotwise_b32", lambda: bc.c2(2048, 1)),
                        # (the tutorial's batch of 32 is launch-bound - ~1 040 kernels of 7-25 us per step; the same loop at
                        #  batch 256 says what the kernels do when a step has work for the chip)
                        ("config2_densenet121_spotwise_b256", lambda: bc.c2(4096, 1, 256))):
            r = fn()
            out[key] = {"value": r["spots_per_s"], "unit": "spots/s", "seconds": r["seconds"], "workload": r["config"]}
            torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------ one worker = one GPU
def worker_main(args):
    import torch
    import torch.nn as nn
    from gridnext_amd import distributed as gdist
    from gridnext_amd import training as gtrain
    from gridnext_amd.synthetic import visium_array
    threads = _pin_to_rank_cpus()                                 # N > 1: this rank's slice of the host cores
    if threads:
        torch.set_num_threads(max(1, threads))
    rank, world, device = gdist.init_from_env(args.backend)
    assert torch.cuda.is_available(), "bench.py measures the HIP path; no HIP device visible"
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)

    if args.train_f:
        free, _ = torch.cuda.mem_get_info(device)
        # a tape larger than DenseNet.tape_budget goes through in recomputed chunks (densenet_train._RecomputeFn): the inputs
        # and one 8-spot chunk must fit
        need = tape_bytes(args.patch, 8) + args.arrays * 4 * H * W * (3 * args.patch ** 2 + GENES)
        if need > free:
            raise SystemExit("bench.py --train-f --patch %d: %.0f GB of resident inputs + one chunk's tape do not fit the "
                             "%.0f GB of free HBM. Use fewer resident --arrays." % (args.patch, need / 1e9, free / 1e9))

    model = build_model(device, args.patch)
    gdist.broadcast_module(model)
    optimizer = torch.optim.Adam(model.corrector.parameters(), lr=1e-3)
    criterion = nn.CrossEntropyLoss()

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

    diag = _RankDiag(device)

    def run_series(train_f, steps, warmup, probe_on):
        """`warmup` untimed + `steps` timed steps; returns (seconds for the timed steps - max over ranks -, last loss,
        probe records)."""
        f_opt = None
        params = list(model.image_classifier.parameters()) + list(model.count_classifier.parameters())
        if train_f:
            for p in params:
                p.requires_grad = True
            f_opt = torch.optim.Adam(params, lr=1e-4)
        else:
            for p in model.patch_classifier.parameters():           # Tutorial_multimodal.ipynb cell 27
                p.requires_grad = False
        stepped = gdist.optimizer_params(optimizer, f_opt)
        model.train()
        model.patch_classifier.eval()

        def step(i):
            inputs, labels = arrays[i % len(arrays)]
            loss, correct, n_fg = gtrain._grid_loss(model, inputs, labels, criterion, 1, True)
            loss.backward()
            diag.allreduce(stepped)
            optimizer.step()
            optimizer.zero_grad()
            if f_opt is not None:
                f_opt.step()
                f_opt.zero_grad()
            return loss

        f_img._probe = None
        for i in range(warmup):
            step(i)
        diag.reset()
        if probe_on:
            f_img._probe = []                                       # (kind, start_event, end_event) per timed launch
        if gdist.is_active():
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            last = step(warmup + i)
        torch.cuda.synchronize()
        own = time.perf_counter() - t0
        if gdist.is_active():
            torch.distributed.barrier()
        elapsed = time.perf_counter() - t0
        if gdist.is_active():
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            elapsed = float(t.item())
        run_series.last_ranks = diag.report(own, steps)
        probe, f_img._probe = f_img._probe, None
        return elapsed, float(last.item()), probe

    # ---- CE vs reference + CPU baseline (rank 0, N = 1 only), BEFORE any optimizer step: initial weights on both sides
    cpu_base = ce = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_base, ce = cpu_leg(model, args.patch, args.mfma, device)
        if args.mfma == 'f32':
            assert ce["abs_diff"] <= 1e-4, "CE of the HIP path differs from the CPU oracle by %.3e (> 1e-4)" % ce["abs_diff"]
    # ---- the same comparison at the benchmark's OWN size: the first resident array, whole (forward + CE on the CPU: 1-3 min)
    ce_full = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_full_grid_ce and args.patch <= 128:
        (xi, xc), yy = arrays[0]
        t0 = time.time()
        ce_full = full_grid_ce(model, args.patch, args.mfma, device, inputs=(xi.cpu(), xc.cpu(), yy.cpu()))
        ce_full["seconds"] = time.time() - t0
        if args.mfma == 'f32':
            assert ce_full["abs_diff"] <= 1e-4, "full-grid CE of the HIP path differs from the CPU oracle by %.3e" % ce_full["abs_diff"]
            assert ce_full["argmax_agree"] == ce_full["argmax_compared"], "full-grid argmax disagreement on decided spots"

    elapsed, last_loss, probe = run_series(args.train_f, args.steps, args.warmup, not args.no_kernel_timing)

    spots = H * W * world * args.steps
    names = [None] * world
    mine = "rank %d: %s (cuda:%d)" % (rank, torch.cuda.get_device_name(device), device.index)
    if gdist.is_active():
        torch.distributed.all_gather_object(names, mine)
    else:
        names = [mine]
    mode = ("f AND g trained (f_opt; DenseNet forward+backward, eval-mode BN as training.py:126)" if args.train_f
            else "f frozen/eval (tutorial mode)")
    result = {
        "metric": "spots/sec training throughput (multimodal f+g)",
        "value": spots / elapsed, "unit": "spots/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.mfma, "data": "synthetic",
        "config": {"workload": "C4: multimodal f(DenseNet-121 @%dpx + count-MLP 2000 genes) + hex g on 78x64 Visium "
                               "grids, 1 array (4992 spots) per GPU per step, %s, "
                               "g trained with Adam, masked CE" % (args.patch, mode),
                   "arrays_per_gpu_per_step": 1, "spots_per_array": H * W, "parallelism": "dp%d" % world,
                   "final_loss": last_loss},
        "rccl_ranks": world, "backend": (torch.distributed.get_backend() if gdist.is_active() else "none (1 process)"),
        "devices": names, "ranks": run_series.last_ranks,
        "threads_per_rank": threads or torch.get_num_threads(),
        # f-trained steps: the DenseNet's gradients are all-reduced bucket by bucket from inside its backward
        # (distributed.BackwardReducer); frozen-f steps have one 108 KB flat call into a persistent buffer
        "allreduce_overlapped": bool(gdist.BackwardReducer.wanted()),
    }
    if probe and os.environ.get('GNX_PROBE_DUMP'):               # per-launch records in launch order (tools/diag/c5_layers.py)
        with open(os.environ['GNX_PROBE_DUMP'], 'w') as fh:
            json.dump([{"kind": r[0], "ms": r[1].elapsed_time(r[2]), "flops": r[3], "bytes": r[4]} for r in probe], fh)
    if probe:
        kern = kernel_table(probe, args.patch, args.steps)
        # HBM bytes per launch from rocprofv3 PMC passes of this same command (FETCH_SIZE and WRITE_SIZE in separate
        # runs, KiB units, FETCH doubled for 16-B/lane loads as MI355X_MICROARCH.md prescribes): tools/pmc_traffic.py,
        # profiles/README.md
        if args.patch == 128 and args.mfma == 'f32':
            attach_traffic(kern, '_trainf' if args.train_f else '')
        elif args.patch == 256 and args.mfma == 'f16' and not args.train_f:
            attach_traffic(kern, '_f16_256')
        if getattr(f_img, 'winograd', False) and args.mfma == 'f32':
            # The Winograd launches execute 2/3 of the direct-convolution multiply-adds: `achieved` / `frac` are the
            # EXECUTED matrix FLOPs against the matrix peak (a fraction of peak must be work the pipe did); the
            # direct-convolution credit (the algorithmic figure of SURVEY 8d) is kept under `direct_conv_*`.
            winograd_credit(kern['conv3x3'], args.patch, args.steps)
        if args.mfma == 'f16':
            # config 5's kernels multiply 16x faster than they can be fed: they are priced against HBM (algorithmic bytes
            # per launch / launch time); the fp32-FLOP figure stays in `matrix_tflops` for reference
            h16 = bool(getattr(f_img, '_used_f16_buffers', False))     # block buffers in fp16: every operand is 2 B
            for kind, name in (('dense_layer', 'dense_layer_f16_kernel / dense_layer_f16_s64_kernel (norm1 .. conv2 in one kernel, bottleneck in LDS)'),
                               ('conv1x1', 'conv1x1_h16_kernel' if h16 else 'conv1x1_f16_kernel'),
                               ('conv3x3', 'conv3x3_dma_kernel<H16, O16>' if h16 else 'conv3x3_dma_kernel<H16> / conv3x3_f16_kernel')):
                if kind not in kern:
                    continue
                kk = kern[kind]
                kk["fp16_block_buffers"] = h16
                gbs = kk["algorithmic_bytes_per_launch_avg"] / (kk["avg_launch_ms"] * 1e-3) / 1e9
                kk.update({"bound": "hbm", "kernel": name, "matrix_tflops": kk["achieved"], "achieved": gbs,
                           "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS})
        order = sorted(kern, key=lambda k: -kern[k]["ms_per_step"])
        result["roofline"] = dict(kern[order[0]])
        if len(order) > 1:
            result["roofline"]["second_kernel"] = kern[order[1]]
        if len(order) > 2:
            result["roofline"]["other_kernels"] = {k: {f: kern[k][f] for f in ("kernel", "achieved", "frac", "ms_per_step",
                                                                                "launches")} for k in order[2:]}
    if ce is not None:
        result["ce_vs_ref"] = ce
    if ce_full is not None:
        result["ce_vs_ref_full_grid"] = ce_full
    if cpu_base is not None:
        result["cpu_baseline"] = cpu_base

    # ---- the real loop feed: arrays in pageable HOST memory, uint8 patches, through a DataLoader and the
    #      pinned double-buffered prefetcher the training loops use; the timed region includes collate, staging and PCIe
    def optional(name, fn):
        """An optional series must never cost the headline line: in a single process its failure is recorded instead of
        raised (with several ranks a failure still ends the run - the ranks are inside collectives together)."""
        try:
            result.setdefault("series", {})[name] = fn()
        except Exception as exc:                                  # noqa: BLE001
            if world > 1:
                raise
            result.setdefault("series", {})[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            torch.cuda.empty_cache()

    if (args.from_host or not args.no_series) and not args.train_f:
        optional("from_host", lambda: from_host_series(args, model, optimizer, criterion, device, rank, world))

    # ---- BASELINE config 5's geometry in the same run: 256-px patches, fp16 MFMA conv path (fp16 block buffers, fp16 stem)
    # (a SCALE run - N > 1 - carries the headline and the host-fed series only, unless --all-series: the other series are
    #  single-GPU characterisations and would lengthen every N for numbers nobody reads there)
    more = not args.no_series and (world == 1 or args.all_series)
    if more and not args.train_f and args.mfma == 'f32' and args.patch == 128:
        optional("config5_f16_256px", lambda: config5_series(args, device, rank, world))
        # the headline's own geometry (128-px patches, f frozen) on the fp16 MFMA conv path: what that path buys where the
        # fp32 number is quoted; its CE is reported against the fp32 oracle, not gated at 1e-4 (dtype f16)
        optional("headline_geometry_f16_128px", lambda: config5_series(args, device, rank, world, P=128))
        optional("headline_split_128px", lambda: split_series(args, device, rank, world))

    # ---- the reference's own patch size (224 px)
    if more and not args.train_f and args.mfma == 'f32' and args.patch == 128:
        optional("patch224_f32", lambda: patch224_series(args, device, rank, world))

    # ---- and its "everything trained" column: 256 px, fp32, f in recomputed chunks (bounded tape)
    if more and not args.train_f and args.mfma == 'f32' and args.patch == 128:
        optional("config5_everything_trained_256px", lambda: config5_trained_series(args, device, rank, world))

    # ---- the f-trained step of the headline geometry (128 px) on the fp16-MFMA gradient path
    if more and not args.train_f and args.mfma == 'f32' and args.patch == 128:
        optional("train_f_f16_128px", lambda: config5_trained_series(args, device, rank, world, P=128))

    # ---- second series of SURVEY 8d in the same run: f trained (DenseNet forward + backward)
    if not args.train_f and more and args.mfma == 'f32' and args.patch == 128:
        free, _ = torch.cuda.mem_get_info(device)
        if tape_bytes(args.patch, H * W) < free:
            def train_f_series():
                el, loss_tf, probe_tf = run_series(True, args.series_steps, 1, not args.no_kernel_timing)
                ser = {"value": H * W * world * args.series_steps / el, "unit": "spots/s",
                       "ms_per_step": 1e3 * el / args.series_steps, "steps": args.series_steps, "warmup": 1,
                       "workload": "the same step with both classifiers trained through f_opt (Adam, lr 1e-4): DenseNet-121 "
                                   "forward with tape + full backward, eval-mode BN (training.py:126)",
                       "final_loss": loss_tf}
                if probe_tf:
                    kt = kernel_table(probe_tf, args.patch, args.series_steps)
                    if getattr(f_img, 'winograd', False) and 'conv3x3' in kt:
                        winograd_credit(kt['conv3x3'], args.patch, args.series_steps)
                    attach_traffic(kt, '_trainf')
                    order = sorted(kt, key=lambda k: -kt[k]["ms_per_step"])
                    ser["roofline"] = dict(kt[order[0]])
                    ser["roofline"]["other_kernels"] = {k: {f: kt[k][f] for f in ("kernel", "achieved", "frac",
                                                                                   "ms_per_step", "launches")}
                                                        for k in order[1:]}
                return ser
            optional("train_f", train_f_series)

            def train_f_split_series():
                # the same f-trained step with the opt-in split-operand kernels: taped forward convs and conv1's weight gradient
                keep = (f_img.split_conv1, f_img.split_conv2, f_img.split_wgrad)
                f_img.split_conv1 = f_img.split_conv2 = f_img.split_wgrad = True
                try:
                    el, loss_tf, _ = run_series(True, args.series_steps, 1, False)
                finally:
                    f_img.split_conv1, f_img.split_conv2, f_img.split_wgrad = keep
                return {"value": H * W * world * args.series_steps / el, "unit": "spots/s",
                        "ms_per_step": 1e3 * el / args.series_steps, "steps": args.series_steps, "warmup": 1,
                        "dtype": "f32 tensors and accumulation; forward conv1 / conv2 and conv1's weight gradient as three bf16 matrix "
                                 "instructions per product (hi/lo split); data gradients and conv2's weight gradient on the fp32 instruction",
                        "workload": "series train_f with DenseNet.split_conv1 / split_conv2 / split_wgrad", "final_loss": loss_tf}
            optional("train_f_split", train_f_split_series)

    # ---- the other BASELINE configs through the product's own loops (tools/bench_configs.py), single process only
    if world == 1 and not args.no_series and not args.train_f and args.mfma == 'f32' and args.patch == 128:
        optional("other_configs", lambda: other_configs(device))
    if rank == 0:
        # ONE lean JSON line on stdout (the contract); the further series - each with its own roofline object - go to a
        # file (default gpurun_out/bench_series.json; the copy of the round's final run is committed as profiles/rNN_series.json)
        # and, one summary line each, to stderr
        series = result.pop("series", None)
        if series:
            path = args.series_out or os.path.join(ROOT, 'gpurun_out', 'bench_series.json')
            try:
                os.makedirs(os.path.dirname(path), exist_ok=True)
                with open(path, 'w') as fh:
                    json.dump({"headline": {k: result[k] for k in ("value", "unit", "ms_per_step", "n_gpus", "steps", "dtype")},
                               "series": series}, fh, indent=1)
                result["series_file"] = os.path.relpath(path, ROOT)
            except OSError as exc:
                result["series_file"] = "not written (%s)" % exc
            result["series_summary"] = {}
            for name, ser in series.items():
                if isinstance(ser, dict) and "value" in ser:
                    result["series_summary"][name] = {"value": ser["value"], "unit": ser.get("unit"), "ms_per_step": ser.get("ms_per_step")}
                    sys.stderr.write("series %-36s %12.1f %s  %s ms/step\n" % (name, ser["value"], ser.get("unit", ""), ser.get("ms_per_step")))
                elif isinstance(ser, dict) and "error" in ser:
                    result["series_summary"][name] = {"error": ser["error"]}
                    sys.stderr.write("series %-36s FAILED: %s\n" % (name, ser["error"]))
                elif isinstance(ser, dict):
                    result["series_summary"][name] = {k: v.get("value") for k, v in ser.items() if isinstance(v, dict)}
        for key in ("cpu_baseline", "ce_vs_ref", "ce_vs_ref_full_grid"):       # (last in the line: the driver keeps its tail)
            if key in result:
                result[key] = result.pop(key)
        print(json.dumps(result), flush=True)
    if gdist.is_active():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


# ------------------------------------------------------------------------------------------ launcher (no GPU, no torch)
def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_cpus(rank, world):
    """The host cores rank `rank` of `world` pins itself to: a contiguous slice of the cores this process may use - with
    the usual enumeration (socket 0's cores first) contiguous slices are NUMA-local, and GPUs 0-3 / 4-7 hang off sockets
    0 / 1 on an 8-GPU MI355X node.  Hyper-thread siblings (the second half of the enumeration) go with their cores."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return []
    if world <= 1 or len(allowed) < world:
        return []
    per = len(allowed) // world
    return allowed[rank * per:(rank + 1) * per]


def _pin_to_rank_cpus():
    """Apply GNX_BENCH_CPUS (set by launch(); under torchrun: derived from LOCAL_RANK / LOCAL_WORLD_SIZE) - affinity and the
    intra-op thread count.  Returns the number of threads per rank (None: untouched)."""
    spec = os.environ.get('GNX_BENCH_CPUS')
    cpus = [int(c) for c in spec.split(',')] if spec else \
        _rank_cpus(int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1'))))
    if not cpus:
        return None
    try:
        os.sched_setaffinity(0, cpus)
    except (AttributeError, OSError):
        return None
    return len(cpus)


def launch(n, argv):
    """Start `n` worker processes (one per GPU) and wait for them.  This process never initialises the GPU: it imports
    neither torch nor the package; children are fresh interpreters (never exec'd over a process that touched the card)."""
    port = os.environ.get('MASTER_PORT') or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY='0')
        # every rank gets its own slice of the host cores: 8 ranks with torch's default intra-op pool (all cores each)
        # contend in the host-fed series' collate / staging (worker_main pins itself to GNX_BENCH_CPUS)
        cpus = _rank_cpus(r, n)
        if cpus:
            env['GNX_BENCH_CPUS'] = ','.join(str(c) for c in cpus)
            env.setdefault('OMP_NUM_THREADS', str(len(cpus)))
            env.setdefault('MKL_NUM_THREADS', str(len(cpus)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), '--worker'] + argv, env=env))
    failed = 0
    pending = dict(enumerate(procs))
    while pending:
        for r, p in list(pending.items()):
            rc = p.poll()
            if rc is None:
                continue
            del pending[r]
            if rc != 0:
                failed = failed or rc
                sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, rc))
                for q in pending.values():           # exact PIDs of our own children, never a pattern
                    q.terminate()
        time.sleep(0.2)
    return failed


def parse(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--patch', type=int, default=128)
    ap.add_argument('--arrays', type=int, default=2, help='distinct synthetic arrays resident per GPU')
    ap.add_argument('--train-f', action='store_true',
                    help='headline line = the second series (SURVEY 8d): both classifiers trained through f_opt')
    ap.add_argument('--mfma', default='f32', choices=['f32', 'f16'],
                    help="matrix-core operand type of the DenseNet convs; 'f16' = BASELINE config 5's fp16 MFMA path "
                         "(fp32 accumulate; NOT the headline, reported as dtype f16)")
    ap.add_argument('--backend', default=None, help='torch.distributed backend (default nccl = RCCL)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--full-grid-ce', action='store_true',
                    help='(default at N = 1) CE vs the CPU oracle on ONE whole 78x64 array: `ce_vs_ref_full_grid`')
    ap.add_argument('--no-full-grid-ce', action='store_true', help='skip the whole-array CE comparison (1-3 min of CPU forward)')
    ap.add_argument('--no-series', action='store_true', help='skip the extra series appended to the default run')
    ap.add_argument('--all-series', action='store_true',
                    help='with --gpus N > 1: also run the single-GPU characterisation series on every rank')
    ap.add_argument('--from-host', action='store_true',
                    help='(with --no-series) still run the fed-from-host-memory series: uint8 patches through the prefetcher')
    ap.add_argument('--series-steps', type=int, default=3)
    ap.add_argument('--series-out', default=None, help='where the further series go as JSON (default gpurun_out/bench_series.json)')
    ap.add_argument('--worker', action='store_true', help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    under_launcher = args.worker or 'RANK' in os.environ or int(os.environ.get('WORLD_SIZE', '1')) > 1
    if args.gpus > 1 and not under_launcher:
        sys.exit(launch(args.gpus, argv))
    worker_main(args)


if __name__ == '__main__':
    main()
