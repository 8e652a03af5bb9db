"""Worker process of tests/test_gpu_data_parallel.py: one data-parallel rank driving the HIP path
(GridNetHexMM: tiny DenseNet + count MLP + hex g, fused masked CE) through `train_gridwise`.

    python tests/dp_gpu_worker.py <out_dir> <backend>          # RANK / WORLD_SIZE / MASTER_* / GNX_* from the environment

GNX_TEST_FOPT = f32 | f16: f is TRAINED too (`f_opt`), so the DenseNet's HIP backward runs under the process group and hands
its gradients to `distributed.BackwardReducer` block by block (gridnext_amd/densenet_train.py, densenet_train_f16.py); `f16`
uses a network of the geometry the fp16-MFMA gradient path takes (growth 32, bottleneck 128, 64 stem channels, 128-px patches).
backend `none`: a plain single process (no process group) - the run the data-parallel ones are compared with.
"""
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch                                                    # noqa: E402
import torch.nn as nn                                           # noqa: E402
from torch.utils.data import DataLoader                         # noqa: E402

TINY = dict(growth_rate=4, block_config=(2, 2), num_init_features=8, bn_size=2, num_classes=5, small_inputs=False)
WIDE = dict(growth_rate=32, block_config=(2, 2, 2, 2), num_init_features=64, bn_size=4, num_classes=5, small_inputs=False)
G, C = 20, 5
FOPT = os.environ.get('GNX_TEST_FOPT', '')


def make_problem(seed_model, fopt=FOPT):
    import gridnext_amd as ga
    from gridnext_amd.synthetic import count_mlp
    torch.manual_seed(seed_model)
    H, W, P = 6, 4, 32                                          # (locals: the problem depends on `fopt` only, never on call order)
    if fopt == 'f16':
        H, W, P = 4, 2, 128                                     # 8 spots per array: one group of the fp16 kernels
        dn = ga.DenseNet(**WIDE)
        dn.mfma = 'f16'
    else:
        dn = ga.DenseNet(**TINY)
    m = ga.GridNetHexMM(dn, count_mlp(G, C), (3, P, P), (G,), (H, W), C)
    gen = torch.Generator().manual_seed(77)
    xi = torch.rand(4, H, W, 3, P, P, generator=gen)
    xc = torch.randint(0, 10, (4, G, H, W), generator=gen).float()
    y = torch.randint(0, C + 1, (4, H, W), generator=gen)
    return m, xi, xc, y


def main():
    out_dir, backend = sys.argv[1], sys.argv[2]
    from gridnext_amd import distributed as gdist
    from gridnext_amd.training import train_gridwise
    rank, world, dev = gdist.init_from_env(backend=None if backend == 'none' else backend)
    assert dev.type == 'cuda'
    m, xi, xc, y = make_problem(1000 + rank)                    # every rank starts from DIFFERENT weights ...
    m.to(dev)
    with torch.no_grad():                                       # ... and has already run an eval forward (derived-weight
        m.eval()                                                # caches are warm with its own initialisation)
        before = m.forward_nhwc([xi[:1].to(dev), xc[:1].to(dev)]).cpu()
    gdist.broadcast_module(m)                                   # rank 0's weights everywhere; caches must not survive
    with torch.no_grad():
        after = m.forward_nhwc([xi[:1].to(dev), xc[:1].to(dev)]).cpu()
    f_opt = None
    if FOPT:
        f_opt = torch.optim.SGD(list(m.image_classifier.parameters()) + list(m.count_classifier.parameters()), lr=0.01)
    else:
        for p in m.patch_classifier.parameters():
            p.requires_grad = False
    data = [((xi[i], xc[i]), y[i]) for i in range(4)]
    dl = {'train': DataLoader(data, batch_size=1, sampler=gdist.ShardedSampler(data)),
          'val': DataLoader(data, batch_size=1, sampler=gdist.ShardedSampler(data))}
    opt = torch.optim.SGD(m.corrector.parameters(), lr=0.05)
    reducer_wanted = bool(gdist.BackwardReducer.wanted())
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        m, vh, th = train_gridwise(m, dl, nn.CrossEntropyLoss(), opt, num_epochs=2, f_opt=f_opt)
    ic = m.image_classifier
    torch.save({'state': {k: v.cpu() for k, v in m.state_dict().items()}, 'vh': vh, 'th': th, 'printed': buf.getvalue(),
                'before': before, 'after': after, 'world': world, 'reducer_wanted': reducer_wanted,
                'backend': torch.distributed.get_backend() if torch.distributed.is_initialized() else 'none',
                'f16_path_ran': 'f16_grad_scale' in ic.__dict__,
                'overflow': int(ic.f16_grad_overflow.item()) if 'f16_grad_overflow' in ic.__dict__ else None},
               os.path.join(out_dir, 'rank%d.pt' % rank))
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
