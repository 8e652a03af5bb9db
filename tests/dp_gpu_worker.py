"""Worker process of tests/test_gpu_data_parallel.py: one data-parallel rank driving the HIP path
(GridNetHexMM: tiny DenseNet + count MLP + hex g, fused masked CE) through `train_gridwise`.

    python tests/dp_gpu_worker.py <out_dir> <backend>          # RANK / WORLD_SIZE / MASTER_* / GNX_* from the environment
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
G, H, W, P, C = 20, 6, 4, 32, 5


def make_problem(seed_model):
    import gridnext_amd as ga
    from gridnext_amd.synthetic import count_mlp
    torch.manual_seed(seed_model)
    m = ga.GridNetHexMM(ga.DenseNet(**TINY), count_mlp(G, C), (3, P, P), (G,), (H, W), C)
    gen = torch.Generator().manual_seed(77)
    xi = torch.rand(4, H, W, 3, P, P, generator=gen)
    xc = torch.randint(0, 10, (4, G, H, W), generator=gen).float()
    y = torch.randint(0, C + 1, (4, H, W), generator=gen)
    return m, xi, xc, y


def main():
    out_dir, backend = sys.argv[1], sys.argv[2]
    from gridnext_amd import distributed as gdist
    from gridnext_amd.training import train_gridwise
    rank, world, dev = gdist.init_from_env(backend=backend)
    assert dev.type == 'cuda'
    m, xi, xc, y = make_problem(1000 + rank)                    # every rank starts from DIFFERENT weights ...
    m.to(dev)
    with torch.no_grad():                                       # ... and has already run an eval forward (derived-weight
        m.eval()                                                # caches are warm with its own initialisation)
        before = m.forward_nhwc([xi[:1].to(dev), xc[:1].to(dev)]).cpu()
    gdist.broadcast_module(m)                                   # rank 0's weights everywhere; caches must not survive
    with torch.no_grad():
        after = m.forward_nhwc([xi[:1].to(dev), xc[:1].to(dev)]).cpu()
    for p in m.patch_classifier.parameters():
        p.requires_grad = False
    data = [((xi[i], xc[i]), y[i]) for i in range(4)]
    dl = {'train': DataLoader(data, batch_size=1, sampler=gdist.ShardedSampler(data)),
          'val': DataLoader(data, batch_size=1, sampler=gdist.ShardedSampler(data))}
    opt = torch.optim.SGD(m.corrector.parameters(), lr=0.05)
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        m, vh, th = train_gridwise(m, dl, nn.CrossEntropyLoss(), opt, num_epochs=2)
    torch.save({'state': {k: v.cpu() for k, v in m.state_dict().items()}, 'vh': vh, 'th': th, 'printed': buf.getvalue(),
                'before': before, 'after': after, 'backend': torch.distributed.get_backend(), 'world': world},
               os.path.join(out_dir, 'rank%d.pt' % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
