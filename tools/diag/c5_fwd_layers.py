"""Per-launch table of config 5's forward (bench.py's `config5_f16_256px` series): GNX_PROBE_DUMP=file, then c5_layers.py."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('GNX_PROBE_DUMP', os.path.join(ROOT, 'gpurun_out', 'c5_fwd_layers.json'))
import bench
from gridnext_amd import distributed as gdist
rank, world, device = gdist.init_from_env(None)
out = bench.config5_series(argparse.Namespace(no_cpu_baseline=True), device, rank, world, steps=3, warmup=1)
print(out["value"], out["ms_per_step"])
