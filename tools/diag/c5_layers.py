"""Per-launch table of the config-5 everything-trained step from `tools/bench_c5_trained.py --layers FILE`: for every kernel kind,
the launches of ONE step in order with their achieved algorithmic TB/s - shows which layer shapes a kernel handles badly."""
import json, sys, collections
recs = json.load(open(sys.argv[1]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
by = collections.defaultdict(list)
for r in recs:
    by[r['kind']].append(r)
for kind, rs in by.items():
    n = len(rs) // steps
    print(kind, n, 'launches per step')
    for i in range(n):
        ms = sum(rs[i + s * n]['ms'] for s in range(steps)) / steps
        b, f = rs[i]['bytes'] or 0, rs[i]['flops'] or 0
        print("   %3d  %7.3f ms  %6.2f TB/s  %7.1f TFLOP/s  %8.1f MB" % (i, ms, b / ms / 1e9, f / ms / 1e9, b / 1e6))
