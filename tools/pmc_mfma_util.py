#!/usr/bin/env python3
"""Matrix-pipe busy fraction per kernel from a rocprofv3 PMC pass of bench.py:
    rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d gpurun_out/<tag>_mfma ... -- python3 bench.py ...
    python tools/pmc_mfma_util.py gpurun_out/<tag>_mfma > profiles/<tag>_pmc_mfma_util.json
busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / 32 (MI355X_MICROARCH.md: the counter sums over 4 SIMDs x 8 XCD
samples); busy_over_insts = 64 for v_mfma_f32_32x32x2_f32 (64 cycles per instruction and SIMD)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.match(r'_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?_kernel)(ILb([01])E)?', name)       # names hipcc's demangler gives up on (_Float16)
    if m:
        return m.group(1) + ('<%s>' % ('true' if m.group(3) == '1' else 'false') if m.group(2) else '')
    m = re.search(r'(\w+_kernel)(<[^>]*>)?', name)
    return (m.group(1) + (m.group(2) or '')) if m else name[:60]


def main(d):
    f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not f:
        raise SystemExit('no counter_collection.csv under ' + d)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    names = {}
    for r in csv.DictReader(open(f[0])):
        per[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
        names[r['Dispatch_Id']] = short(r['Kernel_Name'])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for did, c in per.items():
        a = agg[names[did]]
        a['launches'] += 1
        for k, v in c.items():
            a[k] += v
    out = {}
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', 0)):
        if a.get('SQ_INSTS_MFMA', 0) <= 0 or a.get('SQ_BUSY_CYCLES', 0) <= 0:
            continue
        out[k] = {"launches": int(a['launches']),
                  "mfma_busy_frac": round(a['SQ_VALU_MFMA_BUSY_CYCLES'] / a['SQ_BUSY_CYCLES'] / 32, 3),
                  "mfma_insts_per_launch": int(a['SQ_INSTS_MFMA'] / a['launches']),
                  "busy_over_insts": round(a['SQ_VALU_MFMA_BUSY_CYCLES'] / a['SQ_INSTS_MFMA'], 1)}
    print(json.dumps({"note": __doc__.split('\n')[0], "kernels": out}, indent=1))


if __name__ == '__main__':
    main(sys.argv[1])
