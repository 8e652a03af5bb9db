#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes of bench.py (FETCH_SIZE and WRITE_SIZE need separate
passes: they share TCC counter slots), corrected as MI355X_MICROARCH.md prescribes for gfx950:
  * both counters tick in KiB;
  * FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced reads - global_load, buffer_load and
    LDS-DMA alike - so it is doubled;  WRITE_SIZE is exact (calibration: the stem's 5.23 GB output).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o r --output-format csv -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o r --output-format csv -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNNx_pmc_traffic.json
"""
import collections
import csv
import glob
import json
import os
import re
import sys

def _targs(n, kernel):
    """template arguments of `kernel` in a demangled name, as a list of strings ([] if it is another kernel)"""
    m = re.search(re.escape(kernel) + r'<([^>]*)>', n)
    return [a.strip() for a in m.group(1).split(',')] if m else []


GROUPS = [                                   # first match wins; names as bench.py's kernel kinds
    ('dgrad_wgrad1x1_bn1_f16', lambda n: 'dgrad1x1_bn_f16_kernel' in n and ('ILb1' in n or '<true>' in n)),
    ('dgrad1x1_bn1_f16', lambda n: 'dgrad1x1_bn_f16_kernel' in n),
    ('conv3x3_bwd_f16', lambda n: 'conv3x3_bwd_f16_kernel' in n),
    ('dgrad3x3_bn2_f16', lambda n: 'dgrad3x3_bn_f16_kernel' in n),
    ('wgrad3x3_f16', lambda n: 'wgrad3x3_f16_kernel' in n or 'wgrad3x3_f16_p2_kernel' in n),
    ('wgrad1x1_f16', lambda n: 'wgrad1x1_f16_kernel' in n),
    ('dgrad_wgrad1x1_bn1', lambda n: 'dgrad_wgrad1x1_f32_kernel' in n),
    ('dense_layer', lambda n: 'dense_layer_f16_kernel' in n or 'dense_layer_f16_s64_kernel' in n),
    ('dgrad1x1_bn1', lambda n: 'conv1x1_ws_kernel<false, false, 4, true>' in n),
    ('dgrad3x3_bn2', lambda n: len(_targs(n, 'conv3x3_dma_kernel')) >= 7 and _targs(n, 'conv3x3_dma_kernel')[6] == 'true'),
    ('wgrad1x1', lambda n: 'wgrad1_t_kernel' in n),
    ('wgrad3x3', lambda n: 'wgrad9_t_kernel' in n),
    ('conv1x1_split', lambda n: 'conv1x1_split_kernel' in n),
    ('conv3x3_split', lambda n: 'conv3x3_split_kernel' in n),
    ('split_pack', lambda n: 'split_pack_kernel' in n),
    ('conv1x1', lambda n: 'conv1x1_ws_kernel<true, false' in n or 'conv1x1_ws_kernel<false, false' in n
        or 'conv1x1_kernel<false' in n or 'conv1x1_h16' in n),
    ('transition', lambda n: 'conv1x1_ws_kernel<true, true' in n or 'conv1x1_kernel<true' in n or 'transition_f16_kernel' in n),
    ('conv3x3', lambda n: 'conv3x3_' in n),
    ('stem', lambda n: 'conv_stem' in n),
    ('maxpool', lambda n: 'maxpool' in n),
]


def collect(d, counter):
    f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not f:
        raise SystemExit('no counter_collection.csv under ' + d)
    per = collections.defaultdict(list)
    by_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] != counter:
            continue
        by_dispatch[r['Dispatch_Id']] += float(r['Counter_Value'])
        names[r['Dispatch_Id']] = r['Kernel_Name']
    for did, v in by_dispatch.items():
        for g, match in GROUPS:
            if match(names[did]):
                per[g].append(v * 1024.0)
                break
    return per


def source_hashes():
    """sha1 of every kernel source at the time of the PMC passes: bench.py attaches a traffic figure to a kernel only while
    the sources that kernel is built from still hash to these (a kernel change without a new pass must not ship a stale ratio)."""
    import hashlib
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gridnext_amd', 'csrc')
    out = {}
    for f in sorted(glob.glob(os.path.join(csrc, '*.hip')) + glob.glob(os.path.join(csrc, '*.h'))):
        with open(f, 'rb') as fh:
            out[os.path.basename(f)] = hashlib.sha1(fh.read()).hexdigest()
    return out


def main():
    fetch = collect(sys.argv[1], 'FETCH_SIZE')
    write = collect(sys.argv[2], 'WRITE_SIZE')
    out = {'_sources': source_hashes()}
    for g, _ in GROUPS:
        if g not in fetch or g not in write:
            continue
        fb = 2.0 * sum(fetch[g]) / len(fetch[g])
        wb = sum(write[g]) / len(write[g])
        out[g] = {'launches': len(fetch[g]), 'fetch_bytes_per_launch_corrected': fb, 'write_bytes_per_launch': wb,
                  'hbm_bytes_per_launch': fb + wb}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == '__main__':
    main()
