#!/usr/bin/env python3
"""Per-kernel sums of whatever counters a set of rocprofv3 --pmc passes collected (diagnostic):
    python tools/pmc_dump.py gpurun_out/<dir> [<dir> ...] [--match substring]
prints, per kernel name, launches and each counter's value per launch."""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r'(\w+_kernel)(<[^>]*>)?', name)
    return (m.group(1) + (m.group(2) or '')) if m else name[:60]


def main(argv):
    match = ''
    dirs = []
    it = iter(argv)
    for a in it:
        if a == '--match':
            match = next(it)
        else:
            dirs.append(a)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(set))
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r['Kernel_Name'])
                if match and match not in k:
                    continue
                agg[k][r['Counter_Name']] += float(r['Counter_Value'])
                launches[k][r['Counter_Name']].add((f, r['Dispatch_Id']))
    for k, c in agg.items():
        print(k)
        for name in sorted(c):
            n = len(launches[k][name])
            print("   %-34s %16.0f per launch (%d launches)" % (name, c[name] / n, n))


if __name__ == '__main__':
    main(sys.argv[1:])
