#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs (one row per dispatch x counter x
dimension) -> a small CSV for profiles/.  usage: summarize_pmc.py out.csv in1.csv [in2.csv ...]"""
import collections
import csv
import re
import sys


def base(name):
    m = re.search(r'(\w+_kernel(?:<[\d, ]+>)?)', name)
    return m.group(1) if m else name[:40]


def main():
    out, ins = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    for path in ins:
        for r in csv.DictReader(open(path)):
            k = base(r['Kernel_Name'])
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            disp[k][r['Counter_Name']].add(r['Dispatch_Id'])
    counters = sorted({c for k in agg for c in agg[k]})
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'dispatches'] + counters)
        for k in sorted(agg):
            n = max(len(v) for v in disp[k].values())
            w.writerow([k, n] + ['%.0f' % (agg[k][c] / max(1, len(disp[k][c]))) if c in agg[k] else '' for c in counters])


if __name__ == '__main__':
    main()
