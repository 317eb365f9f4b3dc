#!/usr/bin/env python3
"""Folds a rocprofv3 --pmc pass of tools/b1_levels.py per experiment: the tool prints `SEQ <index> <label> <launches>`
after each experiment, the counter CSV lists the dispatches in order; this prints the mean of every counter per experiment
(HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB, the gfx950 correction of MI355X_MICROARCH.md).

    python tools/b1_levels_pmc.py <log.txt> <counter_collection.csv> [more csv ...]"""
import collections
import csv
import re
import sys

log, csvs = sys.argv[1], sys.argv[2:]
seqs = []
times = {}
last = None
for line in open(log):
    m = re.match(r"SEQ (\d+) (.*) (\d+)$", line.strip())
    if m:
        seqs.append((int(m.group(1)), m.group(2), int(m.group(3))))
        last = m.group(2)
        continue
    m = re.search(r"([\d.]+) us  frac", line)
    if m and last is not None:
        times[last] = float(m.group(1))
        last = None
for path in csvs:
    rows = [r for r in csv.DictReader(open(path)) if "ca_step_b1_kernel" in r["Kernel_Name"]]
    by_counter = collections.defaultdict(list)
    for r in rows:
        by_counter[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    print(f"== {path}")
    for name, vals in sorted(by_counter.items()):
        vals.sort()
        v = [x for _, x in vals]
        need = sum(c for _, _, c in seqs)
        print(f"  {name}: {len(v)} dispatches (log: {need})")
        pos = 0
        for idx, label, cnt in seqs:
            chunk = v[pos + cnt - min(cnt, 100):pos + cnt]      # the last 100 launches of the experiment (steady state)
            pos += cnt
            if not chunk:
                break
            mean = sum(chunk) / len(chunk)
            extra = ""
            if name in ("FETCH_SIZE", "WRITE_SIZE"):
                extra = f"  = {mean * (2 if name == 'FETCH_SIZE' else 1) / 1024:8.1f} MiB per launch" + \
                        (" (doubled: gfx950 correction)" if name == "FETCH_SIZE" else "")
            print(f"    {label:52s} {mean:14.0f}{extra}   [{times.get(label, float('nan')):6.2f} us unprofiled]")
