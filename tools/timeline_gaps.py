#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of a bench.py run and prints, for the step / epilogue pairs of the timed loop,
the durations and the gaps between consecutive dispatches (where do the microseconds between two step kernels go?).
usage: timeline_gaps.py <kernel_trace.csv> [step-kernel-substring]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "ca_step_"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
d_step, g_sf, d_fin, g_fs = [], [], [], []
for (n0, s0, e0), (n1, s1, e1) in zip(seq, seq[1:]):
    if key in n0 and "ca_finish" in n1:
        d_step.append(e0 - s0); g_sf.append(s1 - e0); d_fin.append(e1 - s1)
    if "ca_finish" in n0 and key in n1:
        g_fs.append(s1 - e0)
print(f"{len(g_sf)} step -> epilogue pairs")
for name, v in (("step kernel", d_step), ("gap step -> epilogue", g_sf), ("epilogue kernel", d_fin), ("gap epilogue -> step", g_fs)):
    v = sorted(v)
    if v:
        print(f"  {name:22s} median {v[len(v) // 2]:7d} ns   p10 {v[len(v) // 10]:7d}   p90 {v[9 * len(v) // 10]:7d}")
