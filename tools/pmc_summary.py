#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_latest.json.

Usage: tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> \
           <kernel-name-substring> <instances> <size> <format> <source-note>
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE reports half of a wide coalesced read).

pmc_latest.json is a LIST of records, one per (kernel, workload); each carries the sha256 of the
kernel sources it was measured on (carle_amd._lib.source_hash), and bench.py only quotes a record
whose hash equals that of the sources in the tree (else roofline.traffic is null)."""
import csv
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mean_counter(path, kernel, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return statistics.mean(vals), len(vals)


def main():
    from carle_amd import _lib
    fetch_csv, write_csv, kernel, instances, size, fmt, source = sys.argv[1:8]
    f, nf = mean_counter(fetch_csv, kernel, "FETCH_SIZE")
    w, nw = mean_counter(write_csv, kernel, "WRITE_SIZE")
    rec = {"kernel": kernel, "instances": int(instances), "size": int(size), "format": fmt,
           "fetch_size_kb_raw": f, "write_size_kb": w, "launches": [nf, nw],
           "traffic_bytes_per_launch": int(round((2 * f + w) * 1024)), "source": source,
           "csrc_sha256": _lib.source_hash()}
    out = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        recs = json.load(open(out))
        if isinstance(recs, dict):
            recs = [recs]
    except (OSError, ValueError):
        recs = []
    key = (rec["kernel"], rec["instances"], rec["size"], rec["format"])
    recs = [r for r in recs if (r.get("kernel"), r.get("instances"), r.get("size"), r.get("format", "f32")) != key]
    recs.append(rec)
    json.dump(recs, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
