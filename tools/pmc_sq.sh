#!/bin/bash
# SQ / LDS counters per kernel of one bench.py configuration (rocprofv3 --pmc, kernel-trace only; GPU box).
# Usage: tools/pmc_sq.sh <out-dir> <bench flags...>      e.g. tools/pmc_sq.sh gpurun_out/pmc_ae --rnd --ae
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$1; shift
case $O in /*) ;; *) O=$R/$O;; esac
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY \
    --kernel-trace --output-format csv -d $O/run -- python3 $R/bench.py "$@" --steps 32 --warmup 8 --no-cpu-baseline > $O/log.txt 2>&1
f=$(find $O/run -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' > $O/summary.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES", [0]))):
    n = len(next(iter(c.values())))
    if n < 8:
        continue
    print(k, f"(n={n})")
    for name, v in sorted(c.items()):
        print(f"    {name:24s} mean {sum(v) / len(v):14.0f}")
PY
cat $O/summary.txt
