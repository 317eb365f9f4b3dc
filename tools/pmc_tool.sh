#!/bin/bash
# SQ / LDS counters per kernel of any python tool (rocprofv3 --pmc, kernel-trace only; GPU box).
# Usage: tools/pmc_tool.sh <out-dir> <tool.py> [args...]      e.g. tools/pmc_tool.sh gpurun_out/pmc_nets tools/nets64_time.py
# Two passes (the SQ block has eight counters per pass): activity, then waits.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$1; shift
T=$1; shift
case $O in /*) ;; *) O=$R/$O;; esac
case $T in /*) ;; *) T=$R/$T;; esac
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY \
    --kernel-trace --output-format csv -d $O/run1 -- python3 $T "$@" > $O/log1.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU \
    --kernel-trace --output-format csv -d $O/run2 -- python3 $T "$@" > $O/log2.txt 2>&1
python3 - $(find $O/run1 $O/run2 -name "*counter_collection.csv") <<'PY' > $O/summary.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
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
