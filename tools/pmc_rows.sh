#!/bin/bash
# SQ / LDS counters of the two fused CA + RND2D kernel forms (rocprofv3 --pmc, kernel-trace only), N = 4096.
# Usage (GPU box): tools/pmc_rows.sh [out-dir]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=${1:-gpurun_out/pmc_rows}; case $O in /*) ;; *) O=$R/$O;; esac
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY \
    --kernel-trace --output-format csv -d $O/run -- python3 $R/tools/kbench_rnd_rows.py 4096 > $O/log.txt 2>&1
f=$(find $O/run -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' > $O/summary.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "rnd" not in k:
        continue
    acc[k.split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for name, v in sorted(c.items()):
        print(f"    {name:24s} mean {sum(v) / len(v):14.0f}  (n={len(v)})")
PY
cat $O/summary.txt
