#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof256
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --size 256 --steps 16 --warmup 4 --no-cpu-baseline > $O/pmc_$c.log 2>&1
  cp $(find $O/pmc_$c -name "*counter_collection.csv" | head -1) $O/pmc_$c.csv
done
python3 $R/tools/pmc_summary.py $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv "ca_step_wide_kernel" 4096 256 f32 "f32 256" 
