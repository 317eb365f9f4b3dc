#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's numbers on the GPU box (run through gpurun):
#   kernel-trace stats of the default bench (cfg2), of cfg3 (--rnd), cfg5 (b1 512^2) and the uint8 format,
#   and the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) for cfg2.
# Outputs land in gpurun_out/prof/ ; copy the summaries into profiles/ afterwards.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() {   # name, bench flags...
    name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/$name.log 2>&1
    cp $(find $O/$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
    echo "done $name"
}
stats cfg2_f32 --steps 256 --warmup 16
stats cfg3_rnd --rnd --steps 256 --warmup 16
stats cfg5_b1 --format b1 --size 512 --instances 4096 --rule B36/S23 --steps 64 --warmup 8
stats u8_64 --format u8 --steps 256 --warmup 16
stats u8_256 --format u8 --size 256 --steps 64 --warmup 8
stats f32_256 --size 256 --steps 64 --warmup 8
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/pmc_$c.log 2>&1
    cp $(find $O/pmc_$c -name "*counter_collection.csv" | head -1) $O/pmc_$c.csv
    echo "done pmc $c"
done
