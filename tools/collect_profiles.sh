#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's numbers on the GPU box (run through gpurun):
#   kernel-trace stats of the headline bench (cfg2), its Infinity-Cache-free variant (N=65536), cfg4
#   (--rnd), cfg5 (b1 512^2), and the PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace
#   only) for the dominant kernels of cfg2 / cfg2 at N=65536 / cfg4 / cfg5.
# Outputs land in gpurun_out/prof/ ; tools/pmc_summary.py folds the PMC passes into
# profiles/pmc_latest.json (copied back through gpurun_out/prof/pmc_latest.json).
# Usage: tools/collect_profiles.sh [round-tag]      (default r05)
set -e
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() {   # name, bench flags...
    name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extra-configs > $O/$name.log 2>&1
    cp $(find $O/$name -name "*kernel_stats.csv" | head -1) $O/${TAG}_${name}_kernel_stats.csv
    grep '^{' $O/$name.log > $O/${TAG}_${name}_bench.json || true
    echo "done stats $name"
}
pmc() {     # name, kernel substring, instances, size, format, bench flags...
    name=$1; kern=$2; inst=$3; size=$4; fmt=$5; shift 5
    for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${name}_$c -- python3 $R/bench.py "$@" --steps 16 --warmup 4 --no-cpu-baseline --no-extra-configs > $O/pmc_${name}_$c.log 2>&1
        cp $(find $O/pmc_${name}_$c -name "*counter_collection.csv" | head -1) $O/pmc_${name}_$c.csv
    done
    python3 $R/tools/pmc_summary.py $O/pmc_${name}_FETCH_SIZE.csv $O/pmc_${name}_WRITE_SIZE.csv "$kern" $inst $size $fmt \
        "profiles/${TAG}_pmc.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py $*)" >> $O/${TAG}_pmc.txt
    echo "done pmc $name"
}
# (the PMC passes first: bench.py reports `traffic` from profiles/pmc_latest.json only when that record was taken on the
# kernel sources of the library it runs, so the bench lines collected below carry this run's traffic figures)
: > $O/${TAG}_pmc.txt
# (round 5: the step reads its state bit-packed -- the PACKED_IN instantiation, write-back stores at N=4096, streaming at 65536)
pmc cfg2 "ca_step_64_packed2_kernel" 4096 64 f32
pmc cfg2_n65536 "ca_step_64_kernel<carle::F32Cells, carle::F32Cells, true, true, true, 64, false, true, true, true>" 65536 64 f32 --instances 65536
pmc cfg4 "ca_step_rnd64_kernel<true, false, true, true, true, true>" 4096 64 f32 --rnd
pmc cfg5 "ca_step_b1_kernel" 4096 512 b1 --format b1 --size 512 --instances 4096 --rule B36/S23
pmc f32_256 "ca_step_wide_kernel" 4096 256 f32 --size 256
stats cfg2_f32 --steps 256 --warmup 16
stats cfg2_f32_n65536 --instances 65536 --steps 64 --warmup 8
stats cfg4_rnd --rnd --steps 256 --warmup 16
stats cfg5_b1 --format b1 --size 512 --instances 4096 --rule B36/S23 --steps 64 --warmup 8
stats f32_256 --size 256 --steps 64 --warmup 8
stats rnd_ae_fwd --rnd --ae --steps 256 --warmup 16
stats train_rnd_ae --rnd --ae --train --steps 256
stats default_geom_rnd_ae --size 256 --instances 16 --rnd --ae --steps 64 --warmup 8
stats default_geom_rnd_ae_train --size 256 --instances 16 --rnd --ae --train --steps 64 --warmup 8
python3 $R/tools/train_kernel_time.py > $O/${TAG}_train_kernel_time.txt 2>&1
python3 $R/tools/agent_time.py > $O/${TAG}_agent_net.txt 2>&1
{ python3 $R/tools/kbench_b1.py 4096 512 0 0; python3 $R/tools/kbench_b1.py 8192 512 0 0; python3 $R/tools/kbench_b1.py 16384 256 0 0; } 2>&1 | grep -v amdgpu.ids > $O/${TAG}_kbench_b1_final.txt
python3 $R/tools/profile_host.py --stack rnd+ae --size 256 --instances 16 --burst 200 2>&1 | grep -v amdgpu.ids > $O/${TAG}_host_default_geom.txt
for i in 1 2 3; do python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs 2>/dev/null | grep '^{'; done > $O/${TAG}_driver_style_bench.json
# the default run as the driver starts it: headline + the other BASELINE workloads under "configs" + the CPU baseline
python3 $R/bench.py 2>/dev/null | grep '^{' > $O/${TAG}_default_bench_line.json
# the same headline with every exchange going through a ONE-rank RCCL group
python3 $R/bench.py --single-rank-group --rnd --no-cpu-baseline --no-extra-configs 2>/dev/null | grep '^{' > $O/${TAG}_single_rank_rccl_bench.json
cp $R/profiles/pmc_latest.json $O/pmc_latest.json
echo "all done"
