#!/bin/bash
# bench.py over the configurations quoted in DESIGN.md section 6 (run through gpurun); one JSON line each,
# condensed to "label | env-steps/s | ms/step | dominant kernel us | roofline frac".
# Usage: tools/bench_matrix.sh [out-file]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out/bench_matrix.txt}
: > $OUT
run() {
    label=$1; shift
    python3 $R/bench.py "$@" 2>/dev/null | grep '^{' | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline()); r = d['roofline']
cb = d.get('cpu_baseline')
print('%-58s %10.3e env-steps/s %9.4f ms/step | %-28s %8.2f us frac %.3f%s' % ('$label', d['value'], d['ms_per_step'],
      r['kernel'].split('<')[0], r['avg_launch_us'], r['frac'],
      ('  | cpu port N=%s: %.3e (N=256: %.3e) on %d cores' % (cb['sample'].split('N=')[1].split()[0], cb['value'], cb.get('value_n256', 0), cb['cores'])) if cb else ''))
" >> $OUT
    tail -1 $OUT
}
run "cfg2: N=4096 64x64 f32, action ring (headline)" --steps 256 --warmup 16
run "cfg2 with fused device-side RandomAgent" --agent fused --no-cpu-baseline
run "cfg2 at N=65536 (1 GiB per array: no Infinity Cache)" --instances 65536 --steps 64 --warmup 8 --no-cpu-baseline
run "cfg4 (BASELINE configs[3]): + RND2D forward (fused kernel)" --rnd --no-cpu-baseline
run "cfg4 + AE2D forward" --rnd --ae --no-cpu-baseline
run "reference default geometry: N=16 256x256 + RND2D + AE2D forward" --size 256 --instances 16 --rnd --ae --steps 512 --no-cpu-baseline
run "reference default geometry, training mode (RND2D + AE2D forward+backward)" --size 256 --instances 16 --rnd --ae --train --steps 512 --no-cpu-baseline
run "reference default geometry forward, recorded calls (--replay calls)" --size 256 --instances 16 --rnd --ae --steps 512 --replay calls --no-cpu-baseline
run "reference default geometry bare: N=16 256x256" --size 256 --instances 16 --steps 512 --no-cpu-baseline
run "reference default geometry bare, recorded calls" --size 256 --instances 16 --steps 512 --replay calls --no-cpu-baseline
run "configs[0] through the per-step API: N=1 64x64" --instances 1 --steps 512 --no-cpu-baseline
run "configs[0] through the per-step API, recorded calls" --instances 1 --steps 512 --replay calls --no-cpu-baseline
run "cfg3 shape: N=4096 256x256 f32" --size 256 --steps 64 --warmup 8 --no-cpu-baseline
run "N=1024 512x512 f32" --size 512 --instances 1024 --steps 64 --warmup 8 --no-cpu-baseline
run "N=4096 256x256 uint8 (uint8 actions)" --format u8 --size 256 --steps 128 --warmup 8 --no-cpu-baseline
run "N=4096 64x64 uint8 (host-bound)" --format u8 --no-cpu-baseline
run "cfg5 (BASELINE configs[4]): N=4096 512x512 packed B36/S23" --format b1 --size 512 --rule B36/S23 --steps 128 --warmup 8 --no-cpu-baseline
run "N=4096 64x64 packed (cache-resident, host-bound)" --format b1 --no-cpu-baseline
run "training: RND2D forward+backward, Adam every 64" --rnd --train --no-cpu-baseline
run "training: RND2D + AE2D forward+backward" --rnd --ae --train --no-cpu-baseline
echo "matrix done"
