#!/bin/bash
# Same-box A/B of library builds on the fused CA + RND2D step: tools/kbench_rnd_ab.sh lib1.so lib2.so ... (files in tmp_libs/)
for rep in 1 2; do
  for l in "$@"; do
    cp tmp_libs/$l carle_amd/lib/libcarle_hip.so
    echo "== $l"
    CARLE_RND_ROWS_MIN_N=0 python tools/kbench_rnd_env.py 4096 2>/dev/null | grep "one launch" | tail -2
  done
done
