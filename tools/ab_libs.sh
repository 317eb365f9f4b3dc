#!/bin/bash
# Same-box A/B of library builds (gpurun boxes differ by several per cent; only numbers from one call compare).
# Build the variants here, copy each libcarle_hip.so to tmp_libs/<name>.so, then on the GPU box:
#   tools/ab_libs.sh "<command>" name1.so name2.so ...      (two rounds; <command> runs with each library installed)
cmd=$1; shift
for rep in 1 2; do
  for l in "$@"; do
    cp tmp_libs/$l carle_amd/lib/libcarle_hip.so
    echo "== $l"
    bash -c "$cmd"
  done
done
