#!/bin/bash
# Bit-packed step kernel alone (tools/kbench_b1.py) in its three strip modes at several array sizes: rows = row-per-lane
# loads / stores, lds = coalesced through the per-wave LDS image, lds+nt = the same with streaming hints.
run() { python tools/kbench_b1.py $1 $2 2>&1 | grep -v amdgpu | tail -3 | head -2 | sed "s/^/$3 /"; }
for cfg in "2048 512" "1024 512" "4096 512" "6144 512"; do
  set -- $cfg
  CARLE_B1_LDS_MIN_BYTES=999999999999 run $1 $2 "rows   "
  CARLE_B1_LDS_MIN_BYTES=0 CARLE_B1_NT_MIN_BYTES=999999999999 run $1 $2 "lds    "
  CARLE_B1_LDS_MIN_BYTES=0 CARLE_B1_NT_MIN_BYTES=0 run $1 $2 "lds+nt "
done
for cfg in "4096 256" "16384 256" "32768 256"; do
  set -- $cfg
  CARLE_B1_LDS_MIN_BYTES=999999999999 run $1 $2 "rows   "
  CARLE_B1_LDS_MIN_BYTES=0 CARLE_B1_NT_MIN_BYTES=999999999999 run $1 $2 "lds    "
  CARLE_B1_LDS_MIN_BYTES=0 CARLE_B1_NT_MIN_BYTES=0 run $1 $2 "lds+nt "
done
