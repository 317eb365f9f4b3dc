#!/bin/bash
# The GPU suite six times, each time with the launch-time path choices of the library forced the other way
# (strip modes of the packed kernel, store / load cache hints of the fp32 kernels): every test then runs through
# code it would not reach at its own batch size.  Run on the GPU box: tools/test_all_paths.sh
set -e
BIG=999999999999
CARLE_B1_LDS_MIN_BYTES=0 CARLE_B1_NT_MIN_BYTES=0 CARLE_STREAM_STORE_MIN_BYTES=1 python -m pytest tests -m gpu -x -q
CARLE_B1_LDS_MIN_BYTES=0 CARLE_B1_NT_MIN_BYTES=$BIG CARLE_STREAM_STORE_MIN_BYTES=$BIG CARLE_RND_CACHED_LOAD_MAX_BYTES=$BIG python -m pytest tests -m gpu -x -q
CARLE_B1_LDS_MIN_BYTES=$BIG CARLE_RND_CACHED_LOAD_MAX_BYTES=0 python -m pytest tests -m gpu -x -q
CARLE_TWO_LAUNCH_STEP=1 python -m pytest tests -m gpu -x -q      # step kernel + epilogue everywhere (the form before the one-launch step)
CARLE_PACKED_STATE=0 python -m pytest tests -m gpu -x -q         # round 5: steps that always read the fp32 state
CARLE_PACKED_GRIDS=2 CARLE_STREAM_STORE_MIN_BYTES=1 python -m pytest tests -m gpu -x -q   # two grids per wavefront at every batch size, streaming stores
