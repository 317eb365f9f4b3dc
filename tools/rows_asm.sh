#!/bin/bash
# Compiles carle_amd/csrc/ca_rnd_rows.hip to ISA (/tmp/asm/ca_rnd_rows.s) and prints the resource usage per instantiation
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DCARLE_BUILD_ARCH=\"gfx950\" -Wall -Wno-unused-function -S --cuda-device-only -Rpass-analysis=kernel-resource-usage $EXTRA -o /tmp/asm/ca_rnd_rows.s /root/repo/carle_amd/csrc/ca_rnd_rows.hip 2>&1 | grep -i "remark\|error" | sed 's/.*remark: //' | grep "error\|Name\|Spill\|Scratch\|SGPRs:\|LDS" | sed 's/\[-Rpass.*//'
