#!/usr/bin/env python3
"""Instruction mix of the kernels in a hipcc -S --cuda-device-only listing: static counts per kernel (VALU / SALU /
LDS / VMEM / waits / branches) and the resource lines (VGPRs, scratch, LDS).  Usage: asm_mix.py file.s [substring]"""
import re
import subprocess
import sys

text = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
i = 0
rows = []
while i < len(text):
    m = re.match(r"^(_Z\w+):\s*(;.*)?$", text[i])
    if not m:
        i += 1
        continue
    name = m.group(1)
    mix = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "smem": 0, "wait": 0, "branch": 0, "other": 0}
    j = i + 1
    while j < len(text) and not text[j].startswith(".Lfunc_end"):     # (a kernel may hold several s_endpgm)
        t = text[j].strip().split()
        if t and not t[0].startswith((";", ".", "_")) and not t[0].endswith(":"):
            op = t[0]
            if op.startswith("v_"):
                mix["valu"] += 1
            elif op.startswith(("s_waitcnt", "s_nop", "s_barrier")):
                mix["wait"] += 1
            elif op.startswith(("s_cbranch", "s_branch")):
                mix["branch"] += 1
            elif op.startswith(("s_load", "s_buffer_load")):
                mix["smem"] += 1
            elif op.startswith("s_"):
                mix["salu"] += 1
            elif op.startswith("ds_"):
                mix["lds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                mix["vmem"] += 1
            else:
                mix["other"] += 1
        j += 1
    res = {}
    while j < len(text) and not re.match(r"^_Z\w+:", text[j]):
        for key in ("NumVgprs", "ScratchSize", "LDSByteSize", "Occupancy", "NumSgprs"):
            mm = re.search(r";\s*" + key + r":\s*(\d+)", text[j])
            if mm and key not in res:
                res[key] = int(mm.group(1))
        j += 1
    rows.append((name, mix, res))
    i = j
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (name, mix, res), dn in zip(rows, names):
    if want and want not in dn:
        continue
    print(dn[:150])
    print("   ", " ".join(f"{k}={v}" for k, v in mix.items()), "|", " ".join(f"{k}={v}" for k, v in res.items()))
