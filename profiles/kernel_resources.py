#!/usr/bin/env python3
"""Prints registers / spills / LDS / occupancy per kernel from `make -C raytracer_amd/csrc asm` remarks (stdin)."""
import re
import sys

cur = None
rows = {}
for line in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = v
        rows[cur] = {}
    elif cur:
        rows[cur][k.split(" [")[0]] = v
for name, r in rows.items():
    short = re.sub(r"^_ZN2rt\d+", "", name)
    short = re.sub(r"Ev?NS_8ColdArgs.*$", "", short)
    print("%-44s sgpr %-4s vgpr %-4s scratch %-5s sgpr_spill %-4s vgpr_spill %-4s occ %s" % (
        short[:44], r.get("TotalSGPRs"), r.get("VGPRs"), r.get("ScratchSize"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("Occupancy")))
