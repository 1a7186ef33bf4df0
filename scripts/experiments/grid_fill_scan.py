#!/usr/bin/env python3
"""Round 6: launches that do not fill the chip.  From a rocprofv3 --kernel-trace CSV: per kernel name, the time spent in launches of fewer than 256 / 512
workgroups (256 CUs; most kernels here want >= 2 resident workgroups per CU).     python3 grid_fill_scan.py <kernel_trace.csv> [skip launches]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = rows[skip:]
tot = collections.defaultdict(float); small = collections.defaultdict(float); mid = collections.defaultdict(float); n = collections.Counter(); ex = {}
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]); w = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    wgs = g // max(w, 1)
    k = r["Kernel_Name"][:100]
    tot[k] += d; n[k] += 1
    if wgs < 256: small[k] += d; ex[k] = wgs
    elif wgs < 512: mid[k] += d
T = sum(tot.values())
print(f"total {T / 1e3:.2f} ms; in launches of < 256 workgroups: {sum(small.values()) / 1e3:.2f} ms; 256..511: {sum(mid.values()) / 1e3:.2f} ms")
for k in sorted(tot, key=lambda k: -(small[k] + 0.5 * mid[k]))[:25]:
    if small[k] + mid[k] > 0: print(f"{k:100s} total {tot[k]:9.1f} us  <256 WGs: {small[k]:9.1f} us  256-511: {mid[k]:9.1f} us  (e.g. {ex.get(k, '-')} WGs)")
