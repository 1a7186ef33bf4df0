#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU pass (kernel-trace only) into
profiles/<tag>_mfma_busy.json, per kernel, with rocprofv3's own derived-metric formulas (gfx94x forms, MI355X_MICROARCH.md):
  MfmaUtil % = 100 * sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE per XCD * 1024 SIMDs)
  VALUBusy % = 100 * sum(SQ_ACTIVE_INST_VALU) / 256 CUs / GRBM_GUI_ACTIVE per XCD
GRBM_GUI_ACTIVE is reported summed over the 8 XCDs, hence / 8.
Usage (GPU box):  python scripts/collect_mfma_busy.py <tag> <counter_collection.csv>"""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_kinds import kind_of, sources_sha256
tag, path = sys.argv[1:3]
per = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
disp_seen = set()
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"]
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r.get("Dispatch_Id"))
    if key not in disp_seen:
        disp_seen.add(key); n[k] += 1
out = {}
for k, c in sorted(per.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0)):
    if not (k.startswith("void pd::") or k.startswith("pd::")):
        continue
    act = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if act <= 0:
        continue
    out[k] = {"launches_sampled": n[k], "gpu_cycles_per_launch": act / n[k],
              "MfmaUtil_percent": round(100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (act * 1024), 2),
              "VALUBusy_percent": round(100.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / 256 / act, 2)}
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for k, c in per.items():
    kd = kind_of(k)
    if kd:
        for name, v in c.items():
            agg[kd][name] += v
        agg[kd]["_n"] += n[k]
by_kind = {}
for kd, c in agg.items():
    act = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if act > 0:
        by_kind[kd] = {"launches_sampled": int(c["_n"]), "MfmaUtil_percent": round(100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (act * 1024), 2),
                       "VALUBusy_percent": round(100.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / 256 / act, 2)}
json.dump({"sources_sha256": sources_sha256(), "by_kind": by_kind, "note": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU over `python3 bench.py "
                   "--steps 1 --warmup 0 --batch 32 --inference-steps 2 --no-cpu-baseline --no-roofline` (sums over all launches of each "
                   "kernel; profiled passes run at a lower clock than un-profiled ones)", "kernels": out},
          open(f"profiles/{tag}_mfma_busy.json", "w"), indent=1)
for k, v in out.items():
    print(f"{k[:72]:72s} n={v['launches_sampled']:4d} MfmaUtil {v['MfmaUtil_percent']:6.2f} %  VALUBusy {v['VALUBusy_percent']:6.2f} %")
