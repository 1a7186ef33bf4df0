#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<tag>_hbm_traffic.json.
Usage (GPU box):  python scripts/collect_traffic.py <tag> <fetch_csv> <write_csv>
Corrections per MI355X_MICROARCH.md (HBM section): counters are in KiB-like units of 1024 B; on gfx950 FETCH_SIZE
reports exactly half the bytes of wide coalesced streaming reads, so it is doubled; WRITE_SIZE is exact."""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_kinds import kind_of, sources_sha256
tag, fcsv, wcsv = sys.argv[1:4]
def per_kernel(path, name):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in d.items()}
f, w = per_kernel(fcsv, "FETCH_SIZE"), per_kernel(wcsv, "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w)):
    if not k.startswith("void pd::") and not k.startswith("pd::"):
        continue
    fr, n = f.get(k, (0.0, 0)); wr, _ = w.get(k, (0.0, 0))
    out[k] = {"launches_sampled": n, "fetch_bytes_per_launch": 2.0 * fr * 1024, "write_bytes_per_launch": wr * 1024,
              "hbm_bytes_per_launch": 2.0 * fr * 1024 + wr * 1024}
by_kind = collections.defaultdict(lambda: [0.0, 0.0, 0])       # launch-weighted mean per plan kind
for k, v in out.items():
    kd = kind_of(k)
    if kd:
        by_kind[kd][0] += v["fetch_bytes_per_launch"] * v["launches_sampled"]; by_kind[kd][1] += v["write_bytes_per_launch"] * v["launches_sampled"]
        by_kind[kd][2] += v["launches_sampled"]
by_kind = {kd: {"fetch_bytes_per_launch": f / n, "write_bytes_per_launch": w_ / n, "hbm_bytes_per_launch": (f + w_) / n, "launches_sampled": n}
           for kd, (f, w_, n) in by_kind.items() if n}
json.dump({"sources_sha256": sources_sha256(), "by_kind": by_kind, "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace only) over "
                   "`python3 bench.py --steps 1 --warmup 0 --batch 32 --inference-steps 2 --no-cpu-baseline --no-roofline`; "
                   "FETCH_SIZE x2 (gfx950 correction), x1024 B; averages per launch over all launches of each kernel",
           "kernels": out}, open(f"profiles/{tag}_hbm_traffic.json", "w"), indent=1)
for k, v in out.items():
    print(f"{k[:70]:70s} n={v['launches_sampled']:5d} fetch {v['fetch_bytes_per_launch']/1e6:9.1f} MB  write {v['write_bytes_per_launch']/1e6:9.1f} MB")
