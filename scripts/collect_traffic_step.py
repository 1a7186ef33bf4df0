#!/usr/bin/env python3
"""Round 6: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over exactly ONE step of a side workload (train | sd_train | sd_img2img) ->
profiles/<tag>_hbm_traffic_<workload>.json: total HBM bytes per kernel GROUP over that step (scripts/kernel_kinds.py GROUP_RULES), keyed by
the hash of the kernel sources.  bench.py divides a group's bytes by the number of plan ops it stands for (`roofline.traffic`, per launch
like `achieved`) and quotes measured / algorithmic ratios per group.
Usage (GPU box):  python scripts/collect_traffic_step.py <tag> <workload> <fetch_csv> <write_csv> "<profiled command>" [plan passes it ran]
Corrections per MI355X_MICROARCH.md (HBM section): counter unit 1024 B; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced
streaming reads, so it is doubled; WRITE_SIZE is exact."""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_kinds import group_of, sources_sha256
tag, workload, fcsv, wcsv, cmd = sys.argv[1:6]
passes = float(sys.argv[6]) if len(sys.argv) > 6 else 1.0        # plan passes the profiled command ran (the totals are divided by it)


def totals(path, name):
    d = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            e = d[r["Kernel_Name"]]
            e[0] += float(r["Counter_Value"]); e[1] += 1
    return d


f, w = totals(fcsv, "FETCH_SIZE"), totals(wcsv, "WRITE_SIZE")
groups = collections.defaultdict(lambda: {"fetch_bytes": 0.0, "write_bytes": 0.0, "kernel_launches": 0, "kernels": []})
other = 0.0
for k in sorted(set(f) | set(w)):
    fb, wb = 2.0 * f.get(k, [0.0, 0])[0] * 1024 / passes, w.get(k, [0.0, 0])[0] * 1024 / passes
    g = group_of(k)
    if g is None:
        other += fb + wb
        continue
    e = groups[g]
    e["fetch_bytes"] += fb; e["write_bytes"] += wb; e["kernel_launches"] += max(f.get(k, [0, 0])[1], w.get(k, [0, 0])[1])
    e["kernels"].append(k[:100])
for e in groups.values():
    e["hbm_bytes"] = e["fetch_bytes"] + e["write_bytes"]
out = {"sources_sha256": sources_sha256(), "workload": workload, "profiled_command": cmd, "plan_passes_profiled": passes, "groups": groups, "ungrouped_hbm_bytes": other,
       "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over ONE step; FETCH_SIZE x2 (gfx950), x1024 B; "
               "totals over all launches of the group's kernels in that step"}
json.dump(out, open(f"profiles/{tag}_hbm_traffic_{workload}.json", "w"), indent=1)
for g, e in sorted(groups.items(), key=lambda kv: -kv[1]["hbm_bytes"]):
    print(f"{g:16s} launches {e['kernel_launches']:6d}  fetch {e['fetch_bytes']/1e9:8.2f} GB  write {e['write_bytes']/1e9:8.2f} GB")
