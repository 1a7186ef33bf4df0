#!/usr/bin/env python3
"""Steady-state per-kernel summary from a rocprofv3 --kernel-trace CSV (round 5).

    python scripts/steady_stats.py <kernel_trace.csv> <batches> <discard> [out.csv]

`rocprofv3 --stats` averages over the whole process, including the first batches on a cold chip (5 % faster than the steady state
of this power-coupled workload).  This script cuts the trace at the start of batch `discard` + 1 of `batches` equal batches -- the
launches of the most frequent long kernel are split into `batches` equal runs by order -- and writes the same columns as
rocprofv3's `*_kernel_stats.csv` for what remains (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev)."""
import collections
import csv
import math
import sys


def main():
    path, batches, discard = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    out = sys.argv[4] if len(sys.argv) > 4 else None
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    total_by = collections.Counter()
    for s, e, n in rows:
        total_by[n] += e - s
    anchor = max(total_by, key=total_by.get)                 # the dominant kernel: its launches are evenly spread over the batches
    starts = [s for s, _, n in rows if n == anchor]
    per = len(starts) // batches
    cut = starts[per * discard] if discard > 0 else rows[0][0]
    # the launches just before the first anchor launch of batch discard + 1 belong to that batch too; the anchor sits ~10 % into a
    # forward, so the cut is within one forward of the batch boundary (1 / (100 x batches) of the trace)
    keep = [(s, e, n) for s, e, n in rows if s >= cut]
    acc = collections.defaultdict(list)
    for s, e, n in keep:
        acc[n].append(e - s)
    tot = sum(sum(v) for v in acc.values())
    span = keep[-1][1] - keep[0][0]
    table = []
    for n, v in acc.items():
        m = sum(v) / len(v)
        sd = math.sqrt(sum((x - m) ** 2 for x in v) / len(v))
        table.append((n, len(v), sum(v), m, 100.0 * sum(v) / tot, min(v), max(v), sd))
    table.sort(key=lambda t: -t[2])
    w = csv.writer(open(out, "w", newline="") if out else sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for t in table:
        w.writerow([t[0], t[1], t[2], f"{t[3]:.1f}", f"{t[4]:.2f}", t[5], t[6], f"{t[7]:.1f}"])
    print(f"steady state: batches {discard + 1}..{batches} of {batches} ({len(keep)} of {len(rows)} launches), anchor {anchor[:60]} "
          f"({per} launches per batch), kernel time {tot / 1e6:.1f} ms over a span of {span / 1e6:.1f} ms "
          f"({100.0 * tot / span:.1f} % busy)", file=sys.stderr)


if __name__ == "__main__":
    main()
