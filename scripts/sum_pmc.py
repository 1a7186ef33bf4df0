import collections, csv, sys
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    per[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if "pd::" not in k: continue
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc <= 0: continue
    print(k[:75])
    print("   " + "  ".join(f"{n.replace('SQ_','')}={v/wc*100:.1f}%" for n, v in sorted(c.items()) if n != "SQ_WAVE_CYCLES"))
