#!/bin/bash
# Evidence for the "the d = 8 attention runs at ~2.0 of 2.4 GHz" claim: the clock the chip holds
# while pd_attn_d8 (DMA-staged kernel, B = 32, 32 heads, N = 4096) runs back to back for ~2 s, two ways --
#   (1) GRBM_GUI_ACTIVE / 8 XCDs / kernel duration from rocprofv3 (counter pass with --kernel-trace only; MI355X_MICROARCH.md "DVFS
#       give-back": within 3 % of the in-kernel clock on dispatches of >= 10 ms, reads high on short ones: N = 16384 launches of 26 ms
#       are measured beside the 1.6 ms ones),
#   (2) rocm-smi sclk samples taken by a background loop while the same loop runs un-profiled (what the driver's smi.*.json holds).
# Writes gpurun_out/r3_clock.json.     bash scripts/measure_clock.sh
cd "$(dirname "$0")/.."
ROOT=$(pwd)
export TMPDIR=/tmp
out=$ROOT/gpurun_out; mkdir -p $out
for cfg in "4096 32 1300" "16384 8 80"; do
  set -- $cfg; N=$1; B=$2; IT=$3
  (cd /tmp && rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/clk_$N -- python3 $ROOT/scripts/bench_attn.py --kmax 1 --n $N --batch $B --iters $IT > $out/r3_clock_run_$N.log 2>&1)
done
# (2) un-profiled loop + rocm-smi sampler
( for i in $(seq 1 40); do rocm-smi --showclocks --json 2>/dev/null | tr -d '\n'; echo; sleep 0.1; done > $out/r3_clock_smi.jsonl ) &
SMI=$!
python3 scripts/bench_attn.py --kmax 1 --iters 2500 > $out/r3_clock_bench.log 2>&1
wait $SMI
python3 - "$out" <<'PY'
import csv, glob, json, re, statistics, sys
out = sys.argv[1]
res = {"method": "rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE: clock = counter / 8 XCDs / (End - Start); rocm-smi sclk samples during an un-profiled loop"}
for N in (4096, 16384):
    cc = glob.glob(f"/tmp/clk_{N}/*/*_counter_collection.csv")
    if not cc: continue
    rows = [r for r in csv.DictReader(open(cc[0])) if "attn" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
    clk = []
    for r in rows[len(rows) // 2:]:          # the second half of the loop: the chip is warm
        dur = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
        if dur > 0: clk.append(float(r["Counter_Value"]) / 8.0 / dur / 1e9)
    if clk:
        res[f"N{N}"] = {"launches": len(clk), "kernel": rows[0]["Kernel_Name"][:60], "ghz_median": round(statistics.median(clk), 3),
                        "ghz_p10": round(sorted(clk)[len(clk) // 10], 3), "ghz_p90": round(sorted(clk)[-len(clk) // 10 - 1], 3)}
sclk = []
for line in open(f"{out}/r3_clock_smi.jsonl"):
    for m in re.finditer(r'"sclk[^"]*"\s*:\s*"\(?(\d+)\s*Mhz', line, flags=re.I):
        sclk.append(int(m.group(1)))
if sclk:
    res["rocm_smi_sclk_mhz"] = {"samples": len(sclk), "median": statistics.median(sclk), "min": min(sclk), "max": max(sclk)}
res["bench_line"] = open(f"{out}/r3_clock_bench.log").read().strip().splitlines()[-1:]
json.dump(res, open(f"{out}/r3_clock.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
