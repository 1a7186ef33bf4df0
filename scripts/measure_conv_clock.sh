#!/bin/bash
# The clock the chip holds under the 3x3 convolution (MI355X_MICROARCH.md "DVFS give-back": effective clock = GRBM_GUI_ACTIVE / 8 XCDs /
# kernel duration, counter pass with --kernel-trace only).  Batches are chosen so that a launch lasts >= 0.5 ms (the quotient reads high
# on short dispatches).  Writes gpurun_out/r3_conv_clock.json.     bash scripts/measure_conv_clock.sh
cd "$(dirname "$0")/.."
ROOT=$(pwd)
export TMPDIR=/tmp
out=$ROOT/gpurun_out; mkdir -p $out
i=0
for cfg in "--hw 256 --cin 64 --cout 64 --gn 1 --batch 96 --iters 2500" "--hw 256 --cin 64 --cout 64 --gn 0 --batch 96 --iters 2500" \
           "--hw 64 --cin 256 --cout 256 --gn 1 --batch 256 --iters 2500" "--hw 256 --cin 128 --c1 64 --cout 64 --gn 1 --batch 64 --iters 2000"; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/cclk_$i -- python3 $ROOT/scripts/bench_conv.py $cfg > $out/r3_conv_clock_run_$i.log 2>&1)
  i=$((i+1))
done
python3 - "$out" <<'PY'
import csv, glob, json, statistics, sys
out = sys.argv[1]
names = ["64->64 @256^2 GN+SiLU B96", "64->64 @256^2 plain B96", "256->256 @64^2 GN+SiLU B256", "128+64->64 @256^2 GN+SiLU B64"]
res = {"method": "rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE: clock = counter / 8 XCDs / (End - Start), second half of the loop"}
for i, nm in enumerate(names):
    cc = glob.glob(f"/tmp/cclk_{i}/*/*_counter_collection.csv")
    if not cc: continue
    rows = [r for r in csv.DictReader(open(cc[0])) if "conv_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
    clk, durs = [], []
    for r in rows[len(rows) // 2:]:
        dur = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
        if dur > 0: clk.append(float(r["Counter_Value"]) / 8.0 / dur / 1e9); durs.append(dur)
    if clk:
        res[nm] = {"launches": len(clk), "ms_median": round(statistics.median(durs) * 1e3, 4), "ghz_median": round(statistics.median(clk), 3),
                   "ghz_p10": round(sorted(clk)[len(clk) // 10], 3), "ghz_p90": round(sorted(clk)[-len(clk) // 10 - 1], 3),
                   "bench_line": open(f"{out}/r3_conv_clock_run_{i}.log").read().strip().splitlines()[-1:]}
json.dump(res, open(f"{out}/r3_conv_clock.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
