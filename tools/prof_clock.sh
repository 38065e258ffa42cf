#!/bin/bash
# Effective shader clock of the decode kernel under two library builds:
# GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md, DVFS give-back).
#   tools/prof_clock.sh libA.so libB.so ...
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  name=$(basename $lib .so)
  export COMPEG_LIB="$GRAFT_REPO_ROOT/$lib"
  rm -rf gpurun_out/clk_$name
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/clk_$name -- python3 bench.py --batch 128 --steps 4 --warmup 1 --cpu-seconds 0 --no-verify > gpurun_out/clk_$name.log 2>&1
  python3 - "$name" <<'PY'
import csv, glob, sys, os
name = sys.argv[1]
f = max(glob.glob(f"gpurun_out/clk_{name}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if "decode_fused" in r["Kernel_Name"] and int(r["Grid_Size"]) > 1000000]
by = {}
for r in rows:
    d = by.setdefault(r["Dispatch_Id"], {"t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
for k, d in list(by.items())[-3:]:
    cyc = d["GRBM_GUI_ACTIVE"] / 8
    print(f"{name}: dispatch {k}: {d['t']/1e3:.1f} us, {cyc/1e6:.3f} Mcycles, clock {cyc/d['t']:.3f} GHz, wave-quadcycles {d.get('SQ_WAVE_CYCLES',0)/1e6:.0f}M")
PY
done
