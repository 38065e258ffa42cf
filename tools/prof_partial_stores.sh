#!/bin/bash
# What 32-byte stores cost the write path (TCC counters, own passes): 4:4:4 frames in MCU pairs on 64-byte boundaries
# (1920 across, DRI = 4), pairs across them in every second MCU row (1912 across), single MCUs (DRI = 5).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_partial_stores
rm -rf $OUT && mkdir -p $OUT
for cfg in "1920 1088 4" "1912 1088 4" "1920 1088 5"; do
  tag=$(echo $cfg | tr ' ' '_')
  for pmc in FETCH_SIZE WRITE_SIZE "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    p=$(echo $pmc | tr ' ' '+')
    rocprofv3 --pmc $pmc --output-format csv -d $OUT/${tag}_$p -- python3 tools/partial_store_probe.py $cfg > $OUT/${tag}_$p.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, os, collections
out = "gpurun_out/prof_partial_stores"
for d in sorted(glob.glob(out + "/*/")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "decode_fused" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(os.path.basename(d.rstrip("/")), k, c, "launches", len(v), "mean", sum(v) / len(v))
PY
