#!/bin/bash
# effective shader clock of each build: GRBM_GUI_ACTIVE (sum over 8 XCDs) / 8 / kernel duration, from one rocprofv3 pass with kernel trace
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_clock
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
L=$ROOT/build/libmmfe_fb
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/p -- python3 $ROOT/scripts/ab_fbank.py --rounds 4 --launches 20 base=$ROOT/multi_modal_amd/libmmfe.so abl=${L}_abl.so nold=${L}_nold.so nost=${L}_nost.so > $OUT/ab.log 2>&1
echo rc=$?
python3 - <<PY
import csv, glob, collections
ct = glob.glob("$OUT/p/*/*_counter_collection.csv")[0]
kt = glob.glob("$OUT/p/*/*_kernel_trace.csv")[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
agg = collections.defaultdict(list)
for r in csv.DictReader(open(ct)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or "fbank512" not in r["Kernel_Name"]: continue
    d, name = dur.get(r["Dispatch_Id"], (0, ""))
    if d <= 0: continue
    agg[int(r.get("Grid_Size", 0)), r["Kernel_Name"][-40:]].append((float(r["Counter_Value"]) / 8.0 / d, d))
# the four builds run the same kernel name: separate them by duration clusters instead -> print per-dispatch table summary by rounded duration
rows = []
for r in csv.DictReader(open(ct)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or "fbank512" not in r["Kernel_Name"]: continue
    d, _ = dur.get(r["Dispatch_Id"], (0, ""))
    if d > 0: rows.append((int(r["Dispatch_Id"]), d, float(r["Counter_Value"]) / 8.0 / d))
rows.sort()
import statistics
# dispatch order: ab_fbank launches builds in blocks of 23 (3 warm + 20 timed) per round after the warm-up phase; print every dispatch compactly
with open("$OUT/clock_rows.txt", "w") as fh:
    for i, d, ghz in rows: fh.write(f"{i} {d} {ghz:.3f}\n")
print("dispatches", len(rows), "median GHz", statistics.median(g for _, _, g in rows))
PY
rm -rf $OUT/p
