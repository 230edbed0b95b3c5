#!/bin/bash
# Effective shader clock of several builds of one kernel: ONE rocprofv3 pass (--kernel-trace --pmc GRBM_GUI_ACTIVE) over the interleaved A/B of
# scripts/ab_fbank.py; clock = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / dispatch duration, median over each build's timed dispatches.
# usage (GPU box, repo root): scripts/clock_probe.sh <vgg|logmel|har> <kernel-name-substring> name=lib.so [name=lib.so ...]   -> gpurun_out/prof_clock/
set -u
CFG=$1; NEEDLE=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_clock
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
ARGS=(); NAMES=()
for b in "$@"; do n=${b%%=*}; p=${b#*=}; case $p in /*) ;; *) p=$ROOT/$p;; esac; ARGS+=("$n=$p"); NAMES+=("$n"); done
ROUNDS=4; LAUNCHES=20
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/p -- python3 $ROOT/scripts/ab_fbank.py --config $CFG --rounds $ROUNDS --launches $LAUNCHES "${ARGS[@]}" > $OUT/ab_$CFG.log 2>&1
echo rc=$?
python3 - "$OUT" "$NEEDLE" $ROUNDS $LAUNCHES "${NAMES[@]}" <<'PY'
import csv, glob, statistics, sys
out, needle, rounds, launches, names = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5:]
ct = glob.glob(out + "/p/*/*_counter_collection.csv")[0]
kt = glob.glob(out + "/p/*/*_kernel_trace.csv")[0]
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
rows = sorted((int(r["Dispatch_Id"]), dur[r["Dispatch_Id"]], float(r["Counter_Value"]) / 8.0 / dur[r["Dispatch_Id"]])
              for r in csv.DictReader(open(ct)) if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and needle in r["Kernel_Name"] and dur.get(r["Dispatch_Id"], 0) > 0)
per = 3 + launches                    # ab_fbank.py: 3 untimed + `launches` timed launches per build and round; odd rounds in reverse order
tail = rows[-rounds * len(names) * per:]
agg = {n: [] for n in names}
for r in range(rounds):
    order = names if r % 2 == 0 else names[::-1]
    for k, n in enumerate(order):
        agg[n] += tail[(r * len(names) + k) * per:(r * len(names) + k + 1) * per][3:]
print(f"{'build':<12} {'median ms':>10} {'GHz median':>11} {'min':>6} {'max':>6} {'M cycles':>9}")
for n in names:
    d = statistics.median(b for _, b, _ in agg[n]) / 1e6
    g = [c for _, _, c in agg[n]]
    print(f"{n:<12} {d:>10.4f} {statistics.median(g):>11.3f} {min(g):>6.3f} {max(g):>6.3f} {d * statistics.median(g):>9.3f}")
PY
rm -rf $OUT/p
