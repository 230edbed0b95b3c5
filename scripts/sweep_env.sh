#!/bin/bash
# usage: scripts/sweep_env.sh VAR "v1 v2 ..." <bench_configs target>
VAR=$1; VALS=$2; TARGET=$3
for v in $VALS; do
  echo -n "$VAR=$v  "
  env $VAR=$v timeout -k 10 300 python scripts/bench_configs.py $TARGET 2>&1 | grep config | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms"],4), "ms  frac", round(d["frac_of_8TBs"],4))'
done
