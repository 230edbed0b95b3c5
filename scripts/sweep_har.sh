#!/bin/bash
# Sweep the STFT-32 kernel's rows-per-group / block-size overrides on the HAR configuration.
set -o pipefail
for cfg in "0 0" "32 1024" "28 1024" "24 1024" "16 512" "16 832" "16 1024" "12 640" "8 448" "8 256" "36 1024"; do
  set -- $cfg
  echo -n "R=$1 T=$2 "
  MMFE_STFT_ROWS=$1 MMFE_STFT_THREADS=$2 timeout -k 10 60 python scripts/bench_configs.py har 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms'], d['min_ms'], d['frac_of_8TBs'])" || exit 1
done
