#!/bin/bash
# Sweep the STFT-32 kernel's tuning overrides on the HAR configuration: rows per group and mover waves of the
# wave-specialised form (MMFE_STFT_MOVERS=-1 selects the phase-serial form).
set -o pipefail
for cfg in "0 0" "0 -1" "16 2" "16 3" "16 4" "16 5" "16 6" "12 3" "12 4" "12 6" "8 4" "8 6" "8 8"; do
  set -- $cfg
  echo -n "R=$1 movers=$2 "
  MMFE_STFT_ROWS=$1 MMFE_STFT_MOVERS=$2 timeout -k 10 60 python scripts/bench_configs.py har 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms'], d['min_ms'], d['frac_of_8TBs'])" || exit 1
done
