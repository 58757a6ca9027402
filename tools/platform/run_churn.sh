#!/bin/bash
# N concurrent alloc_churn processes: bash tools/platform/run_churn.sh SECONDS N [MODES]
# MODES: one digit per process, cycled (0 malloc/free every round, 1 one buffer reused, 2 fresh buffer touched first)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
SEC=${1:-60}; N=${2:-4}; MODES=${3:-0}
# hard caps (tests/fuzz/_args.py has the story): at most 4 processes, at most an hour
if ! [[ "$SEC" =~ ^[0-9]+$ ]] || [ "$SEC" -gt 3600 ]; then echo "SECONDS must be an integer <= 3600, got '$SEC'" >&2; exit 2; fi
if ! [[ "$N" =~ ^[0-9]+$ ]] || [ "$N" -lt 1 ] || [ "$N" -gt 4 ]; then echo "N (processes) must be 1..4, got '$N'" >&2; exit 2; fi
OUT=$ROOT/gpurun_out/churn_$N
mkdir -p $OUT
pids=""
for i in $(seq 0 $((N-1))); do
  m=${MODES:$((i % ${#MODES})):1}
  timeout $((SEC + 60)) $ROOT/tools/platform/alloc_churn $SEC $((700 + i)) $m > $OUT/c$i.log 2>&1 &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
grep -h "WRONG" $OUT/c*.log | head -30
tail -q -n 1 $OUT/c*.log
