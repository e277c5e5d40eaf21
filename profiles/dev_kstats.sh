#!/bin/bash
# Development: per-kernel durations of the default bench workload on one lane (run through gpurun from the repo root).
# Usage: profiles/dev_kstats.sh [extra bench.py arguments]; prints the ten longest kernels.
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out/kstats"
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o k -- python3 "$REPO/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --lanes 1 "$@" > "$OUT/bench.json" 2> "$OUT/err.txt"
python3 "$REPO/profiles/summarize.py" "$OUT" "$OUT" > /dev/null
head -14 "$OUT/kernel_stats.csv"
