#!/bin/bash
# Collects the rocprofv3 evidence for the default bench workload on the GPU box (run through gpurun from the repo
# root): per-kernel time statistics, then hardware counters in separate --pmc passes (never combined with other trace
# domains).  Raw output goes to gpurun_out/prof/, the summaries judged live in profiles/ (profiles/summarize.py).
set -e
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out/prof"
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o royale -- python3 "$REPO/bench.py" --steps 10 --no-cpu-baseline --modes default --lanes 1 > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
echo "stats (default mode, one lane: a kernel's own duration) done"
# the same as the engine runs by default: two lanes, the kernels of the two halves of a batch overlapping (their durations stretch)
mkdir -p "$OUT/lanes2"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/lanes2/stats" -o royale -- python3 "$REPO/bench.py" --steps 10 --no-cpu-baseline --modes default --lanes 2 > "$OUT/bench_stats_lanes2.json" 2> "$OUT/stats_lanes2.err"
echo "stats (default mode, two lanes) done"
# the same with the phosphor mask rendered, in its own summary (kernel times differ a lot between the two modes)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/mask/stats" -o royale -- python3 "$REPO/bench.py" --steps 10 --no-cpu-baseline --modes mask --lanes 1 > "$OUT/bench_stats_mask.json" 2> "$OUT/stats_mask.err"
echo "stats (mask rendered) done"
i=0
for CTRS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1))
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT/pmc$i" -o royale -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --modes default --lanes 1 > "$OUT/bench_pmc$i.json" 2> "$OUT/pmc$i.err"
  echo "pmc pass $i ($CTRS) done"
done
python3 "$REPO/profiles/summarize.py" "$OUT" "$OUT"
python3 "$REPO/profiles/summarize.py" "$OUT/mask" "$OUT/mask"
python3 "$REPO/profiles/summarize.py" "$OUT/lanes2" "$OUT/lanes2"
