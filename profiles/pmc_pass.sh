#!/bin/bash
# One rocprofv3 --pmc pass per counter group over a short script (run through gpurun from the repo root):
#   profiles/pmc_pass.sh <tag> <script.py> "<CTRS group 1>" "<CTRS group 2>" ...
# Raw output under gpurun_out/prof_<tag>/, per-kernel averages summarised by profiles/summarize.py.
set -e
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
TAG="$1"; SCRIPT="$2"; shift 2
OUT="$REPO/gpurun_out/prof_$TAG"
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "$@"; do
  i=$((i + 1))
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT/pmc$i" -o run -- python3 "$REPO/$SCRIPT" > "$OUT/run$i.out" 2> "$OUT/run$i.err"
  echo "pmc pass $i ($CTRS) done"
done
mkdir -p "$OUT/stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$REPO/$SCRIPT" > "$OUT/stats.out" 2> "$OUT/stats.err"
python3 "$REPO/profiles/summarize.py" "$OUT" "$OUT"
