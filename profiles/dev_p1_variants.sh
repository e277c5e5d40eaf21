#!/bin/bash
# Development: pass 1's time per frame for a list of -D variants of pass_royale_scan.hip (run through gpurun).
#   profiles/dev_p1_variants.sh "" "-DRC_SCAN_STRIP_ROWS=32" "-DRC_SCAN_ROWS_VECTOR" ...
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
for V in "$@"; do
  bash profiles/dev_variant.sh pass_royale_scan="$V" -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --modes default --lanes ${LANES:-1} > gpurun_out/p1v.json 2> gpurun_out/p1v.err || { echo "variant '$V' failed"; tail -5 gpurun_out/p1v.err; continue; }
  python3 - "$V" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/p1v.json").read().strip().splitlines()[-1])
pp = d["config"].get("per_pass_ms_per_frame") or d.get("per_pass_ms_per_frame")
print("variant [%s]: %.0f frames/s, pass 1 %.2f us" % (sys.argv[1], d["value"], pp[1] * 1e3 if pp else -1), flush=True)
PY
done
