#!/bin/bash
# Development: per-pass times with the mask rendered for -D variants of one kernel file (run through gpurun).
#   profiles/dev_mask_variants.sh pass_royale "" "-DRC_SH_WAVES=10" ...
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
F="$1"; shift
mkdir -p gpurun_out
for V in "$@"; do
  bash profiles/dev_variant.sh $F="$V" -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --modes mask --lanes ${LANES:-1} > gpurun_out/mv.json 2> gpurun_out/mv.err || { echo "variant '$V' failed"; tail -5 gpurun_out/mv.err; continue; }
  python3 - "$V" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/mv.json").read().strip().splitlines()[-1])
m = d.get("mask_rendered", d)
pp = m.get("per_pass_ms_per_frame")
print("variant [%s]: mask %.0f frames/s, per pass us %s" % (sys.argv[1], m["value"], [round(v * 1e3, 2) for v in pp]), flush=True)
PY
done
