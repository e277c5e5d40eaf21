#!/bin/bash
# Development: frames/s of one bench workload for -D variants of one kernel file (run through gpurun).
#   profiles/dev_workload_variants.sh crt-pi pass_crt_pi "" "-DRC_PI_WAVES=5" ...
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
WL="$1"; F="$2"; shift; shift
mkdir -p gpurun_out
for V in "$@"; do
  bash profiles/dev_variant.sh $F="$V" -- python3 bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/wv.json 2> gpurun_out/wv.err || { echo "variant '$V' failed"; tail -5 gpurun_out/wv.err; continue; }
  python3 - "$V" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/wv.json").read().strip().splitlines()[-1])
print("variant [%s]: %.0f frames/s, roofline frac %.3f" % (sys.argv[1], d["value"], d["roofline"]["frac"]), flush=True)
PY
done
