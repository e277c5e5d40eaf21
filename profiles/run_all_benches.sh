#!/bin/bash
# Every bench workload once (run through gpurun from the repo root); lines land in gpurun_out/bench_<workload>.json.
# Usage: profiles/run_all_benches.sh [tag]; copy the files you want judged to profiles/<round>_bench_<workload>.json.
set -e
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$REPO"
mkdir -p gpurun_out
python3 bench.py > gpurun_out/bench_crt-royale.json 2> gpurun_out/bench_crt-royale.err
echo "crt-royale done"
for W in crt-royale-fake-bloom crt-hyllian-glow crt-easymode zfast-crt crt-pi crt-geom ntsc xbr-lv3 xbr-lv2 scalefx scanline lcd-grid-v2 crt-lottes tvout; do
  python3 bench.py --workload $W --cpu-budget 6 > gpurun_out/bench_$W.json 2> gpurun_out/bench_$W.err   # (every line carries its cpu_baseline)
  echo "$W done"
done
python3 bench.py --workload ntsc --fp16-targets --cpu-budget 6 > gpurun_out/bench_ntsc_fp16.json 2> gpurun_out/bench_ntsc_fp16.err
# crt-royale with the last pass in its tex2Daa / ray-cast form (curved geometry)
python3 bench.py --param geom_mode_runtime=1 --cpu-budget 6 > gpurun_out/bench_crt-royale_curved.json 2> gpurun_out/bench_crt-royale_curved.err
python3 bench.py --io > gpurun_out/bench_io.json 2> gpurun_out/bench_io.err
echo "io done"
