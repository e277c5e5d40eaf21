#!/bin/bash
# TEST INFRASTRUCTURE / build-time tool (this container only: needs /root/reference and oracle/_ref/glchain).
# Regenerates the instruction lists of crt-royale's last pass (geometry-aa-last-pass.glsl) from the NIR Mesa llvmpipe
# compiles for it: oracle/gen/royale_last_{vs,fs}.inc for the oracle and the same text under
# retrocapture_amd/csrc/kernels/gen/ for the HIP kernel's general form (pass_royale_last_general.hip).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
REF="${REF:-/root/reference}"
GLSL="$REF/shaders/shaders_glsl"
T="$(mktemp -d)"
trap 'rm -rf "$T"' EXIT
python3 - "$T" <<'PY'
import sys, numpy as np
np.random.default_rng(1).integers(0, 256, (24, 32, 3), dtype=np.uint8).tofile(sys.argv[1] + "/in.rgb")
PY
# a one-pass preset of just that shader: the compiled code does not depend on the rest of the chain
printf 'shaders = 1\nshader0 = %s/crt/shaders/crt-royale/src/crt-royale-geometry-aa-last-pass.glsl\nfilter_linear0 = true\n' "$GLSL" > "$T/last.glslp"
run() {  # $1 = env assignment, $2 = output listing
  ( cd "$REF" && env "$1" MESA_SHADER_CACHE_DISABLE=true RETROCAPTURE_LOG_LEVEL=error "$ROOT/oracle/_ref/glchain" --preset "$T/last.glslp" --input "$T/in.rgb" \
      --w 32 --h 24 --vw 64 --vh 48 --frames 1 --out "$T" ) > /dev/null 2> "$2"
}
run LP_DEBUG=fs "$T/fs.txt"
run GALLIVM_DEBUG=tgsi "$T/vs.txt"
for d in "$ROOT/oracle/gen" "$ROOT/retrocapture_amd/csrc/kernels/gen"; do
  mkdir -p "$d"
  python3 "$HERE/nir2c.py" "$T/fs.txt" --stage fragment --name royale_last_fs > "$d/royale_last_fs.inc"
  python3 "$HERE/nir2c.py" "$T/vs.txt" --stage vertex --name royale_last_vs > "$d/royale_last_vs.inc"
done
echo "generated: $(wc -l < "$ROOT/oracle/gen/royale_last_fs.inc") + $(wc -l < "$ROOT/oracle/gen/royale_last_vs.inc") lines"
