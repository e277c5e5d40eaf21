#!/bin/bash
# TEST INFRASTRUCTURE / build-time tool (this container only: needs /root/reference and oracle/_ref/glchain).
# Regenerates the instruction lists (oracle/glrun/nir2c.py) from the NIR Mesa llvmpipe compiles for a shader of the reference:
#   crt-royale's last pass (geometry-aa-last-pass.glsl, both stages)  -> royale_last_{vs,fs}.inc
#   handheld/shaders/lcd-cgwg/lcd-grid-v2.glsl (fragment stage)        -> lcd_grid_v2_fs.inc
#   handheld/shaders/lcd-cgwg/lcd-grid.glsl (fragment stage)           -> lcd_grid_fs.inc
#   crt/shaders/tvout-tweaks.glsl (fragment stage)                     -> tvout_tweaks_fs.inc
#   misc/image-adjustment.glsl (both stages)                           -> image_adjustment_{vs,fs}.inc
#   windowed/shaders/jinc2-sharper.glsl (fragment stage)               -> jinc2_sharper_fs.inc
#   crt/shaders/crt-lottes.glsl, crt/shaders/fakelottes.glsl (fragment) -> crt_lottes_fs.inc, fakelottes_fs.inc
#   stereoscopic-3d/shaders/side-by-side-simple.glsl (both stages)     -> side_by_side_{vs,fs}.inc
#   handheld/shaders/sameboy-lcd.glsl (fragment stage)                 -> sameboy_lcd_fs.inc
#   crt/shaders/crt-consumer.glsl (fragment stage)                     -> crt_consumer_fs.inc
#   anti-aliasing/shaders/reverse-aa.glsl (fragment stage)             -> reverse_aa_fs.inc
#   anti-aliasing/shaders/advanced-aa.glsl (both stages)               -> advanced_aa_{vs,fs}.inc
# written to oracle/gen/ for the oracle and, the same text, to retrocapture_amd/csrc/kernels/gen/ for the HIP kernels.
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
listing() {  # $1 = shader below shaders_glsl, $2 = env assignment, $3 = output listing
  # a one-pass preset of just that shader: the compiled code does not depend on the rest of the chain
  printf 'shaders = 1\nshader0 = %s/%s\nfilter_linear0 = true\n' "$GLSL" "$1" > "$T/one.glslp"
  ( cd "$REF" && env "$2" MESA_SHADER_CACHE_DISABLE=true RETROCAPTURE_LOG_LEVEL=error "$ROOT/oracle/_ref/glchain" --preset "$T/one.glslp" --input "$T/in.rgb" \
      --w 32 --h 24 --vw 64 --vh 48 --frames 1 --out "$T" ) > /dev/null 2> "$3"
}
emit() {  # $1 = listing, $2 = stage, $3 = name
  for d in "$ROOT/oracle/gen" "$ROOT/retrocapture_amd/csrc/kernels/gen"; do
    mkdir -p "$d"
    python3 "$HERE/nir2c.py" "$1" --stage "$2" --name "$3" > "$d/$3.inc"
  done
}
R=crt/shaders/crt-royale/src/crt-royale-geometry-aa-last-pass.glsl
listing "$R" LP_DEBUG=fs "$T/fs.txt" && emit "$T/fs.txt" fragment royale_last_fs
listing "$R" GALLIVM_DEBUG=tgsi "$T/vs.txt" && emit "$T/vs.txt" vertex royale_last_vs
listing handheld/shaders/lcd-cgwg/lcd-grid-v2.glsl LP_DEBUG=fs "$T/lcd.txt" && emit "$T/lcd.txt" fragment lcd_grid_v2_fs
listing handheld/shaders/lcd-cgwg/lcd-grid.glsl LP_DEBUG=fs "$T/lcd1.txt" && emit "$T/lcd1.txt" fragment lcd_grid_fs
listing crt/shaders/tvout-tweaks.glsl LP_DEBUG=fs "$T/tv.txt" && emit "$T/tv.txt" fragment tvout_tweaks_fs
listing misc/image-adjustment.glsl LP_DEBUG=fs "$T/ia.txt" && emit "$T/ia.txt" fragment image_adjustment_fs
listing misc/image-adjustment.glsl GALLIVM_DEBUG=tgsi "$T/iav.txt" && emit "$T/iav.txt" vertex image_adjustment_vs
listing windowed/shaders/jinc2-sharper.glsl LP_DEBUG=fs "$T/j2.txt" && emit "$T/j2.txt" fragment jinc2_sharper_fs
listing crt/shaders/crt-lottes.glsl LP_DEBUG=fs "$T/lo.txt" && emit "$T/lo.txt" fragment crt_lottes_fs
listing crt/shaders/fakelottes.glsl LP_DEBUG=fs "$T/fl.txt" && emit "$T/fl.txt" fragment fakelottes_fs
S=stereoscopic-3d/shaders/side-by-side-simple.glsl
listing "$S" LP_DEBUG=fs "$T/sbs.txt" && emit "$T/sbs.txt" fragment side_by_side_fs
listing "$S" GALLIVM_DEBUG=tgsi "$T/sbsv.txt" && emit "$T/sbsv.txt" vertex side_by_side_vs
listing handheld/shaders/sameboy-lcd.glsl LP_DEBUG=fs "$T/sl.txt" && emit "$T/sl.txt" fragment sameboy_lcd_fs
listing crt/shaders/crt-consumer.glsl LP_DEBUG=fs "$T/cc.txt" && emit "$T/cc.txt" fragment crt_consumer_fs
listing anti-aliasing/shaders/reverse-aa.glsl LP_DEBUG=fs "$T/ra.txt" && emit "$T/ra.txt" fragment reverse_aa_fs
A=anti-aliasing/shaders/advanced-aa.glsl
listing "$A" LP_DEBUG=fs "$T/aa.txt" && emit "$T/aa.txt" fragment advanced_aa_fs
listing "$A" GALLIVM_DEBUG=tgsi "$T/aav.txt" && emit "$T/aav.txt" vertex advanced_aa_vs
wc -l "$ROOT"/oracle/gen/*.inc
