/* TEST INFRASTRUCTURE - see rc_oracle.h.
 *
 * How a vertex-shader output reaches pixel (x, y): the reference draws one quad as two
 * triangles (BL,BR,TR) and (TR,TL,BL) (ShaderEngine.cpp:2945-2960, :1448) with an identity
 * MVP (:2152-2162), so gl_FragCoord = (x+.5, y+.5) and every varying is a plane a0 + dadx*x +
 * dady*y whose coefficients the rasteriser derives per triangle from the vertex values.
 * llvmpipe's setup (lp_state_setup.c emit_linear_coef; provoking-vertex rotation in
 * lp_setup_vbuf.c) is restated below and reproduces the driver's TexCoord varying
 * bit-for-bit for every target size used by the configs (oracle/probes; /tmp fit in the
 * round-1 log).  Only one of dadx/dady is non-zero for every varying in the supported
 * shaders, so the order of the two fused multiply-adds is immaterial.
 */
#include <math.h>

#include "rc_oracle.h"

o_varying o_varying_setup(float a_bl, float a_br, float a_tr, float a_tl, int W, int H, int out_fmt) {
  o_varying v;
  float fw = (float)W, fh = (float)H;
  float ooa = 1.0f / (fw * fh);
  float hy = fh * ooa; /* |dy * ooa| of the vertical edge  */
  float wx = fw * ooa; /* |dx * ooa| of the horizontal edge */
  /* lower-right triangle, set up as (v0,v1,v2) = (BR,TR,BL) */
  v.dx_lo = hy * (a_br - a_bl);
  v.dy_lo = wx * (a_tr - a_br);
  v.a0_lo = a_br - (v.dx_lo * (fw - 0.5f) + v.dy_lo * (0.0f - 0.5f));
  /* upper-left triangle, set up as (v0,v1,v2) = (TL,BL,TR) */
  v.dx_up = hy * (a_tr - a_tl);
  v.dy_up = wx * (a_tl - a_bl);
  v.a0_up = a_tl - (v.dx_up * (0.0f - 0.5f) + v.dy_up * (fh - 0.5f));
  if (out_fmt == O_FMT_RGBA8) {
    /* Plain 8-bit UNORM colour buffers take llvmpipe's rectangle path (lp_setup_rect.c: the
     * two triangles are recognised as one screen-aligned rect): ONE plane for the whole
     * target, anchored at the TR vertex.  Measured: TexCoord bits recovered through
     * floatBitsToInt match this for RGBA8 targets, and the two-triangle planes for
     * SRGB8_ALPHA8 and RGBA32F targets. */
    float dx = hy * (a_tr - a_tl), dy = wx * (a_tr - a_br);
    float a0 = a_tr - (dx * (fw - 0.5f) + dy * (fh - 0.5f));
    v.dx_lo = v.dx_up = dx;
    v.dy_lo = v.dy_up = dy;
    v.a0_lo = v.a0_up = a0;
  }
  return v;
}

/* The quad of the GL's own blits (glGenerateMipmap: Mesa's u_blitter draws a TRIANGLE_FAN BL, BR, TR, TL): the first
 * triangle is the pass quad's, the second reaches the rasteriser as (TR, TL, BL) - anchored at TR, with the diagonal
 * edge in its y slope (lp_state_setup.c emit_linear_coef restated for these vertices). */
o_varying o_varying_setup_fan(float a_bl, float a_br, float a_tr, float a_tl, int W, int H, int out_fmt) {
  o_varying v = o_varying_setup(a_bl, a_br, a_tr, a_tl, W, H, out_fmt);
  if (out_fmt == O_FMT_RGBA8) return v;
  const float fw = (float)W, fh = (float)H;
  const float ooa = 1.0f / (fw * fh), hy = fh * ooa, wx = fw * ooa;
  v.dx_up = hy * (a_tr - a_tl);
  v.dy_up = wx * (a_tr - a_bl) - wx * (a_tr - a_tl);
  v.a0_up = a_tr - (v.dx_up * (fw - 0.5f) + v.dy_up * (fh - 0.5f));
  return v;
}

float o_varying_at(const o_varying* v, int x, int y, int lower) {
  if (lower) return fmaf(v->dy_lo, (float)y, fmaf(v->dx_lo, (float)x, v->a0_lo));
  return fmaf(v->dy_up, (float)y, fmaf(v->dx_up, (float)x, v->a0_up));
}
