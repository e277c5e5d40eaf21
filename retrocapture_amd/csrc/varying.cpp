#include "varying.h"

namespace rc {

rcd::Plane makePlane(float a_bl, float a_br, float a_tr, float a_tl, int W, int H, int out_fmt) {
  rcd::Plane p;
  const float fw = (float)W, fh = (float)H;
  const float ooa = 1.0f / (fw * fh);
  const float hy = fh * ooa;
  const float wx = fw * ooa;
  if (out_fmt == rcd::FMT_RGBA8) {
    const float dx = hy * (a_tr - a_tl), dy = wx * (a_tr - a_br);
    const float a0 = a_tr - (dx * (fw - 0.5f) + dy * (fh - 0.5f));
    p.dx_lo = p.dx_up = dx;
    p.dy_lo = p.dy_up = dy;
    p.a0_lo = p.a0_up = a0;
    return p;
  }
  // triangle (BL,BR,TR), set up from vertex BR
  p.dx_lo = hy * (a_br - a_bl);
  p.dy_lo = wx * (a_tr - a_br);
  p.a0_lo = a_br - (p.dx_lo * (fw - 0.5f) + p.dy_lo * (0.0f - 0.5f));
  // triangle (TR,TL,BL), set up from vertex TL
  p.dx_up = hy * (a_tr - a_tl);
  p.dy_up = wx * (a_tl - a_bl);
  p.a0_up = a_tl - (p.dx_up * (0.0f - 0.5f) + p.dy_up * (fh - 0.5f));
  return p;
}

// The quad of the GL's own blits (glGenerateMipmap: Mesa's u_blitter draws the TRIANGLE_FAN BL, BR, TR, TL): the first
// triangle is the pass quad's, the second reaches the rasteriser as (TR, TL, BL) - set up from vertex TR with the
// diagonal edge in its y slope (oracle/rc_varying.c o_varying_setup_fan; float mip levels bit-identical with it).
rcd::Plane makePlaneFan(float a_bl, float a_br, float a_tr, float a_tl, int W, int H, int out_fmt) {
  rcd::Plane p = makePlane(a_bl, a_br, a_tr, a_tl, W, H, out_fmt);
  if (out_fmt == rcd::FMT_RGBA8) return p;
  const float fw = (float)W, fh = (float)H;
  const float ooa = 1.0f / (fw * fh), hy = fh * ooa, wx = fw * ooa;
  p.dx_up = hy * (a_tr - a_tl);
  p.dy_up = wx * (a_tr - a_bl) - wx * (a_tr - a_tl);
  p.a0_up = a_tr - (p.dx_up * (fw - 0.5f) + p.dy_up * (fh - 0.5f));
  return p;
}

}  // namespace rc
