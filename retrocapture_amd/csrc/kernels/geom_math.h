// crt/shaders/crt-geom.glsl: the tube geometry shared by its vertex shader (evaluated once per launch on the host,
// kernel_registry.cpp setupCrtGeom) and its fragment shader (kernels/pass_geom.hip).  Reference
// shaders/shaders_glsl/crt/shaders/crt-geom.glsl: intersect 117-123 / 293-299, bkwtrans 125-142 / 301-318,
// fwtrans 144-151, maxscale 153-160.  Operation order as Mesa llvmpipe evaluates it (oracle/rc_passes_geom.c):
// a constant or plain addend next to a two-term dot joins the dot's inner product (x*x + (y*y + 1)), the constant
// C of intersect() shares d*cos*cos with B, 1 - (t + 0.5) is re-associated to 0.5 - t.
#pragma once
#include "rc_device.h"

namespace rcgeom {
using namespace rcd;

// parameter block: the shader's 17 #pragma parameters in declaration order, then what the vertex shader derives
enum : int {
  GP_CRTGAMMA = 0, GP_MONGAMMA, GP_D, GP_CURVATURE, GP_R, GP_CORNERSIZE, GP_CORNERSMOOTH, GP_XTILT, GP_YTILT, GP_OVERSCAN_X,
  GP_OVERSCAN_Y, GP_DOTMASK, GP_SHARPER, GP_SCANLINE_WEIGHT, GP_LUM, GP_INTERLACE, GP_SATURATION,
  GP_SIN_X = 17, GP_SIN_Y, GP_COS_X, GP_COS_Y, GP_STRETCH_X, GP_STRETCH_Y, GP_STRETCH_Z, GP_COUNT
};

struct Tube { float R, d, sx, sy, cx, cy; };
struct V2 { float x, y; };

// SSE min/max operand order: a NaN in either operand returns the second
RC_HD float minps(float a, float b) { return a < b ? a : b; }
RC_HD float maxps(float a, float b) { return a > b ? a : b; }
RC_HD float fix(float c) { return maxps(__builtin_fabsf(c), 1e-5f); }

// Mesa's acos: pi/2 - sign(x) * (pi/2 - sqrt(1 - |x|) * (pi/2 + |x| * (pi/4 - 1 + |x| * (p0 + |x| * p1)))), unfused
RC_HD float acos_(float x) {
  const float half_pi = 1.57079637f;
  const float ax = __builtin_fabsf(x);
  float e = ax * -0.02363318f + 0.08132463f;
  e = ax * e + (0.785398185f - 1.0f);
  e = ax * e + half_pi;
  const float sg = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
  return half_pi - sg * (half_pi - __builtin_sqrtf(1.0f - ax) * e);
}

RC_HD float intersect(const Tube& g, V2 p) {
  const float R = g.R, d = g.d;
  const float dcc = (d * g.cx) * g.cy;
  const float A = (p.x * p.x + p.y * p.y) + d * d;
  const float B = 2.0f * (R * ((p.x * g.sx + p.y * g.sy) - dcc) - d * d);
  const float C = d * d + 2.0f * (dcc * R);
  return (-B - __builtin_sqrtf(B * B - (4.0f * A) * C)) / (2.0f * A);
}

template <bool FIXED>
RC_HD V2 bkwtrans(const Tube& g, V2 p) {
  const float R = g.R;
  const float c = intersect(g, p);
  V2 pt = {c * p.x, c * p.y};
  pt.x = (pt.x - (-R) * g.sx) / R;
  pt.y = (pt.y - (-R) * g.sy) / R;
  const float tx = g.sx / g.cx, ty = g.sy / g.cy;
  const float qx = pt.x / g.cx, qy = pt.y / g.cy;
  const float A = tx * tx + (ty * ty + 1.0f);
  const float B = -2.0f * (qx * tx + qy * ty);
  const float C = qx * qx + (qy * qy - 1.0f);
  const float a = (-B + __builtin_sqrtf(B * B - (4.0f * A) * C)) / (2.0f * A);
  const float ux = (pt.x - a * g.sx) / g.cx, uy = (pt.y - a * g.sy) / g.cy;
  float r = R * acos_(a);
  if (FIXED) r = fix(r);   // the fragment shader's copy wraps r in FIX(), the vertex shader's does not
  const float s = sin_(r / R);
  return V2{(ux * r) / s, (uy * r) / s};
}

RC_HD V2 fwtrans(const Tube& g, V2 uv) {
  const float R = g.R, d = g.d;
  const float r = fix(__builtin_sqrtf(uv.x * uv.x + uv.y * uv.y));
  const float k = sin_(r / R) / r;
  uv.x *= k;
  uv.y *= k;
  const float x = 1.0f - cos_(r / R);
  const float D = uv.x * g.sx + (uv.y * g.sy + (d / R + (x * g.cx) * g.cy));
  return V2{(d * (uv.x * g.cx - x * g.sx)) / D, (d * (uv.y * g.cy - x * g.sy)) / D};
}

// the vertex shader's sinangle / cosangle / stretch (VS main 190-193; aspect = (1.0, 0.75))
RC_HD void vertex_constants(float* P) {
  Tube g;
  g.R = P[GP_R];
  g.d = P[GP_D];
  g.sx = sin_(P[GP_XTILT]) + 0.001f;
  g.sy = sin_(P[GP_YTILT]) + 0.001f;
  g.cx = cos_(P[GP_XTILT]) + 0.001f;
  g.cy = cos_(P[GP_YTILT]) + 0.001f;
  const float ax = 1.0f, ay = 0.75f;
  const float den = 1.0f + ((g.R / g.d) * g.cx) * g.cy;
  const V2 c = bkwtrans<false>(g, V2{(-g.R * g.sx) / den, (-g.R * g.sy) / den});
  const float hx = 0.5f * ax, hy = 0.5f * ay;
  const float lox = fwtrans(g, V2{-hx, c.y}).x / ax, loy = fwtrans(g, V2{c.x, -hy}).y / ay;
  const float hix = fwtrans(g, V2{hx, c.y}).x / ax, hiy = fwtrans(g, V2{c.x, hy}).y / ay;
  P[GP_SIN_X] = g.sx;
  P[GP_SIN_Y] = g.sy;
  P[GP_COS_X] = g.cx;
  P[GP_COS_Y] = g.cy;
  P[GP_STRETCH_X] = ((hix + lox) * ax) * 0.5f;
  P[GP_STRETCH_Y] = ((hiy + loy) * ay) * 0.5f;
  P[GP_STRETCH_Z] = maxps(hix - lox, hiy - loy);
}

}  // namespace rcgeom
