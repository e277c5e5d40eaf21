#include "kernel_registry.h"
#include "list_setup.h"

#include <cmath>
#include <cstring>

#include "kernels/geom_math.h"
#include "royale_setup.h"
#include "varying.h"

namespace rc {
namespace {

void setupTexCoord(const PassGeometry& g, rcd::PassLaunch& L) {
  L.plane[0] = planeU(1.0f, g.out_w, g.out_h, g.out_fmt);
  L.plane[1] = planeV(1.0f, g.out_w, g.out_h, g.out_fmt);
}

void setupNesMini(const PassGeometry& g, rcd::PassLaunch& L) {
  // VS: TEX0 = TexCoord * 1.00001 (crt-nes-mini.glsl:42)
  L.plane[0] = planeU(1.00001f, g.out_w, g.out_h, g.out_fmt);
  L.plane[1] = planeV(1.00001f, g.out_w, g.out_h, g.out_fmt);
}

void setupCrtPi(const PassGeometry& g, rcd::PassLaunch& L) {
  // VS: TEX0 = TexCoord * 1.0001 (crt-pi.glsl:101)
  L.plane[0] = planeU(1.0001f, g.out_w, g.out_h, g.out_fmt);
  L.plane[1] = planeV(1.0001f, g.out_w, g.out_h, g.out_fmt);
}

// crt-geom.glsl VS 176-206: TEX0 = TexCoord * 1.0001, mod_factor = TexCoord.x * TextureSize.x * OutputSize.x / InputSize.x,
// and sinangle / cosangle / stretch, which depend on uniforms only (the same value at all four vertices, so the plane
// equations hand every pixel that value unchanged): evaluated here with the device's own float primitives.
void setupCrtGeom(const PassGeometry& g, rcd::PassLaunch& L) {
  L.plane[0] = planeU(1.0001f, g.out_w, g.out_h, g.out_fmt);
  L.plane[1] = planeV(1.0001f, g.out_w, g.out_h, g.out_fmt);
  const float tsx = (float)g.in_w;
  const float mf1 = ((1.0f * tsx) * (float)g.out_w) / tsx;
  L.plane[2] = makePlane(0.f, mf1, mf1, 0.f, g.out_w, g.out_h, g.out_fmt);
  rcgeom::vertex_constants(L.params);
}
// response-time.glsl FS 126-133: response_time and pow(response_time, 2.0 .. 7.0) as the GL evaluates them - the
// exponents 2 and 4 are lowered to multiplications, the others run the exp2/log2 polynomials
void setupResponseTime(const PassGeometry& g, rcd::PassLaunch& L) {
  setupTexCoord(g, L);
  const float rt = L.params[0];
  L.params[1] = rt;
  L.params[2] = rt * rt;
  L.params[3] = rcd::pow_(rt, 3.0f);
  L.params[4] = (rt * rt) * (rt * rt);
  L.params[5] = rcd::pow_(rt, 5.0f);
  L.params[6] = rcd::pow_(rt, 6.0f);
  L.params[7] = rcd::pow_(rt, 7.0f);
}
// handheld/shaders/color/*-color.glsl: the constants of kernels/pass_basic.hip k_color_matrix (the GL's folded forms, oracle/rc_passes_basic.c)
struct ColorSpec {
  float ga, gs, lum, m[9], inv;
  bool factored_blue;
};
void setupColor(const PassGeometry& g, rcd::PassLaunch& L, const ColorSpec& c) {
  setupTexCoord(g, L);
  L.params[8] = c.ga + (c.gs != 0.0f ? L.params[0] : 0.0f) * c.gs;   // target_gamma +/- the shader's one parameter (gs = 0: none)
  L.params[9] = c.lum;
  for (int k = 0; k < 9; ++k) L.params[10 + k] = c.m[k];
  L.params[19] = c.inv;
  L.params[20] = c.factored_blue ? 1.0f : 0.0f;
}
#define RC_COLOR_SETUP(fn, ...)                                        \
  void fn(const PassGeometry& g, rcd::PassLaunch& L) {                  \
    static const ColorSpec spec = __VA_ARGS__;                          \
    setupColor(g, L, spec);                                             \
  }
RC_COLOR_SETUP(setupGbaColor, {2.2f, 1.0f, 0.94f, {0.82f, 0.24f, -0.06f, 0.125f, 0.665f, 0.21f, 0.195f, 0.075f, 0.73f}, 1.0f / 2.2f, false})
RC_COLOR_SETUP(setupGbcColor, {2.2f, -1.0f, 0.94f, {0.82f, 0.24f, -0.06f, 0.125f, 0.665f, 0.21f, 0.195f, 0.075f, 0.73f}, 1.0f / 2.2f, false})
RC_COLOR_SETUP(setupNdsColor, {1.91f, 0.0f, 0.89f, {0.87f, 0.255f, -0.125f, 0.10f, 0.645f, 0.255f, 0.10f, 0.17f, 0.73f}, 1.0f / 1.91f, false})
RC_COLOR_SETUP(setupPalmColor, {2.2f, 0.0f, 1.0f, {0.83f, 0.26f, -0.09f, 0.073f, 0.677f, 0.25f, 0.085f, 0.12f, 0.795f}, 1.0f / 2.2f, false})
RC_COLOR_SETUP(setupPspColor, {2.21f, 0.0f, 1.0f, {0.98f, 0.20f, -0.18f, 0.04f, 0.795f, 0.165f, 0.01f, 0.01f, 0.98f}, 1.0f / 2.2f, true})
RC_COLOR_SETUP(setupVbaColor, {1.45f, 1.7f, 1.0f, {0.73f, 0.27f, 0.0f, 0.085f, 0.675f, 0.24f, 0.085f, 0.24f, 0.675f}, 1.0f / 1.45f, false})
#undef RC_COLOR_SETUP

// gb-pass-5.glsl VS 46-58: TEX0 (the frame scaled about its centre) and tex_border at the quad's vertices, in the GL's operation order
void setupGbPass5(const PassGeometry& g, rcd::PassLaunch& L) {
  const float* P = L.params;
  const float osx = (float)g.out_w, osy = (float)g.out_h, isx = (float)g.in_w, isy = (float)g.in_h, tsx = isx;
  // at pass index 3 - where this pass sits in the 4-pass presets - the reference hands TextureSize.y the TARGET's height (ShaderEngine.cpp:2418-2421)
  const float tsy = (g.pass_index == 3 && g.out_h != g.in_h) ? (float)g.out_h : isy;
  const float scx = (osx / isx) / P[0], scy = (osy / isy) / P[0];
  const float mx = (0.5f * isx) / tsx, my = (0.5f * isy) / tsy;
  static const float tc[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};   // BL, BR, TR, TL
  float v[4][4];
  for (int k = 0; k < 4; ++k) {
    const float tx = mx + (tc[k][0] + -mx) * scx, ty = my + (tc[k][1] + -my) * scy;
    const float bx = tx * (tsx / isx) + -0.5f, by = ty * (tsy / isy) + -0.5f;
    v[k][0] = tx;
    v[k][1] = ty;
    v[k][2] = 0.5f + ((bx * osx) / P[1]) / scx;
    v[k][3] = 0.5f + ((by * osy) / P[2]) / scy;
  }
  for (int c = 0; c < 4; ++c) L.plane[c] = makePlane(v[0][c], v[1][c], v[2][c], v[3][c], g.out_w, g.out_h, g.out_fmt);
}
// imgborder-*.glsl VS 93-103: screen_coord (the frame placed and scaled) and TEX0 (the border zoomed about its centre) at the vertices
void setupImgBorder(const PassGeometry& g, rcd::PassLaunch& L) {
  const float* P = L.params;
  const float osx = (float)g.out_w, osy = (float)g.out_h, isx = (float)g.in_w, isy = (float)g.in_h, tsx = isx;
  const float tsy = (g.pass_index == 3 && g.out_h != g.in_h) ? (float)g.out_h : isy;   // the reference's TextureSize.y rule for pass index 3
  const float mx = (P[1] * isx) / tsx, my = (P[2] * isy) / tsy;
  const float scx = (osx / P[3]) / P[0], scy = (osy / P[4]) / P[0];
  const float rx = tsx / isx, ry = tsy / isy;
  static const float tc[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};   // BL, BR, TR, TL
  float v[4][4];
  for (int k = 0; k < 4; ++k) {
    v[k][0] = mx + (tc[k][0] + -mx) * scx;
    v[k][1] = my + (tc[k][1] + -my) * scy;
    v[k][2] = 0.4999f + (tc[k][0] * rx + -0.4999f) * P[6];
    v[k][3] = 0.4999f + (tc[k][1] * ry + -0.4999f) * P[7];
  }
  for (int c = 0; c < 4; ++c) L.plane[c] = makePlane(v[0][c], v[1][c], v[2][c], v[3][c], g.out_w, g.out_h, g.out_fmt);
}
// handheld/console-border/shader-files/border.glsl: the same shader without the four OS_MASK parameters (their value is 0)
void setupConsoleBorder(const PassGeometry& g, rcd::PassLaunch& L) {
  for (int k = 8; k < 12; ++k) L.params[k] = 0.0f;
  setupImgBorder(g, L);
}
// ntsc-gauss-pass.glsl VS: pix_no = TexCoord.y * TextureSize.y and one = 1 / TextureSize.y
void setupNtscGauss(const PassGeometry& g, rcd::PassLaunch& L) {
  setupTexCoord(g, L);
  const float tsy = (g.pass_index == 3 && g.out_h != g.in_h) ? (float)g.out_h : (float)g.in_h;   // the reference's TextureSize.y rule for pass index 3
  const float one = 1.0f / tsy;
  L.plane[2] = makePlane(0.f * tsy, 0.f * tsy, 1.f * tsy, 1.f * tsy, g.out_w, g.out_h, g.out_fmt);
  L.plane[3] = makePlane(one, one, one, one, g.out_w, g.out_h, g.out_fmt);
}
// interlacing.glsl: TextureSize.y as the reference hands it to pass index 3 (the target's height when the pass scales its height)
void setupInterlacing(const PassGeometry& g, rcd::PassLaunch& L) {
  setupTexCoord(g, L);
  L.params[8] = (g.pass_index == 3 && g.out_h != g.in_h) ? (float)g.out_h : (float)g.in_h;
}
// shutter-3d.glsl VS 61-73: left_coord / right_coord at the quad's vertices, in the GL's operation order (oracle/rc_passes_basic.c)
void setupShutter3d(const PassGeometry& g, rcd::PassLaunch& L) {
  const float* P = L.params;
  const float isx = (float)g.in_w, isy = (float)g.in_h, tsx = isx, tsy = isy;
  static const float tc[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};   // BL, BR, TR, TL
  float v[4][4];
  for (int k = 0; k < 4; ++k) {
    const float hx = (0.5f * isx) / tsx, hy = (0.5f * isy) / tsy;
    const float tx = tc[k][0] + -hx, ty = tc[k][1] + -hy;
    float x = (tx * 2.0f) * P[0] + P[2], y = (ty * P[0]) * (1.0f / P[5]) + P[1];
    x = x + hx;
    y = y + hy;
    const float sx = ((0.5f + P[3]) * isx) / tsx, sy = 0.0f / tsy;
    v[k][0] = x + -sx;
    v[k][1] = y + -sy;
    v[k][2] = x + sx;
    v[k][3] = y + sy;
  }
  for (int c = 0; c < 4; ++c) L.plane[c] = makePlane(v[0][c], v[1][c], v[2][c], v[3][c], g.out_w, g.out_h, g.out_fmt);
}
// glow/blur_{horiz,vert}.glsl: the nine weights exp(-0.35 i^2) and their sum.  The loop is unrolled by the GL's compiler
// and the constant folded after exp(x) -> exp2(x * log2e) with the constant factor moved onto one operand:
// exp2f(i * (i * (-0.35f * log2e))) in float (oracle/rc_passes_glow.c glow_blur).
void setupGlowBlur(const PassGeometry& g, rcd::PassLaunch& L) {
  setupTexCoord(g, L);
  float total = 0.0f;
  for (int i = -4; i <= 4; ++i) {
    const float fi = (float)i;
    const float k = exp2f(fi * (fi * (-0.35f * 1.4426950408889634f)));
    L.params[i + 4] = k;
    total += k;
  }
  L.params[9] = total;
}

// crt-hyllian-glow.glsl: invX from HFILTER_SHARPNESS (FS 157-165) and the beam profile (FS 174-187)
void setupHyllianGlow(const PassGeometry& g, rcd::PassLaunch& L) {
  setupTexCoord(g, L);
  const float* P = L.params;
  float bp[4] = {P[3], P[1], P[2], P[4]};
  static const float prof[6][4] = {{0.40f, 1.00f, 1.00f, 1.00f}, {0.72f, 1.00f, 1.00f, 1.25f}, {0.60f, 0.50f, 1.00f, 1.25f},
                                   {0.60f, 0.72f, 1.00f, 1.25f}, {0.68f, 0.68f, 1.00f, 1.25f}, {0.70f, 0.50f, 1.00f, 1.80f}};
  for (int k = 1; k <= 6; ++k)
    if (P[0] == (float)k)
      for (int c = 0; c < 4; ++c) bp[c] = prof[k - 1][c];
  const float B = 1.0f - P[5], C = P[5] * 0.5f;
  const float m[16] = {(-B - 6.0f * C) / 6.0f, (12.0f - 9.0f * B - 6.0f * C) / 6.0f, -(12.0f - 9.0f * B - 6.0f * C) / 6.0f, (B + 6.0f * C) / 6.0f,
                       (3.0f * B + 12.0f * C) / 6.0f, (-18.0f + 12.0f * B + 6.0f * C) / 6.0f, (18.0f - 15.0f * B - 12.0f * C) / 6.0f, -C,
                       (-3.0f * B - 6.0f * C) / 6.0f, 0.0f, (3.0f * B + 6.0f * C) / 6.0f, 0.0f,
                       B / 6.0f, (6.0f - 2.0f * B) / 6.0f, B / 6.0f, 0.0f};
  for (int k = 0; k < 16; ++k) L.params[16 + k] = m[k];
  for (int k = 0; k < 4; ++k) L.params[32 + k] = bp[k];
}

// The device's vary() and GL_NEAREST index, evaluated on the host with the same operations.
float hostVary(const rcd::Plane& p, int x, int y) { return std::fmaf(p.dy_lo, (float)y, std::fmaf(p.dx_lo, (float)x, p.a0_lo)); }
int hostNearest(float s, int n) { return (int)std::floor(s * (float)n); }

// ntsc-pass1-svideo-3phase.glsl:69  pix_no = vTexCoord * SourceSize.xy * (outsize.xy / InputSize.xy)
void setupNtscPass1(const PassGeometry& g, rcd::PassLaunch& L) {
  const float tsx = (float)g.in_w, tsy = (float)g.in_h;
  const float px1 = (1.0f * tsx) * ((float)g.out_w / tsx), py1 = (1.0f * tsy) * ((float)g.out_h / tsy);
  L.plane[0] = planeU(1.0f, g.out_w, g.out_h, g.out_fmt);
  L.plane[1] = planeV(1.0f, g.out_w, g.out_h, g.out_fmt);
  L.plane[2] = makePlane(0.f, px1, px1, 0.f, g.out_w, g.out_h, g.out_fmt);
  L.plane[3] = makePlane(0.f, 0.f, py1, py1, g.out_w, g.out_h, g.out_fmt);
}

// True when the 49 taps of ntsc pass 2 form, for every target column, the consecutive source columns
// c(x) - 24 .. c(x) + 24 with c(x + 1) = c(x) + 2 (evaluated with the kernel's own float operations on
// the actual plane): then pass_ntsc.hip may stage one row segment per wave instead of fetching per tap.
bool ntscTapsAreRegular(const PassGeometry& g, const rcd::PassLaunch& L, int taps) {
  if (g.out_fmt != rcd::FMT_RGBA8) return false;  // one plane per varying only on the rectangle path
  const rcd::Plane &pu = L.plane[0], &pv = L.plane[1];
  if (pu.dy_lo != 0.0f || pv.dx_lo != 0.0f || pu.a0_lo != pu.a0_up || pu.dx_lo != pu.dx_up || pv.a0_lo != pv.a0_up ||
      pv.dy_lo != pv.dy_up)
    return false;
  const float one_x = 1.0f / (float)g.in_w;
  int prev = 0;
  for (int x = 0; x < g.out_w; ++x) {
    const float u = hostVary(pu, x, 0);
    const int c = hostNearest(u, g.in_w);
    if (x > 0 && c != prev + 2) return false;
    prev = c;
    for (int k = 1; k <= taps; ++k) {
      const float off = (float)(k - 1 - taps);
      if (hostNearest(u + off * one_x, g.in_w) != c + (k - 1 - taps)) return false;
      if (hostNearest(u + (-off) * one_x, g.in_w) != c - (k - 1 - taps)) return false;
    }
  }
  return true;
}

// ntsc-pass2-3phase-gamma.glsl:48  TEX0.xy = TexCoord.xy - vec2(0.5 / SourceSize.x, 0.0)
void setupNtscPass2Taps(const PassGeometry& g, rcd::PassLaunch& L, int taps) {
  const float sh = 0.5f / (float)g.in_w;
  L.plane[0] = makePlane(0.f - sh, 1.f - sh, 1.f - sh, 0.f - sh, g.out_w, g.out_h, g.out_fmt);
  L.plane[1] = planeV(1.0f, g.out_w, g.out_h, g.out_fmt);
  static thread_local struct { int v[6]; bool ok; } memo = {{-1, -1, -1, -1, -1, -1}, false};
  const int key[6] = {g.in_w, g.in_h, g.out_w, g.out_h, g.out_fmt, taps};
  if (std::memcmp(memo.v, key, sizeof(key)) != 0) {
    std::memcpy(memo.v, key, sizeof(key));
    memo.ok = ntscTapsAreRegular(g, L, taps);
  }
  if (memo.ok) L.flags |= rcd::RC_FLAG_NTSC_REGULAR;
}
void setupNtscPass2(const PassGeometry& g, rcd::PassLaunch& L) { setupNtscPass2Taps(g, L, 24); }
void setupNtscPass2TwoPhase(const PassGeometry& g, rcd::PassLaunch& L) { setupNtscPass2Taps(g, L, 32); }

// xbr-lv3's rule tests depend only on the 21-texel neighbourhood of the source pixel a target
// pixel falls in, PROVIDED the five columns (rows) it samples through five separately
// interpolated coordinates are exactly centre-2 .. centre+2 (before wrapping).  Float rounding
// breaks that where a sample position lands on a texel boundary (e.g. 224 -> 2160 rows: every
// 135th half-row), so the actual planes are evaluated for every target column and row, with the
// device's operations, and the irregular ones are listed: pass_xbr.hip evaluates the rules once
// per source pixel and re-renders the listed rows / columns with the general per-pixel form.
struct XbrPattern {
  bool usable = false;
  int n_rows = 0, n_cols = 0;
  int rows[20], cols[20];
};
XbrPattern xbrPattern(const PassGeometry& g, const rcd::PassLaunch& L) {
  XbrPattern p;
  if (g.out_fmt != rcd::FMT_RGBA8) return p;  // one plane per varying only on the rectangle path
  for (int k = 0; k < 5; ++k) {
    const rcd::Plane &px = L.plane[k], &py = L.plane[5 + k];
    if (px.dy_lo != 0.0f || py.dx_lo != 0.0f) return p;
    if (px.a0_lo != px.a0_up || px.dx_lo != px.dx_up || py.a0_lo != py.a0_up || py.dy_lo != py.dy_up) return p;
  }
  for (int x = 0; x < g.out_w; ++x) {
    const int c = hostNearest(hostVary(L.plane[2], x, 0), g.in_w);
    if (c < 0 || c >= g.in_w) return p;
    bool regular = true;
    for (int k = 0; k < 5; ++k) regular = regular && hostNearest(hostVary(L.plane[k], x, 0), g.in_w) == c + k - 2;
    if (!regular) {
      if (p.n_cols == 20) return p;
      p.cols[p.n_cols++] = x;
    }
  }
  for (int y = 0; y < g.out_h; ++y) {
    const int c = hostNearest(hostVary(L.plane[7], 0, y), g.in_h);
    if (c < 0 || c >= g.in_h) return p;
    bool regular = true;
    for (int k = 0; k < 5; ++k) regular = regular && hostNearest(hostVary(L.plane[5 + k], 0, y), g.in_h) == c + k - 2;
    if (!regular) {
      if (p.n_rows == 20) return p;
      p.rows[p.n_rows++] = y;
    }
  }
  p.usable = true;
  return p;
}

uint64_t scratchXbrLv3(const PassGeometry& g) { return (uint64_t)g.in_w * g.in_h * 4; }  // one rule record per source pixel

// xbr-lv3.glsl:82-99  t1..t7 = TEX0 + multiples of one texel
void setupXbrLv3(const PassGeometry& g, rcd::PassLaunch& L) {
  const float dx = 1.0f / (float)g.in_w, dy = 1.0f / (float)g.in_h;
  const float xo[5] = {-2.0f * dx, -dx, 0.0f, dx, 2.0f * dx};
  const float yo[5] = {-2.0f * dy, -dy, 0.0f, dy, 2.0f * dy};
  for (int k = 0; k < 5; ++k) {
    L.plane[k] = makePlane(0.f + xo[k], 1.f + xo[k], 1.f + xo[k], 0.f + xo[k], g.out_w, g.out_h, g.out_fmt);
    L.plane[5 + k] = makePlane(0.f + yo[k], 0.f + yo[k], 1.f + yo[k], 1.f + yo[k], g.out_w, g.out_h, g.out_fmt);
  }
  // the check costs 5*(out_w+out_h) evaluations; geometry rarely changes between launches
  static thread_local struct { int v[5]; XbrPattern pat; } memo = {{-1, -1, -1, -1, -1}, {}};
  const int key[5] = {g.in_w, g.in_h, g.out_w, g.out_h, g.out_fmt};
  if (std::memcmp(memo.v, key, sizeof(key)) != 0) {
    std::memcpy(memo.v, key, sizeof(key));
    memo.pat = xbrPattern(g, L);
  }
  if (memo.pat.usable) {
    // ints travel as bit patterns in the float parameter block (pass_xbr.hip XBR_P_*)
    L.flags |= rcd::RC_FLAG_XBR_REGULAR;
    std::memcpy(&L.params[6], &memo.pat.n_rows, 4);
    std::memcpy(&L.params[7], &memo.pat.n_cols, 4);
    std::memcpy(&L.params[8], memo.pat.rows, sizeof(int) * (size_t)memo.pat.n_rows);
    std::memcpy(&L.params[28], memo.pat.cols, sizeof(int) * (size_t)memo.pat.n_cols);
  }
}

// xbr-lv2.glsl:111-117: the same coordinate set as xbr-lv3
void setupXbrLv2(const PassGeometry& g, rcd::PassLaunch& L) {
  const float dx = 1.0f / (float)g.in_w, dy = 1.0f / (float)g.in_h;
  const float xo[5] = {-2.0f * dx, -dx, 0.0f, dx, 2.0f * dx};
  const float yo[5] = {-2.0f * dy, -dy, 0.0f, dy, 2.0f * dy};
  for (int k = 0; k < 5; ++k) {
    L.plane[k] = makePlane(0.f + xo[k], 1.f + xo[k], 1.f + xo[k], 0.f + xo[k], g.out_w, g.out_h, g.out_fmt);
    L.plane[5 + k] = makePlane(0.f + yo[k], 0.f + yo[k], 1.f + yo[k], 1.f + yo[k], g.out_w, g.out_h, g.out_fmt);
  }
}

std::vector<KernelEntry> build() {
  std::vector<KernelEntry> r;
  r.push_back({"stock.glsl", "stock", {}, {}, rck::launch_stock, setupTexCoord, false});
  r.push_back({"scanlines/shaders/scanline.glsl", "scanline",
               {{"SCANLINE_BASE_BRIGHTNESS", 0.95f, 0.0f, 1.0f, 0.01f, "Scanline Base Brightness"},
                {"SCANLINE_SINE_COMP_A", 0.0f, 0.0f, 1.0f, 0.02f, "Grid Strength"},
                {"SCANLINE_SINE_COMP_B", 0.25f, 0.0f, 1.0f, 0.05f, "Scanline Strength"},
                {"size", 1.0f, 1.0f, 2.0f, 1.0f, "Grid size"}},
               {}, rck::launch_scanline, setupTexCoord, false});
  r.back().one_lane = true;
  r.push_back({"crt/shaders/crt-pi.glsl", "crt-pi",
               {{"CURVATURE_X", 0.10f, 0.0f, 1.0f, 0.01f, "Screen curvature - horizontal"},
                {"CURVATURE_Y", 0.15f, 0.0f, 1.0f, 0.01f, "Screen curvature - vertical"},
                {"MASK_BRIGHTNESS", 0.70f, 0.0f, 1.0f, 0.01f, "Mask brightness"},
                {"SCANLINE_WEIGHT", 6.0f, 0.0f, 15.0f, 0.1f, "Scanline weight"},
                {"SCANLINE_GAP_BRIGHTNESS", 0.12f, 0.0f, 1.0f, 0.01f, "Scanline gap brightness"},
                {"BLOOM_FACTOR", 1.5f, 0.0f, 5.0f, 0.01f, "Bloom factor"},
                {"INPUT_GAMMA", 2.4f, 0.0f, 5.0f, 0.01f, "Input gamma"},
                {"OUTPUT_GAMMA", 2.2f, 0.0f, 5.0f, 0.01f, "Output gamma"}},
               {}, rck::launch_crt_pi, setupCrtPi, false});
  r.back().one_lane = true;
  // the strip form lists the pixels its gamma tables cannot certify in this scratch: a 256-byte header and one 4-byte entry per pixel
  // (kernels/pass_crt_pi.hip)
  r.back().scratch_bytes = [](const PassGeometry& g) -> uint64_t { return 256 + (uint64_t)g.out_w * g.out_h * 4; };
  r.push_back({"dithering/shaders/bayer-matrix-dithering.glsl", "bayer-matrix-dithering",
               {{"animate", 0.0f, 0.0f, 1.0f, 1.0f, "Dithering Animation"}, {"dither_size", 0.0f, 0.0f, 0.95f, 0.05f, "Dither Size"}},
               {}, rck::launch_bayer, setupTexCoord, false});
  r.push_back({"handheld/shaders/lcd1x.glsl", "lcd1x",
               {{"BRIGHTEN_SCANLINES", 16.0f, 1.0f, 32.0f, 0.5f, "Brighten Scanlines"}, {"BRIGHTEN_LCD", 4.0f, 1.0f, 12.0f, 0.1f, "Brighten LCD"}},
               {}, rck::launch_lcd1x, setupCrtPi, true});   // same TEX0 = TexCoord * 1.0001 as crt-pi
  r.push_back({"handheld/shaders/lcd3x.glsl", "lcd3x",
               {{"brighten_scanlines", 16.0f, 1.0f, 32.0f, 0.5f, "Brighten Scanlines"}, {"brighten_lcd", 4.0f, 1.0f, 12.0f, 0.1f, "Brighten LCD"}},
               {}, rck::launch_lcd3x, setupTexCoord, true});
  r.push_back({"scalenx/shaders/epx.glsl", "epx", {}, {}, rck::launch_epx, setupTexCoord, true});
  // scalefx/scalefx.glslp (kernels/pass_scalefx.hip): TEX0 = TexCoord in every pass
  r.push_back({"scalefx/shaders/scalefx-pass0.glsl", "scalefx-pass0", {}, {}, rck::launch_scalefx0, setupTexCoord, true});
  r.push_back({"scalefx/shaders/scalefx-pass1.glsl", "scalefx-pass1",
               {{"SFX_CLR", 0.50f, 0.01f, 1.00f, 0.01f, "ScaleFX Threshold"}, {"SFX_SAA", 1.00f, 0.00f, 1.00f, 1.00f, "ScaleFX Filter AA"}},
               {}, rck::launch_scalefx1, setupTexCoord, true});
  r.push_back({"scalefx/shaders/scalefx-pass2.glsl", "scalefx-pass2", {}, {"PassPrev2Texture"}, rck::launch_scalefx2, setupTexCoord, true});
  r.push_back({"scalefx/shaders/scalefx-pass3.glsl", "scalefx-pass3", {{"SFX_SCN", 1.0f, 0.0f, 1.0f, 1.0f, "ScaleFX Filter Corners"}},
               {}, rck::launch_scalefx3, setupTexCoord, true});
  r.push_back({"scalefx/shaders/scalefx-pass4.glsl", "scalefx-pass4", {}, {"PassPrev5Texture"}, rck::launch_scalefx4, setupTexCoord, true});
  r.push_back({"interpolation/shaders/quilez.glsl", "quilez", {}, {}, rck::launch_quilez, setupTexCoord, true});
  r.push_back({"interpolation/shaders/smootheststep.glsl", "smootheststep", {}, {}, rck::launch_smootheststep, setupTexCoord, true});
  r.push_back({"interpolation/shaders/sharp-bilinear.glsl", "sharp-bilinear",
               {{"SHARP_BILINEAR_PRE_SCALE", 4.0f, 1.0f, 10.0f, 1.0f, "Sharp Bilinear Prescale"},
                {"AUTO_PRESCALE", 1.0f, 0.0f, 1.0f, 1.0f, "Automatic Prescale"}},
               {}, rck::launch_sharp_bilinear, setupTexCoord, true});
  r.push_back({"crt/shaders/crt-nes-mini.glsl", "crt-nes-mini",
               {{"SCANTHICK", 2.0f, 2.0f, 4.0f, 2.0f, "Scanline Thickness"},
                {"INTENSITY", 0.15f, 0.0f, 1.0f, 0.01f, "Scanline Intensity"},
                {"BRIGHTBOOST", 0.15f, 0.0f, 1.0f, 0.01f, "Luminance Boost"}},
               {}, rck::launch_crt_nes_mini, setupNesMini, false});
  // crt/crt-geom.glslp (kernels/pass_geom.hip); reads FrameCount (interlacing simulation for >= 400 source lines)
  r.push_back({"crt/shaders/crt-geom.glsl", "crt-geom",
               {{"CRTgamma", 2.4f, 0.1f, 5.0f, 0.1f, "CRTGeom Target Gamma"},
                {"monitorgamma", 2.2f, 0.1f, 5.0f, 0.1f, "CRTGeom Monitor Gamma"},
                {"d", 1.6f, 0.1f, 3.0f, 0.1f, "CRTGeom Distance"},
                {"CURVATURE", 1.0f, 0.0f, 1.0f, 1.0f, "CRTGeom Curvature Toggle"},
                {"R", 2.0f, 0.1f, 10.0f, 0.1f, "CRTGeom Curvature Radius"},
                {"cornersize", 0.03f, 0.001f, 1.0f, 0.005f, "CRTGeom Corner Size"},
                {"cornersmooth", 1000.0f, 80.0f, 2000.0f, 100.0f, "CRTGeom Corner Smoothness"},
                {"x_tilt", 0.0f, -0.5f, 0.5f, 0.05f, "CRTGeom Horizontal Tilt"},
                {"y_tilt", 0.0f, -0.5f, 0.5f, 0.05f, "CRTGeom Vertical Tilt"},
                {"overscan_x", 100.0f, -125.0f, 125.0f, 1.0f, "CRTGeom Horiz. Overscan %"},
                {"overscan_y", 100.0f, -125.0f, 125.0f, 1.0f, "CRTGeom Vert. Overscan %"},
                {"DOTMASK", 0.3f, 0.0f, 1.0f, 0.1f, "CRTGeom Dot Mask Strength"},
                {"SHARPER", 1.0f, 1.0f, 3.0f, 1.0f, "CRTGeom Sharpness"},
                {"scanline_weight", 0.3f, 0.1f, 0.5f, 0.05f, "CRTGeom Scanline Weight"},
                {"lum", 0.0f, 0.0f, 1.0f, 0.01f, "CRTGeom Luminance"},
                {"interlace_detect", 1.0f, 0.0f, 1.0f, 1.0f, "CRTGeom Interlacing Simulation"},
                {"SATURATION", 1.0f, 0.0f, 2.0f, 0.05f, "CRTGeom Saturation"}},
               {}, rck::launch_crt_geom, setupCrtGeom, false});
  r.push_back({"crt/shaders/crt-easymode.glsl", "crt-easymode",
               {{"SHARPNESS_H", 0.5f, 0.0f, 1.0f, 0.05f, "Sharpness Horizontal"},
                {"SHARPNESS_V", 1.0f, 0.0f, 1.0f, 0.05f, "Sharpness Vertical"},
                {"MASK_STRENGTH", 0.3f, 0.0f, 1.0f, 0.01f, "Mask Strength"},
                {"MASK_DOT_WIDTH", 1.0f, 1.0f, 100.0f, 1.0f, "Mask Dot Width"},
                {"MASK_DOT_HEIGHT", 1.0f, 1.0f, 100.0f, 1.0f, "Mask Dot Height"},
                {"MASK_STAGGER", 0.0f, 0.0f, 100.0f, 1.0f, "Mask Stagger"},
                {"MASK_SIZE", 1.0f, 1.0f, 100.0f, 1.0f, "Mask Size"},
                {"SCANLINE_STRENGTH", 1.0f, 0.0f, 1.0f, 0.05f, "Scanline Strength"},
                {"SCANLINE_BEAM_WIDTH_MIN", 1.5f, 0.5f, 5.0f, 0.5f, "Scanline Beam Width Min."},
                {"SCANLINE_BEAM_WIDTH_MAX", 1.5f, 0.5f, 5.0f, 0.5f, "Scanline Beam Width Max."},
                {"SCANLINE_BRIGHT_MIN", 0.35f, 0.0f, 1.0f, 0.05f, "Scanline Brightness Min."},
                {"SCANLINE_BRIGHT_MAX", 0.65f, 0.0f, 1.0f, 0.05f, "Scanline Brightness Max."},
                {"SCANLINE_CUTOFF", 400.0f, 1.0f, 1000.0f, 1.0f, "Scanline Cutoff"},
                {"GAMMA_INPUT", 2.0f, 0.1f, 5.0f, 0.1f, "Gamma Input"},
                {"GAMMA_OUTPUT", 1.8f, 0.1f, 5.0f, 0.1f, "Gamma Output"},
                {"BRIGHT_BOOST", 1.2f, 1.0f, 2.0f, 0.01f, "Brightness Boost"},
                {"DILATION", 1.0f, 0.0f, 1.0f, 1.0f, "Dilation"}},
               {}, rck::launch_crt_easymode, setupTexCoord, false});
  // crt/zfast-crt.glslp: the six parameters are the names the reference overwrites with fixed values (shader_engine.cpp)
  r.push_back({"crt/shaders/zfast_crt.glsl", "zfast-crt",
               {{"BLURSCALEX", 0.30f, 0.0f, 1.0f, 0.05f, "Blur Amount X-Axis"},
                {"LOWLUMSCAN", 6.0f, 0.0f, 10.0f, 0.5f, "Scanline Darkness - Low"},
                {"HILUMSCAN", 8.0f, 0.0f, 50.0f, 1.0f, "Scanline Darkness - High"},
                {"BRIGHTBOOST", 1.25f, 0.5f, 1.5f, 0.05f, "Dark Pixel Brightness Boost"},
                {"MASK_DARK", 0.25f, 0.0f, 1.0f, 0.05f, "Mask Effect Amount"},
                {"MASK_FADE", 0.8f, 0.0f, 1.0f, 0.05f, "Mask/Scanline Fade"}},
               {}, rck::launch_zfast_crt, setupCrtPi, false});
  // crt/crt-hyllian-glow.glslp, the reference's smoke-test default preset (kernels/pass_glow.hip)
  r.push_back({"crt/shaders/glow/linearize.glsl", "glow-linearize", {{"INPUT_GAMMA", 2.4f, 2.0f, 2.6f, 0.02f, "Input Gamma"}}, {},
               rck::launch_glow_linearize, setupTexCoord, false});
  r.push_back({"crt/shaders/hyllian/crt-hyllian-glow/crt-hyllian-glow.glsl", "crt-hyllian-glow",
               {{"BEAM_PROFILE", 0.0f, 0.0f, 6.0f, 1.0f, "BEAM PROFILE (BP)"},
                {"BEAM_MIN_WIDTH", 0.86f, 0.0f, 1.0f, 0.02f, "  Custom [If   BP=0.00] MIN BEAM WIDTH"},
                {"BEAM_MAX_WIDTH", 1.0f, 0.0f, 1.0f, 0.02f, "  Custom [If   BP=0.00] MAX BEAM WIDTH"},
                {"SCANLINES_STRENGTH", 0.58f, 0.0f, 1.0f, 0.02f, "  Custom [If   BP=0.00] SCANLINES STRENGTH"},
                {"COLOR_BOOST", 1.25f, 1.0f, 2.0f, 0.05f, "  Custom [If   BP=0.00] COLOR BOOST"},
                {"HFILTER_SHARPNESS", 1.0f, 0.0f, 1.0f, 0.02f, "HORIZONTAL FILTER SHARPNESS"},
                {"CRT_ANTI_RINGING", 1.0f, 0.0f, 1.0f, 0.1f, "ANTI RINGING"},
                {"InputGamma", 2.4f, 0.0f, 5.0f, 0.1f, "INPUT GAMMA"},
                {"OutputGamma", 2.2f, 0.0f, 5.0f, 0.1f, "OUTPUT GAMMA"},
                {"VSCANLINES", 0.0f, 0.0f, 1.0f, 1.0f, "SCANLINES DIRECTION"}},
               {}, rck::launch_crt_hyllian_glow, setupHyllianGlow, false});
  r.push_back({"crt/shaders/glow/threshold.glsl", "glow-threshold",
               {{"GLOW_WHITEPOINT", 1.0f, 0.5f, 1.1f, 0.02f, "Glow Whitepoint"}, {"GLOW_ROLLOFF", 3.0f, 1.2f, 6.0f, 0.1f, "Glow Rolloff"}}, {},
               rck::launch_glow_threshold, setupTexCoord, false});
  {
    KernelEntry e{"crt/shaders/glow/blur_horiz.glsl", "glow-blur-h", {}, {}, rck::launch_glow_blur_h, setupGlowBlur, false};
    e.mip_aware = true;
    e.ignores_texture_height = true;   // FS 87: dx = 4.0 * SourceSize.z only
    r.push_back(e);
  }
  r.push_back({"crt/shaders/glow/blur_vert.glsl", "glow-blur-v", {}, {}, rck::launch_glow_blur_v, setupGlowBlur, false});
  {
    KernelEntry e{"crt/shaders/hyllian/crt-hyllian-glow/resolve2.glsl", "hyllian-resolve2",
                  {{"BLOOM_STRENGTH", 0.45f, 0.0f, 0.8f, 0.05f, "Glow Strength"},
                   {"OUTPUT_GAMMA", 2.2f, 1.8f, 2.6f, 0.02f, "Monitor Gamma"},
                   {"PHOSPHOR_LAYOUT", 4.0f, 0.0f, 19.0f, 1.0f, "PHOSPHOR LAYOUT"},
                   {"MASK_INTENSITY", 0.5f, 0.0f, 1.0f, 0.1f, "MASK INTENSITY"}},
                  {"PassPrev4Texture"}, rck::launch_hyllian_resolve2, setupTexCoord, false};
    r.push_back(e);
  }
  // conformance fixture of this repository (tests/fixtures/conformance/): pins PassFeedback, which no
  // shader of the reference's tree declares
  r.push_back({"conformance/feedback-persist.glsl", "feedback-persist", {{"PERSIST", 0.8f, 0.0f, 1.0f, 0.05f, "Persistence"}},
               {"PassFeedback0", "PassFeedback1"}, rck::launch_feedback_persist, setupTexCoord, false});
  {
    // ... and one that pins the STALE SIZE UNIFORMS of the history re-draw (a history shader that reads its size uniforms)
    KernelEntry e{"conformance/history-size.glsl", "history-size", {{"HS_MIX", 0.3f, 0.0f, 1.0f, 0.05f, "History weight"}},
                  {"PrevTexture", "Prev1Texture"}, rck::launch_history_size, setupTexCoord, false};
    e.stale_size_uniforms = true;
    r.push_back(e);
  }
  r.push_back({"motionblur/shaders/mix_frames.glsl", "mix-frames", {}, {"PrevTexture"}, rck::launch_mix_frames, setupCrtPi,
               false, true, nullptr, nullptr, true});  // VS: TEX0 = TexCoord * 1.0001 (mix_frames.glsl:53)
  // the other four motionblur/ presets: frame history down to Prev6Texture (kernels/pass_basic.hip)
  r.push_back({"motionblur/shaders/motionblur-simple.glsl", "motionblur-simple", {},
               {"Prev6Texture", "Prev5Texture", "Prev4Texture", "Prev3Texture", "Prev2Texture", "Prev1Texture", "PrevTexture"},
               rck::launch_motionblur_simple, setupTexCoord, false, true, nullptr, nullptr, true});
  r.push_back({"motionblur/shaders/braid-rewind.glsl", "braid-rewind", {},
               {"Prev6Texture", "Prev5Texture", "Prev4Texture", "Prev3Texture", "Prev2Texture", "Prev1Texture", "PrevTexture"},
               rck::launch_braid_rewind, setupTexCoord, false, true, nullptr, nullptr, true});
  r.push_back({"motionblur/shaders/response-time.glsl", "response-time", {{"response_time", 0.333f, 0.0f, 0.777f, 0.111f, "LCD Response Time"}},
               {"PrevTexture", "Prev1Texture", "Prev2Texture", "Prev3Texture", "Prev4Texture", "Prev5Texture", "Prev6Texture"},
               rck::launch_response_time, setupResponseTime, false, true, nullptr, nullptr, true});
  r.push_back({"motionblur/shaders/mix_frames_smart.glsl", "mix-frames-smart", {{"DEFLICKER_EMPHASIS", 0.0f, 0.0f, 1.0f, 0.01f, "Deflicker Emphasis"}},
               {"PrevTexture", "Prev1Texture", "Prev2Texture", "Prev3Texture", "Prev4Texture"},
               rck::launch_mix_frames_smart, setupCrtPi, false, true, nullptr, nullptr, true});   // VS: TEX0 = TexCoord * 1.0001
  r.push_back({"handheld/shaders/lcd-cgwg/lcd-grid.glsl", "lcd-grid",
               {{"GRID_STRENGTH", 0.05f, 0.0f, 1.0f, 0.01f, "LCD Grid Strength"}, {"gamma", 2.2f, 1.0f, 5.0f, 0.1f, "LCD Input Gamma"}}, {},
               rck::launch_lcd_grid, setupTexCoord, false});
  // handheld/lcd-grid-v2.glslp and the lcd-grid-v2-<colour>[-motionblur] chains (kernels/pass_lcd_grid.hip)
  r.push_back({"handheld/shaders/lcd-cgwg/lcd-grid-v2.glsl", "lcd-grid-v2",
               {{"RSUBPIX_R", 1.0f, 0.0f, 1.0f, 0.01f, "Colour of R subpixel: R"}, {"RSUBPIX_G", 0.0f, 0.0f, 1.0f, 0.01f, "Colour of R subpixel: G"},
                {"RSUBPIX_B", 0.0f, 0.0f, 1.0f, 0.01f, "Colour of R subpixel: B"}, {"GSUBPIX_R", 0.0f, 0.0f, 1.0f, 0.01f, "Colour of G subpixel: R"},
                {"GSUBPIX_G", 1.0f, 0.0f, 1.0f, 0.01f, "Colour of G subpixel: G"}, {"GSUBPIX_B", 0.0f, 0.0f, 1.0f, 0.01f, "Colour of G subpixel: B"},
                {"BSUBPIX_R", 0.0f, 0.0f, 1.0f, 0.01f, "Colour of B subpixel: R"}, {"BSUBPIX_G", 0.0f, 0.0f, 1.0f, 0.01f, "Colour of B subpixel: G"},
                {"BSUBPIX_B", 1.0f, 0.0f, 1.0f, 0.01f, "Colour of B subpixel: B"}, {"gain", 1.0f, 0.5f, 2.0f, 0.05f, "Gain"},
                {"gamma", 3.0f, 0.5f, 5.0f, 0.1f, "LCD Input Gamma"}, {"outgamma", 2.2f, 0.5f, 5.0f, 0.1f, "LCD Output Gamma"},
                {"blacklevel", 0.05f, 0.0f, 0.5f, 0.01f, "Black level"}, {"ambient", 0.0f, 0.0f, 0.5f, 0.01f, "Ambient"},
                {"BGR", 0.0f, 0.0f, 1.0f, 1.0f, "BGR"}},
               {}, rck::launch_lcd_grid_v2, setupTexCoord, false});
  {
    // borders/resources/imgborder-{sgb,gameboy-player,sgba}.glsl: one text, three sets of #pragma defaults
    struct D { const char* id; const char* name; float box, rx, ry; };
    static const D variants[] = {{"borders/resources/imgborder-sgb.glsl", "imgborder-sgb", 1.0f, 160.0f, 144.0f},
                                 {"borders/resources/imgborder-gameboy-player.glsl", "imgborder-gameboy-player", 2.0f, 240.0f, 160.0f},
                                 {"borders/resources/imgborder-sgba.glsl", "imgborder-sgba", 1.0f, 240.0f, 160.0f}};
    for (const D& d : variants) {
      KernelEntry e{d.id, d.name,
                    {{"box_scale", d.box, 1.0f, 10.0f, 1.0f, "Image Scale"}, {"location_x", 0.5f, 0.0f, 1.0f, 0.05f, "Viewport X Pos."},
                     {"location_y", 0.5f, 0.0f, 1.0f, 0.05f, "Viewport Y Pos."}, {"in_res_x", d.rx, 100.0f, 600.0f, 1.0f, "Viewport Size X"},
                     {"in_res_y", d.ry, 64.0f, 512.0f, 1.0f, "Viewport Size Y"}, {"border_on_top", 0.0f, 0.0f, 1.0f, 1.0f, "Show Viewport"},
                     {"border_zoom_x", 1.0f, 0.0f, 4.0f, 0.01f, "Border Zoom X"}, {"border_zoom_y", 1.0f, 0.0f, 4.0f, 0.01f, "Border Zoom Y"},
                     {"OS_MASK_TOP", 0.0f, 0.0f, 1.0f, 0.01f, "OS Mask Top"}, {"OS_MASK_BOTTOM", 0.0f, 0.0f, 1.0f, 0.01f, "OS Mask Bottom"},
                     {"OS_MASK_LEFT", 0.0f, 0.0f, 1.0f, 0.01f, "OS Mask Left"}, {"OS_MASK_RIGHT", 0.0f, 0.0f, 1.0f, 0.01f, "OS Mask Right"}},
                    {"BORDER"}, rck::launch_imgborder, setupImgBorder, false};
      e.texture_height_override = true;   // setupImgBorder reads PassGeometry::pass_index
      r.push_back(e);
    }
  }
  {
    KernelEntry e{"handheld/console-border/shader-files/border.glsl", "console-border",
                  {{"box_scale", 4.0f, 1.0f, 10.0f, 1.0f, "Image Scale"}, {"location_x", 0.5f, 0.0f, 1.0f, 0.05f, "Viewport X Pos."},
                   {"location_y", 0.5f, 0.0f, 1.0f, 0.05f, "Viewport Y Pos."}, {"in_res_x", 320.0f, 100.0f, 600.0f, 1.0f, "Viewport Size X"},
                   {"in_res_y", 240.0f, 64.0f, 512.0f, 1.0f, "Viewport Size Y"}, {"border_on_top", 1.0f, 0.0f, 1.0f, 1.0f, "Show Viewport"},
                   {"border_zoom_x", 1.0f, 0.0f, 4.0f, 0.01f, "Border Zoom X"}, {"border_zoom_y", 1.0f, 0.0f, 4.0f, 0.01f, "Border Zoom Y"}},
                  {"BORDER"}, rck::launch_imgborder, setupConsoleBorder, false};
    e.texture_height_override = true;
    r.push_back(e);
  }
  {
    KernelEntry e{"ntsc/shaders/ntsc-gauss-pass.glsl", "ntsc-gauss",
                  {{"NTSC_CRT_GAMMA", 2.5f, 0.0f, 10.0f, 0.1f, "NTSC CRT Gamma"}, {"NTSC_DISPLAY_GAMMA", 2.1f, 0.0f, 10.0f, 0.1f, "NTSC Display Gamma"}},
                  {}, rck::launch_ntsc_gauss, setupNtscGauss, false};
    e.texture_height_override = true;
    r.push_back(e);
  }
  {
    // the two instruction-list hubs (kernels/pass_lists.hip, list_setup.cpp); both restated under the pass-index-3 TextureSize.y rule
    KernelEntry t{"crt/shaders/tvout-tweaks.glsl", "tvout-tweaks",
                  {{"TVOUT_RESOLUTION", 256.0f, 0.0f, 1024.0f, 32.0f, "TVOut Signal Resolution"}, {"TVOUT_COMPOSITE_CONNECTION", 0.0f, 0.0f, 1.0f, 1.0f, "TVOut Composite Enable"},
                   {"TVOUT_TV_COLOR_LEVELS", 0.0f, 0.0f, 1.0f, 1.0f, "TVOut TV Color Levels Enable"}, {"TVOUT_RESOLUTION_Y", 256.0f, 0.0f, 1024.0f, 32.0f, "TVOut Luma (Y) Resolution"},
                   {"TVOUT_RESOLUTION_I", 83.2f, 0.0f, 256.0f, 8.0f, "TVOut Chroma (I) Resolution"}, {"TVOUT_RESOLUTION_Q", 25.6f, 0.0f, 256.0f, 8.0f, "TVOut Chroma (Q) Resolution"}},
                  {}, rck::launch_tvout_tweaks, setupTvoutTweaks, false};
    t.texture_height_override = true;
    r.push_back(t);
    KernelEntry lo{"crt/shaders/crt-lottes.glsl", "crt-lottes",
                   {{"hardScan", -8.0f, -20.0f, 0.0f, 1.0f, "hardScan"}, {"hardPix", -3.0f, -20.0f, 0.0f, 1.0f, "hardPix"}, {"warpX", 0.031f, 0.0f, 0.125f, 0.01f, "warpX"},
                    {"warpY", 0.041f, 0.0f, 0.125f, 0.01f, "warpY"}, {"maskDark", 0.5f, 0.0f, 2.0f, 0.1f, "maskDark"}, {"maskLight", 1.5f, 0.0f, 2.0f, 0.1f, "maskLight"},
                    {"scaleInLinearGamma", 1.0f, 0.0f, 1.0f, 1.0f, "scaleInLinearGamma"}, {"shadowMask", 3.0f, 0.0f, 4.0f, 1.0f, "shadowMask"},
                    {"brightBoost", 1.0f, 0.0f, 2.0f, 0.05f, "brightness boost"}, {"hardBloomPix", -1.5f, -2.0f, -0.5f, 0.1f, "bloom-x soft"},
                    {"hardBloomScan", -2.0f, -4.0f, -1.0f, 0.1f, "bloom-y soft"}, {"bloomAmount", 0.15f, 0.0f, 1.0f, 0.05f, "bloom ammount"},
                    {"shape", 2.0f, 0.0f, 10.0f, 0.05f, "filter kernel shape"}},
                   {}, rck::launch_crt_lottes, setupCrtLottes, false};
    lo.texture_height_override = true;
    r.push_back(lo);
    KernelEntry fl{"crt/shaders/fakelottes.glsl", "fakelottes",
                   {{"shadowMask", 1.0f, 0.0f, 4.0f, 1.0f, "shadowMask"}, {"SCANLINE_SINE_COMP_B", 0.40f, 0.0f, 1.0f, 0.05f, "Scanline Intensity"},
                    {"warpX", 0.031f, 0.0f, 0.125f, 0.01f, "warpX"}, {"warpY", 0.041f, 0.0f, 0.125f, 0.01f, "warpY"}, {"maskDark", 0.5f, 0.0f, 2.0f, 0.1f, "maskDark"},
                    {"maskLight", 1.5f, 0.0f, 2.0f, 0.1f, "maskLight"}, {"crt_gamma", 2.5f, 1.0f, 4.0f, 0.05f, "CRT Gamma"},
                    {"monitor_gamma", 2.2f, 1.0f, 4.0f, 0.05f, "Monitor Gamma"}, {"SCANLINE_SINE_COMP_A", 0.0f, 0.0f, 0.10f, 0.01f, "Scanline Sine Comp A"},
                    {"SCANLINE_BASE_BRIGHTNESS", 0.95f, 0.0f, 1.0f, 0.01f, "Scanline Base Brightness"}},
                   {}, rck::launch_fakelottes, setupFakeLottes, false};
    fl.texture_height_override = true;
    r.push_back(fl);
    KernelEntry sb{"stereoscopic-3d/shaders/side-by-side-simple.glsl", "side-by-side-simple",
                   {{"eye_sep", 0.30f, -1.0f, 5.0f, 0.05f, "Eye Separation"}, {"y_loc", 0.25f, -1.0f, 1.0f, 0.01f, "Vertical Placement"},
                    {"BOTH", 0.51f, -2.0f, 2.0f, 0.005f, "Horizontal Placement"}, {"ana_zoom", 0.75f, -2.0f, 2.0f, 0.05f, "Zoom"},
                    {"WIDTH", 3.05f, 1.0f, 7.0f, 0.05f, "Side-by-Side Image Width"}, {"HEIGHT", 2.0f, 1.0f, 5.0f, 0.1f, "Side-by-Side Image Height"},
                    {"warpX", 0.1f, 0.0f, 0.5f, 0.05f, "Lens Warp Correction X"}, {"warpY", 0.1f, 0.0f, 0.5f, 0.05f, "Lens Warp Correction Y"},
                    {"pulfrich", 0.0f, 0.0f, 0.5f, 0.25f, "Pulfrich Effect"}},
                   {}, rck::launch_side_by_side, setupSideBySide, false};
    sb.texture_height_override = true;
    r.push_back(sb);
    KernelEntry sl{"handheld/shaders/sameboy-lcd.glsl", "sameboy-lcd",
                   {{"COLOR_LOW", 0.8f, 0.0f, 1.5f, 0.05f, "Color Low"}, {"COLOR_HIGH", 1.0f, 0.0f, 1.5f, 0.05f, "Color High"},
                    {"SCANLINE_DEPTH", 0.1f, 0.0f, 2.0f, 0.05f, "Scanline Depth"}},
                   {}, rck::launch_sameboy_lcd, setupSameboyLcd, false};
    sl.texture_height_override = true;
    r.push_back(sl);
    KernelEntry cc{"crt/shaders/crt-consumer.glsl", "crt-consumer",
                   {{"blurx", 0.25f, -2.0f, 2.0f, 0.05f, "Convergence X"},
                    {"blury", -0.15f, -2.0f, 2.0f, 0.05f, "Convergence Y"},
                    {"warpx", 0.03f, 0.0f, 0.12f, 0.01f, "Curvature X"},
                    {"warpy", 0.04f, 0.0f, 0.12f, 0.01f, "Curvature Y"},
                    {"corner", 0.01f, 0.0f, 0.1f, 0.01f, "Corner size"},
                    {"smoothness", 400.0f, 25.0f, 600.0f, 5.0f, "Border Smoothness"},
                    {"scanlow", 6.0f, 1.0f, 15.0f, 1.0f, "Beam low"},
                    {"scanhigh", 8.0f, 1.0f, 15.0f, 1.0f, "Beam high"},
                    {"beamlow", 1.35f, 0.5f, 2.5f, 0.05f, "Scanlines dark"},
                    {"beamhigh", 1.05f, 0.5f, 2.5f, 0.05f, "Scanlines bright"},
                    {"brightboost1", 1.1f, 0.0f, 3.0f, 0.05f, "Bright boost dark pixels"},
                    {"brightboost2", 1.05f, 0.0f, 3.0f, 0.05f, "Bright boost bright pixels"},
                    {"Shadowmask", 7.0f, -1.0f, 8.0f, 1.0f, "Mask Type"},
                    {"masksize", 1.0f, 1.0f, 2.0f, 1.0f, "Mask Size"},
                    {"MaskDark", 0.5f, 0.0f, 2.0f, 0.1f, "Mask dark"},
                    {"MaskLight", 1.5f, 0.0f, 2.0f, 0.1f, "Mask light"},
                    {"slotmask", 0.0f, 0.0f, 1.0f, 0.05f, "Slot Mask Strength"},
                    {"slotwidth", 2.0f, 1.0f, 6.0f, 0.5f, "Slot Mask Width"},
                    {"double_slot", 1.0f, 1.0f, 2.0f, 1.0f, "Slot Mask Height: 2x1 or 4x1"},
                    {"slotms", 1.0f, 1.0f, 2.0f, 1.0f, "Slot Mask Size"},
                    {"GAMMA_IN", 2.5f, 0.0f, 4.0f, 0.1f, "Gamma In"},
                    {"GAMMA_OUT", 2.2f, 0.0f, 4.0f, 0.1f, "Gamma Out"},
                    {"glow", 0.05f, 0.0f, 0.5f, 0.01f, "Glow Strength"},
                    {"Size", 1.0f, 0.1f, 4.0f, 0.05f, "Glow Size"},
                    {"sat", 1.1f, 0.0f, 2.0f, 0.05f, "Saturation"},
                    {"contrast", 1.0f, 0.0f, 2.0f, 0.05f, "Contrast, 1.0:Off"},
                    {"nois", 0.0f, 0.0f, 32.0f, 1.0f, "Noise"},
                    {"WP", 0.0f, -100.0f, 100.0f, 5.0f, "Color Temperature %"},
                    {"inter", 1.0f, 0.0f, 1.0f, 1.0f, "Interlacing Toggle"},
                    {"vignette", 1.0f, 0.0f, 1.0f, 1.0f, "Vignette On/Off"},
                    {"vpower", 0.2f, 0.0f, 1.0f, 0.01f, "Vignette Power"},
                    {"vstr", 40.0f, 0.0f, 50.0f, 1.0f, "Vignette strength"},
                    {"alloff", 0.0f, 0.0f, 1.0f, 1.0f, "Switch off shader"}},
                   {}, rck::launch_crt_consumer, setupCrtConsumer, false};
    cc.texture_height_override = true;
    r.push_back(cc);
    KernelEntry ra{"anti-aliasing/shaders/reverse-aa.glsl", "reverse-aa", {{"REVERSEAA_SHARPNESS", 2.0f, 0.0f, 10.0f, 0.01f, "ReverseAA Sharpness"}}, {}, rck::launch_reverse_aa, setupReverseAa, false};
    ra.texture_height_override = true;
    r.push_back(ra);
    KernelEntry aa{"anti-aliasing/shaders/advanced-aa.glsl", "advanced-aa",
                   {{"AA_RESOLUTION_X", 0.0f, 0.0f, 1920.0f, 1.0f, "AA Input Res X"}, {"AA_RESOLUTION_Y", 0.0f, 0.0f, 1920.0f, 1.0f, "AA Input Res Y"}},
                   {}, rck::launch_advanced_aa, setupAdvancedAa, false};
    aa.texture_height_override = true;
    r.push_back(aa);
    KernelEntry j{"windowed/shaders/jinc2-sharper.glsl", "jinc2-sharper", {}, {}, rck::launch_jinc2_sharper, setupJinc2Sharper, false};
    j.texture_height_override = true;
    r.push_back(j);
    KernelEntry a{"misc/image-adjustment.glsl", "image-adjustment",
                  {{"ia_target_gamma", 2.2f, 0.1f, 5.0f, 0.1f, "Target Gamma"}, {"ia_monitor_gamma", 2.2f, 0.1f, 5.0f, 0.1f, "Monitor Gamma"},
                   {"ia_overscan_percent_x", 0.0f, -25.0f, 25.0f, 1.0f, "Horizontal Overscan %"}, {"ia_overscan_percent_y", 0.0f, -25.0f, 25.0f, 1.0f, "Vertical Overscan %"},
                   {"ia_saturation", 1.0f, 0.0f, 5.0f, 0.1f, "Saturation"}, {"ia_contrast", 1.0f, 0.0f, 10.0f, 0.05f, "Contrast"},
                   {"ia_luminance", 1.0f, 0.0f, 2.0f, 0.1f, "Luminance"}, {"ia_black_level", 0.0f, -0.3f, 0.3f, 0.01f, "Black Level"},
                   {"ia_bright_boost", 0.0f, -1.0f, 1.0f, 0.05f, "Brightness Boost"}, {"ia_R", 1.0f, 0.0f, 2.0f, 0.05f, "Red Channel"},
                   {"ia_G", 1.0f, 0.0f, 2.0f, 0.05f, "Green Channel"}, {"ia_B", 1.0f, 0.0f, 2.0f, 0.05f, "Blue Channel"},
                   {"ia_ZOOM", 1.0f, 0.0f, 4.0f, 0.01f, "Zoom Factor"}, {"ia_XPOS", 0.0f, -2.0f, 2.0f, 0.005f, "X Modifier"},
                   {"ia_YPOS", 0.0f, -2.0f, 2.0f, 0.005f, "Y Modifier"}, {"ia_TOPMASK", 0.0f, 0.0f, 1.0f, 0.0025f, "Overscan Mask Top"},
                   {"ia_BOTMASK", 0.0f, 0.0f, 1.0f, 0.0025f, "Overscan Mask Bottom"}, {"ia_LMASK", 0.0f, 0.0f, 1.0f, 0.0025f, "Overscan Mask Left"},
                   {"ia_RMASK", 0.0f, 0.0f, 1.0f, 0.0025f, "Overscan Mask Right"}, {"ia_GRAIN_STR", 0.0f, 0.0f, 72.0f, 6.0f, "Film Grain"},
                   {"ia_SHARPEN", 0.0f, 0.0f, 1.0f, 0.05f, "Sharpen"}, {"ia_FLIP_HORZ", 0.0f, 0.0f, 1.0f, 1.0f, "Flip Horiz Axis"},
                   {"ia_FLIP_VERT", 0.0f, 0.0f, 1.0f, 1.0f, "Flip Vert Axis"}},
                  {}, rck::launch_image_adjustment, setupImageAdjustment, false};
    a.texture_height_override = true;
    // the shader "flips" by moving the quad itself (flip_pos = 1 - VertexCoord on clip-space coordinates: x in [0, 2]), which leaves half the
    // target to the clear colour and the rest to clipped triangles: not restated
    a.validate = [](const float* p) -> const char* {
      return (p[21] > 0.5f || p[22] > 0.5f) ? "image-adjustment.glsl: ia_FLIP_HORZ / ia_FLIP_VERT move the quad half off the target (clipped geometry), which is not restated"
                                           : nullptr;
    };
    r.push_back(a);
  }
  {
    KernelEntry e{"misc/interlacing.glsl", "interlacing",
                  {{"percent", 0.0f, 0.0f, 1.0f, 0.05f, "Interlacing Scanline Bright %"}, {"enable_480i", 1.0f, 0.0f, 1.0f, 1.0f, "Enable 480i Mode"},
                   {"top_field_first", 0.0f, 0.0f, 1.0f, 1.0f, "Top Field First Enable"}},
                  {}, rck::launch_interlacing, setupInterlacing, false};
    e.texture_height_override = true;
    r.push_back(e);
  }
  // ntsc/shaders/ntsc-stock.glsl: the text of stock.glsl (a plain copy, llvmpipe's blit rules included)
  r.push_back({"ntsc/shaders/ntsc-stock.glsl", "ntsc-stock", {}, {}, rck::launch_stock, setupTexCoord, false, true, nullptr, nullptr, true});
  r.push_back({"crt/shaders/crt-potato/shader-files/crt-potato.glsl", "crt-potato", {}, {"MASK"}, rck::launch_crt_potato, setupTexCoord, false});
  // handheld/shaders/sameboy-palettes/: byte-identical copies of motionblur/shaders/response-time.glsl and gb-palette/gb-palette.glsl
  r.push_back({"handheld/shaders/sameboy-palettes/response-time.glsl", "sameboy-response-time", {{"response_time", 0.333f, 0.0f, 0.777f, 0.111f, "LCD Response Time"}},
               {"PrevTexture", "Prev1Texture", "Prev2Texture", "Prev3Texture", "Prev4Texture", "Prev5Texture", "Prev6Texture"},
               rck::launch_response_time, setupResponseTime, false, true, nullptr, nullptr, true});
  r.push_back({"handheld/shaders/sameboy-palettes/gb-palette.glsl", "sameboy-gb-palette", {}, {"COLOR_PALETTE"}, rck::launch_gb_palette, setupTexCoord, false, true,
               nullptr, nullptr, true});
  r.push_back({"handheld/shaders/gb-palette/gb-palette.glsl", "gb-palette", {}, {"COLOR_PALETTE"}, rck::launch_gb_palette, setupTexCoord, false, true,
               nullptr, nullptr, true});
  r.push_back({"reshade/shaders/LUT/LUT.glsl", "reshade-lut", {{"LUT_Size", 16.0f, 1.0f, 64.0f, 1.0f, "LUT Size"}}, {"SamplerLUT"},
               rck::launch_lut, setupTexCoord, false, true, nullptr, nullptr, true});   // reads no size uniform
  r.push_back({"handheld/console-border/shader-files/gb-pass-5.glsl", "gb-pass-5",
               {{"SCALE", 0.6667f, 0.6667f, 1.5f, 0.33333f, "Box Scale"}, {"OUT_X", 1600.0f, 1600.0f, 4800.0f, 8000.0f, "Out X"},
                {"OUT_Y", 800.0f, 800.0f, 2400.0f, 400.0f, "Out Y"}},
               {"BORDER"}, rck::launch_gb_pass_5, setupGbPass5, false});
  r.back().texture_height_override = true;   // setupGbPass5 reads PassGeometry::pass_index
  r.push_back({"handheld/shaders/mgba/agb001.glsl", "agb001", {}, {}, rck::launch_agb001, setupTexCoord, false});
  r.push_back({"handheld/shaders/retro-v2.glsl", "retro-v2", {{"RETRO_PIXEL_SIZE", 0.84f, 0.0f, 1.0f, 0.01f, "Retro Pixel Size"}}, {},
               rck::launch_retro_v2, setupTexCoord, false});
  // handheld/<name>-color.glslp (kernels/pass_basic.hip k_color_matrix); none of them reads a size uniform
  r.push_back({"handheld/shaders/color/gba-color.glsl", "gba-color", {{"darken_screen", 1.0f, -0.25f, 1.0f, 0.05f, "Darken Screen"}}, {},
               rck::launch_color_matrix, setupGbaColor, false, true, nullptr, nullptr, true});
  r.push_back({"handheld/shaders/color/gbc-color.glsl", "gbc-color", {{"lighten_screen", 1.0f, 0.0f, 1.0f, 0.05f, "Lighten Screen"}}, {},
               rck::launch_color_matrix, setupGbcColor, false, true, nullptr, nullptr, true});
  r.push_back({"handheld/shaders/color/vba-color.glsl", "vba-color", {{"darken_screen", 1.0f, -1.0f, 1.0f, 0.05f, "Darken Intensity"}}, {},
               rck::launch_color_matrix, setupVbaColor, false, true, nullptr, nullptr, true});
  r.push_back({"handheld/shaders/color/nds-color.glsl", "nds-color", {}, {}, rck::launch_color_matrix, setupNdsColor, false, true, nullptr, nullptr, true});
  r.push_back({"handheld/shaders/color/palm-color.glsl", "palm-color", {}, {}, rck::launch_color_matrix, setupPalmColor, false, true, nullptr, nullptr, true});
  r.push_back({"handheld/shaders/color/psp-color.glsl", "psp-color", {}, {}, rck::launch_color_matrix, setupPspColor, false, true, nullptr, nullptr, true});
  r.push_back({"handheld/shaders/color/gbc-gambatte-color.glsl", "gbc-gambatte-color", {}, {}, rck::launch_gbc_gambatte_color, setupTexCoord, false, true,
               nullptr, nullptr, true});
  // two more frame-history shaders (kernels/pass_basic.hip): stereoscopic-3d/shutter-to-side-by-side.glslp and misc/anti-flicker.glsl
  r.push_back({"stereoscopic-3d/shaders/shutter-3d.glsl", "shutter-3d",
               {{"ZOOM", 1.0f, 0.0f, 2.0f, 0.01f, "Zoom"}, {"vert_pos", 0.0f, -2.0f, 2.0f, 0.01f, "Vertical Modifier"},
                {"horz_pos", 0.0f, -2.0f, 2.0f, 0.01f, "Horizontal Modifier"}, {"separation", 0.0f, -2.0f, 2.0f, 0.01f, "Eye Separation"},
                {"flicker", 0.0f, 0.0f, 1.0f, 0.25f, "Hold Last Frame (reduce flicker)"}, {"height_mod", 1.0f, 0.0f, 2.0f, 0.01f, "Image Height"},
                {"swap_eye", 0.0f, 0.0f, 1.0f, 1.0f, "Swap Eye Sequence"}},
               {"PrevTexture"}, rck::launch_shutter_3d, setupShutter3d, false, true, nullptr, nullptr, true});   // InputSize / TextureSize only as a ratio
  r.push_back({"misc/anti-flicker.glsl", "anti-flicker", {{"lum_diff_thresh", 0.5f, 0.0f, 1.0f, 0.05f, "Flicker Luma Diff. Threshold"}},
               {"PrevTexture", "Prev1Texture"}, rck::launch_anti_flicker, setupTexCoord, false, true, nullptr, nullptr, true});
  r.push_back({"ntsc/shaders/ntsc-pass1-svideo-3phase.glsl", "ntsc-pass1-svideo-3phase", {}, {},
               rck::launch_ntsc_pass1, setupNtscPass1, false});
  r.push_back({"ntsc/shaders/ntsc-pass2-3phase-gamma.glsl", "ntsc-pass2-3phase-gamma", {}, {},
               rck::launch_ntsc_pass2, setupNtscPass2, true});
  // the rest of the ntsc family (same kernels, other template arguments)
  r.push_back({"ntsc/shaders/ntsc-pass1-composite-3phase.glsl", "ntsc-pass1-composite-3phase", {}, {},
               rck::launch_ntsc_pass1_composite_3phase, setupNtscPass1, false});
  r.push_back({"ntsc/shaders/ntsc-pass1-svideo-2phase.glsl", "ntsc-pass1-svideo-2phase", {}, {},
               rck::launch_ntsc_pass1_svideo_2phase, setupNtscPass1, false});
  r.push_back({"ntsc/shaders/ntsc-pass1-composite-2phase.glsl", "ntsc-pass1-composite-2phase", {}, {},
               rck::launch_ntsc_pass1_composite_2phase, setupNtscPass1, false});
  r.push_back({"ntsc/shaders/ntsc-pass2-3phase-linear.glsl", "ntsc-pass2-3phase-linear", {}, {},
               rck::launch_ntsc_pass2_3phase_linear, setupNtscPass2, true});
  r.push_back({"ntsc/shaders/ntsc-pass2-3phase.glsl", "ntsc-pass2-3phase", {}, {}, rck::launch_ntsc_pass2_3phase_plain,
               setupNtscPass2, true});
  r.push_back({"ntsc/shaders/ntsc-pass2-2phase-gamma.glsl", "ntsc-pass2-2phase-gamma", {}, {},
               rck::launch_ntsc_pass2_2phase_gamma, setupNtscPass2TwoPhase, true});
  r.push_back({"ntsc/shaders/ntsc-pass2-2phase-linear.glsl", "ntsc-pass2-2phase-linear", {}, {},
               rck::launch_ntsc_pass2_2phase_linear, setupNtscPass2TwoPhase, true});
  r.push_back({"ntsc/shaders/ntsc-pass2-2phase.glsl", "ntsc-pass2-2phase", {}, {}, rck::launch_ntsc_pass2_2phase_plain,
               setupNtscPass2TwoPhase, true});
  r.push_back({"xbr/shaders/xbr-lv3.glsl", "xbr-lv3",
               {{"XBR_Y_WEIGHT", 48.0f, 0.0f, 100.0f, 1.0f, "Y Weight"},
                {"XBR_EQ_THRESHOLD", 10.0f, 0.0f, 50.0f, 1.0f, "EQ Threshold"},
                {"XBR_EQ_THRESHOLD2", 2.0f, 0.0f, 4.0f, 1.0f, "EQ Threshold 2"},
                {"XBR_LV2_COEFFICIENT", 2.0f, 1.0f, 3.0f, 1.0f, "Lv2 Coefficient"},
                {"corner_type", 3.0f, 1.0f, 3.0f, 1.0f, "Corner Calculation"}},
               {}, rck::launch_xbr_lv3, setupXbrLv3, true, true, nullptr, scratchXbrLv3});
  {
    // parity "partial": the shader reads an unassigned variable (oracle/rc_passes_ntsc_xbr.c)
    KernelEntry e{"xbr/shaders/xbr-lv2.glsl", "xbr-lv2",
                  {{"XBR_SCALE", 3.0f, 1.0f, 5.0f, 1.0f, "xBR Scale"},   // `//#pragma parameter ...`: the reference's scan is not comment-aware
                   {"XBR_Y_WEIGHT", 48.0f, 0.0f, 100.0f, 1.0f, "Y Weight"},
                   {"XBR_EQ_THRESHOLD", 15.0f, 0.0f, 50.0f, 1.0f, "Eq Threshold"},
                   {"XBR_LV1_COEFFICIENT", 0.5f, 0.0f, 30.0f, 0.5f, "Lv1 Coefficient"},
                   {"XBR_LV2_COEFFICIENT", 2.0f, 1.0f, 3.0f, 0.1f, "Lv2 Coefficient"},
                   {"small_details", 0.0f, 0.0f, 1.0f, 1.0f, "Preserve Small Details"}},
                  {}, rck::launch_xbr_lv2, setupXbrLv2, true};
    r.push_back(e);
  }
  registerRoyaleKernels(r);
  return r;
}

}  // namespace

const std::vector<KernelEntry>& allKernels() {
  static const std::vector<KernelEntry> k = build();
  return k;
}

std::string shaderIdentity(const std::string& shaderPath) {
  const std::string marker = "shaders_glsl/";
  size_t k = shaderPath.rfind(marker);
  if (k != std::string::npos) return shaderPath.substr(k + marker.size());
  // last two components, e.g. ".../shaders/crt-pi.glsl"
  size_t a = shaderPath.find_last_of('/');
  if (a == std::string::npos) return shaderPath;
  size_t b = a == 0 ? std::string::npos : shaderPath.find_last_of('/', a - 1);
  return b == std::string::npos ? shaderPath : shaderPath.substr(b + 1);
}

const KernelEntry* findKernel(const std::string& shaderPath) {
  const std::string id = shaderIdentity(shaderPath);
  const KernelEntry* tail_match = nullptr;
  for (const KernelEntry& e : allKernels()) {
    const std::string want = e.identity;
    if (id == want) return &e;
    // identity given relative to something deeper (e.g. "shaders/crt-pi.glsl" for
    // "crt/shaders/crt-pi.glsl"): accept a suffix match on whole path components
    if (want.size() > id.size() && want.compare(want.size() - id.size(), id.size(), id) == 0 &&
        want[want.size() - id.size() - 1] == '/')
      tail_match = &e;
    if (id.size() > want.size() && id.compare(id.size() - want.size(), want.size(), want) == 0 &&
        id[id.size() - want.size() - 1] == '/')
      tail_match = &e;
  }
  return tail_match;
}

}  // namespace rc
