/* TEST INFRASTRUCTURE - see rc_oracle.h.
 *
 * crt/crt-hyllian-glow.glslp (the reference's smoke-test default preset, tools/smoke-test.sh), 6 passes:
 *   P0 crt/shaders/glow/linearize.glsl                                   FS 88-93
 *   P1 crt/shaders/hyllian/crt-hyllian-glow/crt-hyllian-glow.glsl        FS 146-247
 *   P2 crt/shaders/glow/threshold.glsl                                   FS 90-96
 *   P3 crt/shaders/glow/blur_horiz.glsl   (mipmap_input, 1/4 size)       FS 84-98
 *   P4 crt/shaders/glow/blur_vert.glsl                                   FS 84-98
 *   P5 crt/shaders/hyllian/crt-hyllian-glow/resolve2.glsl                FS 129-189, 408-424
 * Operation order as Mesa's compiler leaves it, pinned by float-precision goldens (tests/golden/f32_*).
 */
#include <math.h>

#include "rc_oracle.h"

static inline float minps(float a, float b) { return a < b ? a : b; } /* SSE minps: NaN -> b */
static inline float maxps(float a, float b) { return a > b ? a : b; }
static inline float clampf(float x, float lo, float hi) { return minps(maxps(x, lo), hi); }

#define ENTER unsigned csr_ = o_fp_enter()
#define LEAVE o_fp_leave(csr_)

/* P0: params INPUT_GAMMA */
void o_pass_glow_linearize(const o_pass_args* a) {
  ENTER;
  const int W = a->out_w, H = a->out_h;
  const float g = a->params[0];
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const o_vec4 c = o_sample(a->in, o_varying_at(&tu, x, y, lo), o_varying_at(&tv, x, y, lo));
      const o_vec4 o = {o_pow(c.x, g), o_pow(c.y, g), o_pow(c.z, g), 1.0f};
      o_store_pixel(a, x, y, o);
    }
  LEAVE;
}

/* P2: params GLOW_WHITEPOINT, GLOW_ROLLOFF */
void o_pass_glow_threshold(const o_pass_args* a) {
  ENTER;
  const int W = a->out_w, H = a->out_h;
  const float wp = a->params[0], roll = a->params[1];
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const o_vec4 c = o_sample(a->in, o_varying_at(&tu, x, y, lo), o_varying_at(&tv, x, y, lo));
      const float in3[3] = {c.x, c.y, c.z};
      float out[3];
      for (int k = 0; k < 3; ++k) out[k] = o_pow(clampf((1.15f * in3[k]) / wp, 0.0f, 1.0f), roll);
      const o_vec4 o = {out[0], out[1], out[2], 1.0f};
      o_store_pixel(a, x, y, o);
    }
  LEAVE;
}

/* P3 / P4: 9-tap Gaussian, weights exp(-0.35 i^2) folded at compile time; P3 steps 4 texels and samples a
 * mip-mapped input (GL_LINEAR_MIPMAP_LINEAR), P4 steps 1 texel vertically */
static void glow_blur(const o_pass_args* a, int horizontal) {
  const int W = a->out_w, H = a->out_h;
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  const float step = horizontal ? 4.0f * (1.0f / (float)a->in->w) : 1.0f / (float)a->in->h;
  float k[9], k_total = 0.0f;
  for (int i = -4; i <= 4; ++i) {
    const float fi = (float)i;
    /* the loop is unrolled and exp(-0.35 * i * i) of a constant i is folded at compile time - after the lowering
     * exp(x) -> exp2(x * log2e) and the re-association that moves the constant factor onto one operand:
     * exp2f(i * (i * (-0.35f * log2e))) in float (pinned with a float probe of blur_vert on exact texels: bit-identical;
     * a correctly rounded exp() differs in the last bit of two of the nine weights) */
    k[i + 4] = exp2f(fi * (fi * (-0.35f * 1.4426950408889634f)));
    k_total += k[i + 4];
  }
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const int x0 = x & ~1, y0 = y & ~1;
      const float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      /* the quad's coordinates as this pixel's triangle extrapolates them */
      const float ux0 = o_varying_at(&tu, x0, y, lo), ux1 = o_varying_at(&tu, x0 + 1, y, lo);
      const float vx0 = o_varying_at(&tv, x0, y, lo), vx1 = o_varying_at(&tv, x0 + 1, y, lo);
      const float uy0 = o_varying_at(&tu, x, y0, lo), uy1 = o_varying_at(&tu, x, y0 + 1, lo);
      const float vy0 = o_varying_at(&tv, x, y0, lo), vy1 = o_varying_at(&tv, x, y0 + 1, lo);
      float col[3] = {0.0f, 0.0f, 0.0f};
      for (int i = -4; i <= 4; ++i) {
        const float off = (float)i * step;
        o_vec4 c;
        if (horizontal)
          c = o_sample_quad(a->in, u + off, v + 0.0f, ux0 + off, ux1 + off, vx0 + 0.0f, vx1 + 0.0f, uy0 + off, uy1 + off, vy0 + 0.0f, vy1 + 0.0f);
        else
          c = o_sample_quad(a->in, u + 0.0f, v + off, ux0 + 0.0f, ux1 + 0.0f, vx0 + off, vx1 + off, uy0 + 0.0f, uy1 + 0.0f, vy0 + off, vy1 + off);
        col[0] += k[i + 4] * c.x;
        col[1] += k[i + 4] * c.y;
        col[2] += k[i + 4] * c.z;
      }
      const o_vec4 o = {col[0] / k_total, col[1] / k_total, col[2] / k_total, 1.0f};
      o_store_pixel(a, x, y, o);
    }
}
void o_pass_glow_blur_h(const o_pass_args* a) { ENTER; glow_blur(a, 1); LEAVE; }
void o_pass_glow_blur_v(const o_pass_args* a) { ENTER; glow_blur(a, 0); LEAVE; }

/* P1: params BEAM_PROFILE, BEAM_MIN_WIDTH, BEAM_MAX_WIDTH, SCANLINES_STRENGTH, COLOR_BOOST, HFILTER_SHARPNESS,
 * CRT_ANTI_RINGING, InputGamma, OutputGamma, VSCANLINES */
void o_pass_crt_hyllian_glow(const o_pass_args* a) {
  ENTER;
  const int W = a->out_w, H = a->out_h;
  const float* P = a->params;
  float bp[4] = {P[3], P[1], P[2], P[4]}; /* SCANLINES_STRENGTH, BEAM_MIN_WIDTH, BEAM_MAX_WIDTH, COLOR_BOOST */
  static const float prof[6][4] = {{0.40f, 1.00f, 1.00f, 1.00f}, {0.72f, 1.00f, 1.00f, 1.25f}, {0.60f, 0.50f, 1.00f, 1.25f},
                                   {0.60f, 0.72f, 1.00f, 1.25f}, {0.68f, 0.68f, 1.00f, 1.25f}, {0.70f, 0.50f, 1.00f, 1.80f}};
  for (int k = 1; k <= 6; ++k)
    if (P[0] == (float)k) for (int c = 0; c < 4; ++c) bp[c] = prof[k - 1][c];
  const float sharp = P[5], anti = P[6], gin = P[7], gout = P[8], vs = P[9];
  const float B = 1.0f - sharp, C = sharp * 0.5f;
  /* invX (columns as written, GLSL mat4 is column-major: invX[c][r]) */
  const float m[4][4] = {{(-B - 6.0f * C) / 6.0f, (12.0f - 9.0f * B - 6.0f * C) / 6.0f, -(12.0f - 9.0f * B - 6.0f * C) / 6.0f, (B + 6.0f * C) / 6.0f},
                         {(3.0f * B + 12.0f * C) / 6.0f, (-18.0f + 12.0f * B + 6.0f * C) / 6.0f, (18.0f - 15.0f * B - 12.0f * C) / 6.0f, -C},
                         {(-3.0f * B - 6.0f * C) / 6.0f, 0.0f, (3.0f * B + 6.0f * C) / 6.0f, 0.0f},
                         {B / 6.0f, (6.0f - 2.0f * B) / 6.0f, B / 6.0f, 0.0f}};
  const float tsx = (float)a->in->w, tsy = (float)a->in->h;
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  /* mix(a, b, t) with a run-time t: x + t*(y - x) */
#define MIXRT(a_, b_, t_) ((a_) + (t_) * ((b_) - (a_)))
  const float dxx = MIXRT(1.0f / tsx, 0.0f, vs), dxy = MIXRT(0.0f, 1.0f / tsy, vs);
  const float dyx = MIXRT(0.0f, 1.0f / tsx, vs), dyy = MIXRT(1.0f / tsy, 0.0f, vs);
  const float scan = 4.0f * bp[0], bmin = bp[1], bmax = bp[2], boost = bp[3];
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      const float pcx = u * tsx + -0.5f, pcy = v * tsy + 0.5f;
      const float flx = floorf(pcx), fly = floorf(pcy);
      const float tcx = MIXRT((flx + 0.5f) / tsx, (flx + 1.0f) / tsx, vs);
      const float tcy = MIXRT((fly + 0.5f) / tsy, (fly + -0.5f) / tsy, vs);
      const float frx = pcx - flx, fry = pcy - fly;
      const float fpx = MIXRT(frx, fry, vs), fpy = MIXRT(fry, frx, vs);
      o_vec4 c[2][4];
      for (int r = 0; r < 2; ++r)
        for (int k = 0; k < 4; ++k) {
          const float kx = (float)(k - 1);
          float su = tcx + kx * dxx, sv = tcy + kx * dxy;
          if (r == 0) { su -= dyx; sv -= dyy; }
          o_vec4 t = o_sample(a->in, su, sv);
          c[r][k].x = o_pow(t.x, gin); c[r][k].y = o_pow(t.y, gin); c[r][k].z = o_pow(t.z, gin); c[r][k].w = o_pow(t.w, gin);
        }
      const float lobes[4] = {fpx * fpx * fpx, fpx * fpx, fpx, 1.0f};
      float ip[4]; /* invX * lobes: columns accumulated left to right */
      for (int r = 0; r < 4; ++r) ip[r] = ((m[0][r] * lobes[0] + m[1][r] * lobes[1]) + m[2][r] * lobes[2]) + m[3][r] * lobes[3];
      /* row 1 has a literal 0 in column 2 and lobes[3] is the literal 1: what is left is two products and a plain
       * addend, and the addend joins the inner product (in-situ float probe, bit-identical) */
      ip[1] = m[1][1] * lobes[1] + (m[0][1] * lobes[0] + m[3][1]);
      float col[2][4];
      for (int r = 0; r < 2; ++r) {
        const float* q0 = &c[r][0].x; const float* q1 = &c[r][1].x; const float* q2 = &c[r][2].x; const float* q3 = &c[r][3].x;
        for (int ch = 0; ch < 4; ++ch) {
          float v0 = ((q0[ch] * ip[0] + q1[ch] * ip[1]) + q2[ch] * ip[2]) + q3[ch] * ip[3];
          const float mn = minps(q1[ch], q2[ch]), mx = maxps(q1[ch], q2[ch]);
          const float cl = clampf(v0, mn, mx);
          col[r][ch] = MIXRT(v0, cl, anti);
        }
      }
      const float pos0 = fpy, pos1 = 1.0f - fpy;
      float out[4];
      for (int ch = 0; ch < 4; ++ch) {
        const float lum0 = MIXRT(bmin, bmax, col[0][ch]);
        const float lum1 = MIXRT(bmin, bmax, col[1][ch]);
        float d0 = (scan * pos0) / (lum0 + 0.0000001f), d1 = (scan * pos1) / (lum1 + 0.0000001f);
        /* exp(-d*d) = exp2(((-d) * d) * log2e) with the constant moved onto one factor: exp2(d * (d * -log2e))
         * (in-situ float probe, bit-identical) */
        d0 = o_exp2(d0 * (d0 * -1.4426950408889634f));
        d1 = o_exp2(d1 * (d1 * -1.4426950408889634f));
        const float cc = boost * (col[0][ch] * d0 + col[1][ch] * d1);
        out[ch] = o_pow(cc, 1.0f / gout);
      }
      const o_vec4 o = {out[0], out[1], out[2], out[3]};
      o_store_pixel(a, x, y, o);
    }
  LEAVE;
}

/* resolve2.glsl mask_weights (129-400), layouts 3 and 6..19: the colour of cell [w][z], w = floor(mod(coord.y, ny)),
 * z = floor(mod(coord.x, nx)) with coord = gl_FragCoord (pixel + 0.5: the float mod is the integer one, the quotient
 * never comes within 0.03 of an integer).  The shader's ternary chains test w == 1, w == 2, ... and leave the LAST row
 * to w == 0; rows below are indexed by w.  R G B M(agenta) Y(ellow) C(yan) K(black): a channel is 1 where the colour
 * has it and 1 - MASK_INTENSITY elsewhere.  Layout 12 reads w without ever writing it (`int w, z = 0;`): the GL's
 * compiler resolves the undefined comparison to false - its second table - on every row (golden). */
typedef struct { int nx, ny; const char* rows[6]; } mask_layout;
static const mask_layout k_mask_layouts[20] = {
    [3] = {4, 3, {"KKMG", "MGKK", "MGMG"}},
    [6] = {4, 1, {"RGBK"}},
    [7] = {5, 1, {"RMBGG"}},
    [8] = {7, 1, {"RRYGCBB"}},
    [9] = {4, 1, {"RYCB"}},
    [10] = {4, 1, {"RMCG"}},
    [11] = {4, 2, {"BKRG", "RGBK"}},
    [12] = {4, 1, {"CBRY"}},
    [13] = {4, 4, {"CBRY", "RYCB", "RYCB", "CBRY"}},
    [14] = {6, 3, {"KKKMGK", "MGKKKK", "MGKMGK"}},
    [15] = {8, 4, {"KKKKRYCB", "RYCBRYCB", "RYCBKKKK", "RYCBRYCB"}},
    [16] = {4, 3, {"KKYB", "YBKK", "YBYB"}},
    [17] = {10, 4, {"RRKKKKBBGG", "RMBGGRMBGG", "KBBGGRRKKK", "RMBGGRMBGG"}},
    [18] = {10, 4, {"RRKKKKGGBB", "RYGBBRYGBB", "KGGBBRRKKK", "RYGBBRYGBB"}},
    [19] = {14, 6, {"KKKKKKKKRRYGCB", "RRYGCBBRRYGCBB", "RRYGCBBRRYGCBB", "RRYGCBBKKKKKKK", "RRYGCBBRRYGCBB", "RRYGCBBRRYGCBB"}},
};
static int mask_bits(char c) {   /* bit 0 red, 1 green, 2 blue */
  switch (c) {
    case 'R': return 1; case 'G': return 2; case 'B': return 4; case 'M': return 5; case 'Y': return 3; case 'C': return 6;
    default: return 0;
  }
}

/* P5: params BLOOM_STRENGTH, OUTPUT_GAMMA, PHOSPHOR_LAYOUT, MASK_INTENSITY; extra[0] = PassPrev4Texture. */
void o_pass_hyllian_resolve2(const o_pass_args* a) {
  ENTER;
  const int W = a->out_w, H = a->out_h;
  const float strength = a->params[0], gamma = a->params[1], intensity = a->params[3];
  const int layout = (int)a->params[2];
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  const float on = 1.0f, off = 1.0f - intensity;
  const float magenta[3] = {on, off, on}, green[3] = {off, on, off}, yellow[3] = {on, on, off}, blue[3] = {off, off, on};
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      const o_vec4 s = o_sample(a->extra[0], u, v), b = o_sample(a->in, u, v);
      const float fx = (float)x + 0.5f, fy = (float)y + 0.5f;
      const float mx = floorf(fx - 2.0f * floorf(fx / 2.0f)), my = floorf(fy - 2.0f * floorf(fy / 2.0f));
      float w3[3] = {1.0f, 1.0f, 1.0f};
      const float* c0 = 0; const float* c1 = 0;
      if (layout == 1 || layout == 2) { c0 = magenta; c1 = green; }
      if (layout == 4 || layout == 5) { c0 = yellow; c1 = blue; }
      if (c0) {
        for (int k = 0; k < 3; ++k) {
          const float ap = c0[k] + mx * (c1[k] - c0[k]);
          if (layout == 2 || layout == 5) {
            const float inv = c1[k] + mx * (c0[k] - c1[k]);
            w3[k] = ap + my * (inv - ap);
          } else w3[k] = ap;
        }
      } else if (layout >= 3 && layout <= 19 && k_mask_layouts[layout].nx) {
        const mask_layout* m = &k_mask_layouts[layout];
        const int bits = mask_bits(m->rows[y % m->ny][x % m->nx]);
        for (int k = 0; k < 3; ++k) w3[k] = ((bits >> k) & 1) ? on : off;
      }
      const float s3[3] = {s.x, s.y, s.z}, b3[3] = {b.x, b.y, b.z};
      float out[3];
      for (int k = 0; k < 3; ++k) {
        float src = 1.15f * s3[k];
        src = src + strength * b3[k];
        src = src * w3[k];
        out[k] = o_pow(clampf(src, 0.0f, 1.0f), 1.0f / gamma);
      }
      const o_vec4 o = {out[0], out[1], out[2], 1.0f};
      o_store_pixel(a, x, y, o);
    }
  LEAVE;
}
