/* TEST INFRASTRUCTURE - see rc_oracle.h.
 *
 * Hand restatement of the arithmetic of three more reference shader assets:
 *   ntsc pass 1  shaders/shaders_glsl/ntsc/shaders/ntsc-pass1-svideo-3phase.glsl (VS 64-70, FS 147-166)
 *   ntsc pass 2  shaders/shaders_glsl/ntsc/shaders/ntsc-pass2-3phase-gamma.glsl  (VS 44-49, FS 215-281;
 *                the file keeps "#version 130", so the unrolled macro_loopz branch is the one compiled)
 *   xbr-lv3      shaders/shaders_glsl/xbr/shaders/xbr-lv3.glsl                   (VS 77-100, FS 171-352)
 * Operation order follows what Mesa's GLSL->NIR lowering produces (measured with
 * oracle/_ref/glprobe): vec*mat is a dot per column evaluated x*c0 + (y*c1 + z*c2);
 * mat*vec accumulates columns left to right; mod(x,c) = x - c*floor(x/c) with a true
 * division; smoothstep with constant edges divides by the folded (e1-e0); nothing is fused.
 */
#include <math.h>

#include "rc_oracle.h"

/* ------------------------------------------------------------------------- ntsc pass 1 -- */
/* The four pass-1 files differ only in two #defines (lines 3-4): COMPOSITE / SVIDEO select the
 * cross-talk matrix mix_mat (lines 15-25, 118-124), TWO_PHASE / THREE_PHASE the chroma phase
 * (lines 150-155) and CHROMA_MOD_FREQ (lines 9-13). */
static void ntsc_pass1_body(const o_pass_args* a, int composite, int two_phase) {
  const int W = a->out_w, H = a->out_h;
  const float tsx = (float)a->in->w, tsy = (float)a->in->h; /* TextureSize == InputSize */
  /* VS :69  pix_no = vTexCoord * SourceSize.xy * (outsize.xy / InputSize.xy) */
  const float px1 = (1.0f * tsx) * ((float)W / tsx), py1 = (1.0f * tsy) * ((float)H / tsy);
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt);
  o_varying tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  o_varying pu = o_varying_setup(0.f, px1, px1, 0.f, W, H, a->out_fmt);
  o_varying pv = o_varying_setup(0.f, 0.f, py1, py1, W, H, a->out_fmt);
  const float pi = 3.14159265f;
  const float k_phase = two_phase ? pi : 0.6667f * pi;              /* folded constants */
  const float k_freq = two_phase ? (4.0f * pi) / 15.0f : pi / 3.0f; /* CHROMA_MOD_FREQ */
  const float period = two_phase ? 2.0f : 3.0f;
  const float fc = (float)a->frame_count;
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      int lo = o_lower_tri(x, y, W, H);
      float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      float pnx = o_varying_at(&pu, x, y, lo), pny = o_varying_at(&pv, x, y, lo);
      o_vec4 col = o_sample(a->in, u, v);
      /* rgb2yiq: col * yiq_mat */
      float yy = col.x * 0.2989f + (col.y * 0.5870f + col.z * 0.1140f);
      float ii = col.x * 0.5959f + (col.y * -0.2744f + col.z * -0.3216f);
      float qq = col.x * 0.2115f + (col.y * -0.5229f + col.z * 0.3114f);
      float m = pny - period * floorf(pny / period);
      float chroma_phase = k_phase * (m + fc);
      float mod_phase = chroma_phase + pnx * k_freq;
      float i_mod = o_cos(mod_phase), q_mod = o_sin(mod_phase);
      ii *= i_mod; qq *= q_mod;           /* modulate */
      /* yiq *= mix_mat: a dot per column, x*c0 + (y*c1 + z*c2); the factors 1, 2 and 0 are exact */
      float my, mi, mq;
      /* (composite, first column (1,1,1): ii and qq are products, so the plain addend yy joins the
       * pending multiply-add's addend first: (yy + ii) + qq, float goldens) */
      if (composite) { my = (yy + qq) + ii; mi = yy + (ii * 2.0f + 0.0f); mq = yy + (0.0f + qq * 2.0f); }
      else { my = yy; mi = ii * 2.0f; mq = qq * 2.0f; }
      mi *= i_mod; mq *= q_mod;           /* demodulate */
      o_vec4 o = {my, mi, mq, 1.0f};
      o_store_pixel(a, x, y, o);
    }
}
#define NTSC_P1(NAME, C, T)                              \
  void o_pass_ntsc_pass1_##NAME(const o_pass_args* a) {  \
    unsigned csr = o_fp_enter();                         \
    ntsc_pass1_body(a, C, T);                            \
    o_fp_leave(csr);                                     \
  }
NTSC_P1(svideo_3phase, 0, 0)
NTSC_P1(composite_3phase, 1, 0)
NTSC_P1(svideo_2phase, 0, 1)
NTSC_P1(composite_2phase, 1, 1)

/* ------------------------------------------------------------------------- ntsc pass 2 -- */
/* ntsc-pass2-{2,3}phase{,-gamma,-linear}.glsl: symmetric FIR of 2*TAPS+1 taps (3-phase: TAPS 24,
 * tables :87-137; 2-phase: TAPS 32, tables :116-182), then YIQ->RGB and pow(rgb, 2.5/2.0) (-gamma),
 * pow(rgb, 2.4) (-linear) or nothing. */
static const float k_luma3[25] = {
    -0.000012020f, -0.000022146f, -0.000013155f, -0.000012020f, -0.000049979f, -0.000113940f, -0.000122150f,
    -0.000005612f, 0.000170516f,  0.000237199f,  0.000169640f,  0.000285688f,  0.000984574f,  0.002018683f,
    0.002002275f,  -0.000909882f, -0.007049081f, -0.013222860f, -0.012606931f, 0.002460860f,  0.035868225f,
    0.084016453f,  0.135563500f,  0.175261268f,  0.190176552f};
static const float k_chroma3[25] = {
    -0.000118847f, -0.000271306f, -0.000502642f, -0.000930833f, -0.001451013f, -0.002064744f, -0.002700432f,
    -0.003241276f, -0.003524948f, -0.003350284f, -0.002491729f, -0.000721149f, 0.002164659f,  0.006313635f,
    0.011789103f,  0.018545660f,  0.026414396f,  0.035100710f,  0.044196567f,  0.053207202f,  0.061590275f,
    0.068803602f,  0.074356193f,  0.077856564f,  0.079052396f};
static const float k_luma2[33] = {
    -0.000174844f, -0.000205844f, -0.000149453f, -0.000051693f, 0.000000000f,  -0.000066171f, -0.000245058f,
    -0.000432928f, -0.000472644f, -0.000252236f, 0.000198929f,  0.000687058f,  0.000944112f,  0.000803467f,
    0.000363199f,  0.000013422f,  0.000253402f,  0.001339461f,  0.002932972f,  0.003983485f,  0.003026683f,
    -0.001102056f, -0.008373026f, -0.016897700f, -0.022914480f, -0.021642347f, -0.008863273f, 0.017271957f,
    0.054921920f,  0.098342579f,  0.139044281f,  0.168055832f,  0.178571429f};
static const float k_chroma2[33] = {
    0.001384762f, 0.001678312f, 0.002021715f, 0.002420562f, 0.002880460f, 0.003406879f, 0.004004985f,
    0.004679445f, 0.005434218f, 0.006272332f, 0.007195654f, 0.008204665f, 0.009298238f, 0.010473450f,
    0.011725413f, 0.013047155f, 0.014429548f, 0.015861306f, 0.017329037f, 0.018817382f, 0.020309220f,
    0.021785952f, 0.023227857f, 0.024614500f, 0.025925203f, 0.027139546f, 0.028237893f, 0.029201910f,
    0.030015081f, 0.030663170f, 0.031134640f, 0.031420995f, 0.031517031f};

/* epilogue: 0 plain, 1 gamma (2.5/2.0), 2 linear (2.4) */
static void ntsc_pass2_body(const o_pass_args* a, int taps, const float* luma, const float* chroma, int epilogue) {
  const int W = a->out_w, H = a->out_h;
  const float tsx = (float)a->in->w;
  /* VS :48  TEX0.xy = TexCoord.xy - vec2(0.5 / SourceSize.x, 0.0) */
  const float sh = 0.5f / tsx;
  o_varying tu = o_varying_setup(0.f - sh, 1.f - sh, 1.f - sh, 0.f - sh, W, H, a->out_fmt);
  o_varying tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  const float one_x = 1.0f / tsx;
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      int lo = o_lower_tri(x, y, W, H);
      float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      float sy = 0.f, si = 0.f, sq = 0.f;
      for (int c = 1; c <= taps; ++c) {
        float off = (float)(c - 1 - taps);
        o_vec4 p = o_sample(a->in, u + off * one_x, v);
        o_vec4 n = o_sample(a->in, u + (-off) * one_x, v);
        sy = sy + (p.x + n.x) * luma[c - 1];
        si = si + (p.y + n.y) * chroma[c - 1];
        sq = sq + (p.z + n.z) * chroma[c - 1];
      }
      o_vec4 m = o_sample(a->in, u, v);
      sy = sy + m.x * luma[taps];
      si = si + m.y * chroma[taps];
      sq = sq + m.z * chroma[taps];
      /* yiq2rgb: signal * yiq2rgb_mat = a dot per column, x*1.0 + (y*c1 + z*c2).  With the first
       * factor 1.0 the plain addend x meets the pending multiply-add y*c1 + (z*c2) and is added to its
       * addend first (see rc_passes_royale.c blur9): (x + z*c2) + y*c1 (float goldens) */
      float r = (sy + sq * 0.6210f) + si * 0.956f;
      float g = (sy + sq * -0.6474f) + si * -0.2720f;
      float b = (sy + sq * 1.7046f) + si * -1.1060f;
      o_vec4 o = {r, g, b, 1.0f};
      if (epilogue) {
        const float gm = epilogue == 1 ? 2.5f / 2.0f : 2.4f;
        o.x = o_pow(r, gm); o.y = o_pow(g, gm); o.z = o_pow(b, gm);
      }
      o_store_pixel(a, x, y, o);
    }
}
#define NTSC_P2(NAME, TAPS, L, C, E)                     \
  void o_pass_ntsc_pass2_##NAME(const o_pass_args* a) {  \
    unsigned csr = o_fp_enter();                         \
    ntsc_pass2_body(a, TAPS, L, C, E);                   \
    o_fp_leave(csr);                                     \
  }
NTSC_P2(3phase_gamma, 24, k_luma3, k_chroma3, 1)
NTSC_P2(3phase_linear, 24, k_luma3, k_chroma3, 2)
NTSC_P2(3phase, 24, k_luma3, k_chroma3, 0)
NTSC_P2(2phase_gamma, 32, k_luma2, k_chroma2, 1)
NTSC_P2(2phase_linear, 32, k_luma2, k_chroma2, 2)
NTSC_P2(2phase, 32, k_luma2, k_chroma2, 0)

/* ----------------------------------------------------------------------------- xbr-lv3 -- */
typedef struct { float v[4]; } f4;
typedef struct { int v[4]; } b4;

static inline f4 f4_df(f4 A, f4 B) { f4 r; for (int k = 0; k < 4; ++k) r.v[k] = fabsf(A.v[k] - B.v[k]); return r; }
static inline b4 b4_le(f4 A, float t) { b4 r; for (int k = 0; k < 4; ++k) r.v[k] = A.v[k] <= t; return r; }
static inline b4 b4_lt(f4 A, float t) { b4 r; for (int k = 0; k < 4; ++k) r.v[k] = A.v[k] < t; return r; }
static inline b4 b4_and(b4 A, b4 B) { b4 r; for (int k = 0; k < 4; ++k) r.v[k] = A.v[k] && B.v[k]; return r; }
static inline b4 b4_or(b4 A, b4 B) { b4 r; for (int k = 0; k < 4; ++k) r.v[k] = A.v[k] || B.v[k]; return r; }
static inline b4 b4_not(b4 A) { b4 r; for (int k = 0; k < 4; ++k) r.v[k] = !A.v[k]; return r; }
static inline b4 b4_ne(f4 A, f4 B) { b4 r; for (int k = 0; k < 4; ++k) r.v[k] = A.v[k] != B.v[k]; return r; }
static inline f4 sw(f4 A, int i0, int i1, int i2, int i3) { f4 r = {{A.v[i0], A.v[i1], A.v[i2], A.v[i3]}}; return r; }
#define YZWX(A) sw(A, 1, 2, 3, 0)
#define WXYZ(A) sw(A, 3, 0, 1, 2)
#define ZWXY(A) sw(A, 2, 3, 0, 1)

/* transpose(mat4x3(P0,P1,P2,P3)) * w  (FS :230-246) */
static inline f4 lum4(o_vec4 p0, o_vec4 p1, o_vec4 p2, o_vec4 p3, const float* w) {
  f4 r;
  r.v[0] = (p0.x * w[0] + p0.y * w[1]) + p0.z * w[2];
  r.v[1] = (p1.x * w[0] + p1.y * w[1]) + p1.z * w[2];
  r.v[2] = (p2.x * w[0] + p2.y * w[1]) + p2.z * w[2];
  r.v[3] = (p3.x * w[0] + p3.y * w[1]) + p3.z * w[2];
  return r;
}
/* weighted_distance (FS :166-169).  The GLSL compiler rebalances the five-term sum into
 * ((ab + ac) + (de + df)) + 4*gh (the only association of all 5-leaf trees that matches llvmpipe
 * on a noise image; measured). */
static inline f4 wd(f4 a, f4 b, f4 c, f4 d, f4 e, f4 f, f4 g, f4 h) {
  f4 ab = f4_df(a, b), ac = f4_df(a, c), de = f4_df(d, e), dff = f4_df(d, f), gh = f4_df(g, h), r;
  for (int k = 0; k < 4; ++k) r.v[k] = ((ab.v[k] + ac.v[k]) + (de.v[k] + dff.v[k])) + 4.0f * gh.v[k];
  return r;
}
/* smoothstep(C - delta, C + delta, A*fp.y + B*fp.x) (FS :267-271, :302-306).  The edges fold to
 * constants and the compiler moves the additive constant inward: the numerator is evaluated as
 * (A*fy - e0) + B*fx (float probes taken inside the complete shader, every component of the five calls), except where
 * A = B = +-1 (fx45.x and fx45.z): there the sum is plain, (fy + fx) or -(fy + fx), and keeps the outer subtraction. */
static inline f4 line_sstep(const float* A, const float* B, const float* C, float fy, float fx) {
  f4 r;
  for (int k = 0; k < 4; ++k) {
    const float e0 = C[k] - 0.4f, e1 = C[k] + 0.4f;
    float num;
    if ((A[k] == -1.0f && B[k] == -1.0f) || (A[k] == 1.0f && B[k] == 1.0f)) num = (A[k] * fy + B[k] * fx) - e0;
    else num = (A[k] * fy - e0) + B[k] * fx;
    float t = num / (e1 - e0);
    t = t > 0.0f ? t : 0.0f;
    t = t < 1.0f ? t : 1.0f;
    r.v[k] = t * (t * (3.0f - 2.0f * t));
  }
  return r;
}
static inline o_vec4 mix3(o_vec4 a, o_vec4 b, float t) {
  o_vec4 r = {a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z), 1.0f};
  return r;
}
static inline float c_df(o_vec4 a, o_vec4 b) {
  return (fabsf(a.x - b.x) + fabsf(a.y - b.y)) + fabsf(a.z - b.z);
}

/* params: XBR_Y_WEIGHT, XBR_EQ_THRESHOLD, XBR_EQ_THRESHOLD2, XBR_LV2_COEFFICIENT, corner_type */
static void xbr_lv3_body(const o_pass_args* a) {
  const int W = a->out_w, H = a->out_h;
  const float yw = a->params[0], thr = a->params[1], thr2 = a->params[2], lv2 = a->params[3], corner = a->params[4];
  const float tsx = (float)a->in->w, tsy = (float)a->in->h;
  /* VS :82-99 */
  const float dx = 1.0f / tsx, dy = 1.0f / tsy;
  const float xoff[5] = {-2.0f * dx, -dx, 0.0f, dx, 2.0f * dx};
  const float yoff[5] = {-2.0f * dy, -dy, 0.0f, dy, 2.0f * dy};
  o_varying vx[5], vy[5];
  for (int k = 0; k < 5; ++k) {
    vx[k] = o_varying_setup(0.f + xoff[k], 1.f + xoff[k], 1.f + xoff[k], 0.f + xoff[k], W, H, a->out_fmt);
    vy[k] = o_varying_setup(0.f + yoff[k], 0.f + yoff[k], 1.f + yoff[k], 1.f + yoff[k], W, H, a->out_fmt);
  }
  const float w[3] = {yw * 0.299f, yw * 0.587f, yw * 0.114f};
  static const float Ao[4] = {1.0f, -1.0f, -1.0f, 1.0f}, Bo[4] = {1.0f, 1.0f, -1.0f, -1.0f}, Co[4] = {1.5f, 0.5f, -0.5f, 0.5f};
  static const float Bx[4] = {0.5f, 2.0f, -0.5f, -2.0f}, Cx[4] = {1.0f, 1.0f, -0.5f, 0.0f};
  static const float By[4] = {2.0f, 0.5f, -2.0f, -0.5f}, Cy[4] = {2.0f, 0.0f, -1.0f, 0.5f};
  static const float Az[4] = {6.0f, -2.0f, -6.0f, 2.0f}, Bz[4] = {2.0f, 6.0f, -2.0f, -6.0f}, Cz[4] = {5.0f, 3.0f, -3.0f, -1.0f};
  static const float Aw[4] = {2.0f, -6.0f, -2.0f, 6.0f}, Bw[4] = {6.0f, 2.0f, -6.0f, -2.0f}, Cw[4] = {5.0f, -1.0f, -3.0f, 3.0f};
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      int lo = o_lower_tri(x, y, W, H);
      float cx[5], cy[5];
      for (int k = 0; k < 5; ++k) {
        cx[k] = o_varying_at(&vx[k], x, y, lo);
        cy[k] = o_varying_at(&vy[k], x, y, lo);
      }
      float fpx = cx[2] * tsx, fpy = cy[2] * tsy;
      fpx = fpx - floorf(fpx);
      fpy = fpy - floorf(fpy);
#define T(i, j) o_sample(a->in, cx[i], cy[j])
      o_vec4 A1 = T(1, 0), B1 = T(2, 0), C1 = T(3, 0);
      o_vec4 A = T(1, 1), B = T(2, 1), C = T(3, 1);
      o_vec4 D = T(1, 2), E = T(2, 2), F = T(3, 2);
      o_vec4 G = T(1, 3), Hh = T(2, 3), I = T(3, 3);
      o_vec4 G5 = T(1, 4), H5 = T(2, 4), I5 = T(3, 4);
      o_vec4 A0 = T(0, 1), D0 = T(0, 2), G0 = T(0, 3);
      o_vec4 C4 = T(4, 1), F4 = T(4, 2), I4 = T(4, 3);
#undef T
      f4 b = lum4(B, D, Hh, F, w), c = lum4(C, A, G, I, w), e = lum4(E, E, E, E, w);
      f4 d = YZWX(b), f = WXYZ(b), g = ZWXY(c), h = ZWXY(b), i = WXYZ(c);
      f4 i4 = lum4(I4, C1, A0, G5, w), i5 = lum4(I5, C4, A1, G0, w), h5 = lum4(H5, F4, B1, D0, w);
      f4 f4_ = YZWX(h5), c1 = YZWX(i4), g0 = WXYZ(i5), b1 = ZWXY(h5), d0 = WXYZ(h5);
      b4 r1;
      b4 ne_ef_eh = b4_and(b4_ne(e, f), b4_ne(e, h));
#define EQ(P, Q) b4_lt(f4_df(P, Q), thr)
#define EQ2(P, Q) b4_lt(f4_df(P, Q), thr2)
      if (corner == 1.0f) {
        r1 = ne_ef_eh;
      } else if (corner == 2.0f) {
        b4 t = b4_and(b4_not(EQ(f, b)), b4_not(EQ(h, d)));
        t = b4_or(t, EQ(e, i));
        t = b4_and(t, b4_not(EQ(f, i4)));
        t = b4_and(t, b4_not(EQ(h, i5)));
        t = b4_or(t, EQ(e, g));
        t = b4_or(t, EQ(e, c));
        r1 = b4_and(ne_ef_eh, t);
      } else {
        b4 t1 = b4_or(b4_and(b4_not(EQ(f, b)), b4_not(EQ(f, c))), b4_and(b4_not(EQ(h, d)), b4_not(EQ(h, g))));
        b4 t2 = b4_and(EQ(e, i), b4_or(b4_and(b4_not(EQ(f, f4_)), b4_not(EQ(f, i4))),
                                       b4_and(b4_not(EQ(h, h5)), b4_not(EQ(h, i5)))));
        b4 t3 = b4_or(EQ(e, g), EQ(e, c));
        r1 = b4_and(ne_ef_eh, b4_or(t1, b4_or(t2, t3)));
      }
      b4 r2_left = b4_and(b4_ne(e, g), b4_ne(d, g));
      b4 r2_up = b4_and(b4_ne(e, c), b4_ne(b, c));
      b4 r3_left = b4_and(EQ2(g, g0), b4_not(EQ2(d0, g0)));
      b4 r3_up = b4_and(EQ2(c, c1), b4_not(EQ2(b1, c1)));
#undef EQ
#undef EQ2
      f4 fx45 = line_sstep(Ao, Bo, Co, fpy, fpx), fx30 = line_sstep(Ao, Bx, Cx, fpy, fpx);
      f4 fx60 = line_sstep(Ao, By, Cy, fpy, fpx), fx15 = line_sstep(Az, Bz, Cz, fpy, fpx);
      f4 fx75 = line_sstep(Aw, Bw, Cw, fpy, fpx);
      f4 wd1 = wd(e, c, g, i, h5, f4_, h, f), wd2 = wd(h, d, i5, f, i4, b, e, i);
      f4 dfg = f4_df(f, g), dhc = f4_df(h, c), def = f4_df(e, f), deh = f4_df(e, h);
      b4 edr, edr_left, edr_up, px, nc45, nc30, nc60, nc15, nc75, nc;
      float maximo[4];
      for (int k = 0; k < 4; ++k) {
        edr.v[k] = (wd1.v[k] < wd2.v[k]) && r1.v[k];
        edr_left.v[k] = (lv2 * dfg.v[k] <= dhc.v[k]) && r2_left.v[k];
        edr_up.v[k] = (dfg.v[k] >= lv2 * dhc.v[k]) && r2_up.v[k];
        nc45.v[k] = edr.v[k] && (fx45.v[k] != 0.0f);
        nc30.v[k] = edr.v[k] && edr_left.v[k] && (fx30.v[k] != 0.0f);
        nc60.v[k] = edr.v[k] && edr_up.v[k] && (fx60.v[k] != 0.0f);
        nc15.v[k] = edr.v[k] && edr_left.v[k] && r3_left.v[k] && (fx15.v[k] != 0.0f);
        nc75.v[k] = edr.v[k] && edr_up.v[k] && r3_up.v[k] && (fx75.v[k] != 0.0f);
        px.v[k] = def.v[k] <= deh.v[k];
        nc.v[k] = nc75.v[k] || nc15.v[k] || nc30.v[k] || nc60.v[k] || nc45.v[k];
        float f45 = (nc45.v[k] ? 1.0f : 0.0f) * fx45.v[k], f30 = (nc30.v[k] ? 1.0f : 0.0f) * fx30.v[k];
        float f60 = (nc60.v[k] ? 1.0f : 0.0f) * fx60.v[k], f15 = (nc15.v[k] ? 1.0f : 0.0f) * fx15.v[k];
        float f75 = (nc75.v[k] ? 1.0f : 0.0f) * fx75.v[k];
        float m1 = f15 > f75 ? f15 : f75, m2 = f30 > f60 ? f30 : f60;
        float m3 = m1 > m2 ? m1 : m2;
        maximo[k] = m3 > f45 ? m3 : f45;
      }
      /* :335-343; with no rule firing the GLSL leaves pix/blend undefined: the compiler drops
       * the select against the undefined value, i.e. the last arm is taken (its blend is 0) */
      const o_vec4 pk[4] = {px.v[0] ? F : Hh, px.v[1] ? B : F, px.v[2] ? D : B, px.v[3] ? Hh : D};
      int k1 = nc.v[0] ? 0 : nc.v[1] ? 1 : nc.v[2] ? 2 : 3;
      int k2 = nc.v[3] ? 3 : nc.v[2] ? 2 : nc.v[1] ? 1 : 0;
      o_vec4 res1 = mix3(E, pk[k1], maximo[k1]);
      o_vec4 res2 = mix3(E, pk[k2], maximo[k2]);
      /* mix(res1, res2, step(c_df(E,res1), c_df(E,res2))): a mix whose weight is a bool-to-float
       * is compiled to a select, so res2 is taken exactly */
      o_vec4 res = c_df(E, res2) < c_df(E, res1) ? res1 : res2;
      res.w = 1.0f;
      o_store_pixel(a, x, y, res);
    }
}
void o_pass_xbr_lv3(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  xbr_lv3_body(a);
  o_fp_leave(csr);
}

/* ----------------------------------------------------------------------------- xbr-lv2 -- */
/* shaders/shaders_glsl/xbr/shaders/xbr-lv2.glsl (VS 100-117 = the same 5x5 coordinate set as xbr-lv3, FS 260-361),
 * CORNER_C + SMOOTH_TIPS as the file defines them; both branches of small_details.
 * params: XBR_SCALE (a commented-out `//#pragma parameter` line that the reference's scan still picks up; unused),
 * XBR_Y_WEIGHT, XBR_EQ_THRESHOLD, XBR_LV1_COEFFICIENT, XBR_LV2_COEFFICIENT, small_details
 *
 * The shader reads a variable it never assigns: `f4` (declared FS 295; xbr-lv3 has f4 = h5.yzwx).  What llvmpipe makes
 * of its uses was fitted on the goldens: in the five-term weighted distance wd1 it reads as `i`, eq(f, f4) in the
 * CORNER_C rule comes out true, and in the seven-term distances of the small_details branch |x - f4| is 0.  With the
 * line equations pinned by float-target probes taken inside the complete shader (line_clamp below) the oracle is
 * byte-exact on all five 8-bit goldens and bit-identical on both float goldens (both branches of small_details). */
static inline float dot_rgbw(o_vec4 p) { return p.x * 14.352f + (p.y * 28.176f + p.z * 5.472f); }
static inline f4 lumc(o_vec4 p0, o_vec4 p1, o_vec4 p2, o_vec4 p3) {
  f4 r = {{dot_rgbw(p0), dot_rgbw(p1), dot_rgbw(p2), dot_rgbw(p3)}};
  return r;
}
/* clamp((A*fp.y + B*fp.x + delta - C [- Ci]) / (2*delta), 0, 1).  `delta` is a mutable global, so nothing is folded at
 * compile time; pinned with float-target probes taken inside the complete shader (all 16 components bit-identical):
 *  - the 30 and 60 degree lines: the plain addends combine first and the products join from the inside out,
 *    B*fx + (A*fy + (delta - C));
 *  - the two 45 degree lines (fx45, fx45i) share s = B*fx + (A*fy + delta): (s - Co) and (s - (Co + Ci));
 * then a true division by 2*delta and the clamp. */
static inline f4 line_clamp(const float* A, const float* B, const float* dl, const float* C, float ci, int shared, float fy, float fx) {
  f4 r;
  for (int k = 0; k < 4; ++k) {
    const float cc = ci != 0.0f ? C[k] + ci : C[k];
    const float num = shared ? (B[k] * fx + (A[k] * fy + dl[k])) - cc : B[k] * fx + (A[k] * fy + (dl[k] - cc));
    float t = num / (2.0f * dl[k]);
    t = t > 0.0f ? t : 0.0f;
    r.v[k] = t < 1.0f ? t : 1.0f;
  }
  return r;
}
/* weighted_distance (FS 222-225), seven terms: balanced like wd's five, (((ab + ac) + (de + df)) + (ij + kl)) + 2*gh
 * (of eight candidate trees the best fit on the float golden; the differences between them are below the
 * residual this shader's parity carries anyway) */
static inline f4 wd7(f4 a, f4 b, f4 c, f4 d, f4 e, f4 f, f4 g, f4 h, f4 i, f4 j, f4 k, f4 l) {
  f4 t1 = f4_df(a, b), t2 = f4_df(a, c), t3 = f4_df(d, e), t4 = f4_df(d, f), t5 = f4_df(i, j), t6 = f4_df(k, l), t7 = f4_df(g, h), r;
  for (int q = 0; q < 4; ++q) r.v[q] = (((t1.v[q] + t2.v[q]) + (t3.v[q] + t4.v[q])) + (t5.v[q] + t6.v[q])) + 2.0f * t7.v[q];
  return r;
}
/* mul(mat4x3(p0..p3), y_weight * Y) = (y_weight * Y) * mat: one dot per column, x*c0 + (y*c1 + z*c2) */
static inline f4 lumy(o_vec4 p0, o_vec4 p1, o_vec4 p2, o_vec4 p3, const float* yw) {
  const o_vec4* p[4] = {&p0, &p1, &p2, &p3};
  f4 r;
  for (int k = 0; k < 4; ++k) r.v[k] = yw[0] * p[k]->x + (yw[1] * p[k]->y + yw[2] * p[k]->z);
  return r;
}
static void xbr_lv2_body(const o_pass_args* a) {
  const int W = a->out_w, H = a->out_h;
  const float thr = a->params[2], lv2 = a->params[4];
  const int details = !(a->params[5] < 0.5f);
  const float yw[3] = {a->params[1] * 0.2126f, a->params[1] * 0.7152f, a->params[1] * 0.0722f};   /* y_weight * Y */
  const float tsx = (float)a->in->w, tsy = (float)a->in->h;
  const float dx = 1.0f / tsx, dy = 1.0f / tsy;
  const float xoff[5] = {-2.0f * dx, -dx, 0.0f, dx, 2.0f * dx};
  const float yoff[5] = {-2.0f * dy, -dy, 0.0f, dy, 2.0f * dy};
  o_varying vx[5], vy[5];
  for (int k = 0; k < 5; ++k) {
    vx[k] = o_varying_setup(0.f + xoff[k], 1.f + xoff[k], 1.f + xoff[k], 0.f + xoff[k], W, H, a->out_fmt);
    vy[k] = o_varying_setup(0.f + yoff[k], 0.f + yoff[k], 1.f + yoff[k], 1.f + yoff[k], W, H, a->out_fmt);
  }
  static const float Ao[4] = {1.0f, -1.0f, -1.0f, 1.0f}, Bo[4] = {1.0f, 1.0f, -1.0f, -1.0f}, Co[4] = {1.5f, 0.5f, -0.5f, 0.5f};
  static const float Bx[4] = {0.5f, 2.0f, -0.5f, -2.0f}, Cx[4] = {1.0f, 1.0f, -0.5f, 0.0f};
  static const float By[4] = {2.0f, 0.5f, -2.0f, -0.5f}, Cy[4] = {2.0f, 0.0f, -1.0f, 0.5f};
  const float third = 1.0f / 3.0f, sixth = 0.5f / 3.0f;
  const float delta[4] = {third, third, third, third}, delta_l[4] = {sixth, third, sixth, third}, delta_u[4] = {third, sixth, third, sixth};
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      int lo = o_lower_tri(x, y, W, H);
      float cx[5], cy[5];
      for (int k = 0; k < 5; ++k) {
        cx[k] = o_varying_at(&vx[k], x, y, lo);
        cy[k] = o_varying_at(&vy[k], x, y, lo);
      }
      float fpx = cx[2] * tsx, fpy = cy[2] * tsy;
      fpx = fpx - floorf(fpx);
      fpy = fpy - floorf(fpy);
#define T(i, j) o_sample(a->in, cx[i], cy[j])
      o_vec4 A1 = T(1, 0), B1 = T(2, 0), C1 = T(3, 0);
      o_vec4 A = T(1, 1), B = T(2, 1), C = T(3, 1);
      o_vec4 D = T(1, 2), E = T(2, 2), F = T(3, 2);
      o_vec4 G = T(1, 3), Hh = T(2, 3), I = T(3, 3);
      o_vec4 G5 = T(1, 4), H5 = T(2, 4), I5 = T(3, 4);
      o_vec4 A0 = T(0, 1), D0 = T(0, 2), G0 = T(0, 3);
      o_vec4 C4 = T(4, 1), F4 = T(4, 2), I4 = T(4, 3);
#undef T
      f4 b = lumc(B, D, Hh, F), c = lumc(C, A, G, I), e = lumc(E, E, E, E);
      f4 d = YZWX(b), f = WXYZ(b), g = ZWXY(c), h = ZWXY(b), i = WXYZ(c);
      f4 i4 = lumc(I4, C1, A0, G5), i5 = lumc(I5, C4, A1, G0), h5 = lumc(H5, F4, B1, D0);
      if (details) { i4 = lumy(I4, C1, A0, G5, yw); i5 = lumy(I5, C4, A1, G0, yw); h5 = lumy(H5, F4, B1, D0, yw); }
      /* `f4` is declared (FS 295) but never assigned in this file - xbr-lv3 has f4 = h5.yzwx - so wd1 and the
       * CORNER_C rule read an undefined value, which llvmpipe materialises as 0 */
      const f4 f4_ = i;   /* the unassigned `f4` as wd1 sees it (see above) */
      (void)A1; (void)B1; (void)G0; (void)D0;
#define EQ(P, Q) b4_le(f4_df(P, Q), thr) /* step(df, thr): df <= thr */
      b4 ne_ef_eh = b4_and(b4_ne(e, f), b4_ne(e, h));
      /* CORNER_C: irlv1 = irlv0 * (neq(f,b)*neq(f,c) + neq(h,d)*neq(h,g) + eq(e,i)*(neq(f,f4)*neq(f,i4) + neq(h,h5)*neq(h,i5))
       * + eq(e,g) + eq(e,c)); all factors are 0/1, edr tests step(0.5, irlv1) */
      b4 t1 = b4_or(b4_and(b4_not(EQ(f, b)), b4_not(EQ(f, c))), b4_and(b4_not(EQ(h, d)), b4_not(EQ(h, g))));
      /* neq(f, f4) * neq(f, i4) drops out: eq(f, f4) reads as true (see above) */
      b4 t2 = b4_and(EQ(e, i), b4_and(b4_not(EQ(h, h5)), b4_not(EQ(h, i5))));
      b4 t3 = b4_or(EQ(e, g), EQ(e, c));
      b4 r1 = b4_and(ne_ef_eh, b4_or(t1, b4_or(t2, t3)));
#undef EQ
      b4 r2_left = b4_and(b4_ne(e, g), b4_ne(d, g));
      b4 r2_up = b4_and(b4_ne(e, c), b4_ne(b, c));
      f4 fx45i = line_clamp(Ao, Bo, delta, Co, 0.25f, 1, fpy, fpx), fx45 = line_clamp(Ao, Bo, delta, Co, 0.0f, 1, fpy, fpx);
      f4 fx30 = line_clamp(Ao, Bx, delta_l, Cx, 0.0f, 0, fpy, fpx), fx60 = line_clamp(Ao, By, delta_u, Cy, 0.0f, 0, fpy, fpx);
      f4 wd1 = wd(e, c, g, i, h5, f4_, h, f), wd2 = wd(h, d, i5, f, i4, b, e, i);
      if (details) {   /* FS 322-323 */
        /* the unassigned f4 again: |x - f4| comes out 0 in both calls (fitted: of 144 combinations of what the two
         * uses could read, only "f4 = its partner" reproduces llvmpipe) */
        wd1 = wd7(e, c, g, i, i, h5, h, f, b, d, i4, i5);
        wd2 = wd7(h, d, i5, f, b, i4, e, i, g, h5, c, c);
      }
      f4 dfg = f4_df(f, g), dhc = f4_df(h, c), def = f4_df(e, f), deh = f4_df(e, h);
      float maximos[4];
      int px[4];
      for (int k = 0; k < 4; ++k) {
        const int edri = (wd1.v[k] <= wd2.v[k]) && ne_ef_eh.v[k];
        const int edr = (wd1.v[k] + 0.1f <= wd2.v[k]) && r1.v[k];
        const int edr_l = (lv2 * dfg.v[k] <= dhc.v[k]) && r2_left.v[k] && edr;
        const int edr_u = (lv2 * dhc.v[k] <= dfg.v[k]) && r2_up.v[k] && edr;
        const float f45 = (edr ? 1.0f : 0.0f) * fx45.v[k], f30 = (edr_l ? 1.0f : 0.0f) * fx30.v[k];
        const float f60 = (edr_u ? 1.0f : 0.0f) * fx60.v[k], f45i = (edri ? 1.0f : 0.0f) * fx45i.v[k];
        px[k] = def.v[k] <= deh.v[k];
        const float m1 = f30 > f60 ? f30 : f60, m2 = f45 > f45i ? f45 : f45i;
        maximos[k] = m1 > m2 ? m1 : m2;
      }
      /* res1 = mix(mix(E, mix(H,F,px.x), maximos.x), mix(B,D,px.z), maximos.z); px is 0/1: the inner mixes are selects */
      o_vec4 res1 = mix3(E, px[0] ? F : Hh, maximos[0]);
      res1 = mix3(res1, px[2] ? D : B, maximos[2]);
      o_vec4 res2 = mix3(E, px[1] ? B : F, maximos[1]);
      res2 = mix3(res2, px[3] ? Hh : D, maximos[3]);
      o_vec4 res = c_df(E, res2) < c_df(E, res1) ? res1 : res2;
      res.w = 0.0f;  /* FragColor.xyz only: alpha is never written and comes out 0 on llvmpipe */
      o_store_pixel(a, x, y, res);
    }
}
void o_pass_xbr_lv2(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  xbr_lv2_body(a);
  o_fp_leave(csr);
}

/* ntsc/shaders/ntsc-gauss-pass.glsl (ntsc/ntsc-{256px,320px}[-svideo]-gauss-scanline.glslp, ntsc.glslp, ntsc-svideo.glslp): five source
 * lines around the target pixel, pow(., NTSC_CRT_GAMMA), weighted with exp(-5 d^2) = exp2((-7.213475 d) d) of the line distance,
 * x 1.15, pow(., 1 / NTSC_DISPLAY_GAMMA).  VS: one = 1 / TextureSize.y, pix_no = TexCoord.y * TextureSize.y.  params: the two gammas. */
void o_pass_ntsc_gauss(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  const float tsy = (a->pass_index == 3 && H != a->in->h) ? (float)H : (float)a->in->h;   /* the reference's TextureSize.y rule for pass index 3 */
  const float one = 1.0f / tsy, crt = a->params[0], inv = 1.0f / a->params[1];
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  o_varying pn = o_varying_setup(0.f * tsy, 0.f * tsy, 1.f * tsy, 1.f * tsy, W, H, a->out_fmt);
  o_varying po = o_varying_setup(one, one, one, one, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo), o1 = o_varying_at(&po, x, y, lo);
      const float pno = o_varying_at(&pn, x, y, lo), fr = pno - floorf(pno);
      const float off[5] = {-2.0f * o1, -o1, 0.0f, o1, 2.0f * o1}, d[5] = {1.5f + fr, 0.5f + fr, fr + -0.5f, -1.5f + fr, -2.5f + fr};
      float acc[3] = {0.f, 0.f, 0.f};
      for (int k = 0; k < 5; ++k) {
        const o_vec4 t = o_sample(a->in, u, k == 2 ? v : v + off[k]);
        const float w = o_exp2((-7.213475f * d[k]) * d[k]);
        const float c[3] = {o_pow(t.x, crt) * w, o_pow(t.y, crt) * w, o_pow(t.z, crt) * w};
        for (int q = 0; q < 3; ++q) acc[q] = k == 0 ? c[q] : acc[q] + c[q];
      }
      const o_vec4 out = {o_pow(1.15f * acc[0], inv), o_pow(1.15f * acc[1], inv), o_pow(1.15f * acc[2], inv), 1.0f};
      o_store_pixel(a, x, y, out);
    }
  o_fp_leave(csr);
}
