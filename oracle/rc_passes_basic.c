/* TEST INFRASTRUCTURE - see rc_oracle.h.
 *
 * Hand restatement of the arithmetic of three reference shader assets:
 *   stock    shaders/shaders_glsl/stock.glsl                      (passthrough)
 *   scanline shaders/shaders_glsl/scanlines/shaders/scanline.glsl (VS line 50, FS 107-113)
 *   crt-pi   shaders/shaders_glsl/crt/shaders/crt-pi.glsl         (VS 96-103, FS 131-232;
 *            compile-time switches SCANLINES, MULTISAMPLE, GAMMA, MASK_TYPE 1 as shipped)
 * Operation order follows the GLSL expression trees; nothing is fused.
 */
#include <math.h>

#include "rc_oracle.h"

static void store_px(const o_pass_args* a, int x, int y, o_vec4 c) {
  size_t i = ((size_t)y * a->out_w + x) * 4;
  if (a->out_fmt == O_FMT_F32) {
    float* d = (float*)a->dst + i;
    d[0] = c.x; d[1] = c.y; d[2] = c.z; d[3] = c.w;
  } else if (a->out_fmt == O_FMT_SRGB8) {
    uint8_t* d = (uint8_t*)a->dst + i;
    d[0] = o_store_srgb8(c.x); d[1] = o_store_srgb8(c.y); d[2] = o_store_srgb8(c.z);
    d[3] = o_store_unorm8(c.w);
  } else {
    uint8_t* d = (uint8_t*)a->dst + i;
    d[0] = o_store_unorm8(c.x); d[1] = o_store_unorm8(c.y); d[2] = o_store_unorm8(c.z);
    d[3] = o_store_unorm8(c.w);
  }
}

void o_store_pixel(const o_pass_args* a, int x, int y, o_vec4 c) { store_px(a, x, y, c); }

static void o_pass_stock_body(const o_pass_args* a);
void o_pass_stock(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  o_pass_stock_body(a);
  o_fp_leave(csr);
}
/* llvmpipe executes a shader that only copies one RGBA8 texture to a plain RGBA8 target (clamp to
 * edge) through its blit fast path (Mesa lp_linear_sampler.c), which steps the texture coordinate in
 * 16.16 fixed point, re-anchored every 64 target pixels.  Measured on the GL (exact on every size
 * tried, including ones full of samples that land exactly on texel boundaries such as 240 -> 1080):
 *   D   = trunc(f32(d * T) * 65536)                       per-pixel step (d: plane slope, T: texture size)
 *   S0  = trunc(f32(f32(f32(d * T) * j) + f32(a0 * T)) * 65536)   at the 64-pixel tile origin j
 *   idx = clamp((S0 + (x - j) * D) >> 16, 0, T - 1)
 * LINEAR: the same coordinate minus half a texel (S - 32768); texel pair i = S >> 16 and i + 1 (clamped),
 * weight w = (S >> 8) & 255, every lerp a + (((b - a) * w) >> 8) on bytes.  A 64x64 target tile whose
 * samples need no clamping lerps horizontally first, then vertically; a tile that touches the texture
 * edge does it the other way round (llvmpipe has a separate clamping fetch routine).  Fitted on the GL:
 * 0 mismatches over 4.4e5 pixels of seven geometries. */
static int blit_coord(float a0, float d, int texsize, int x) {
  const float T = (float)texsize, K = 65536.0f;
  const float fd = d * T;
  const int D = (int)(fd * K);
  const int j = x & ~63;
  const float s0 = fd * (float)j + a0 * T;
  const int S0 = (int)(s0 * K);
  return S0 + (x - j) * D;
}
static int clampidx(int i, int n) { return i < 0 ? 0 : (i > n - 1 ? n - 1 : i); }
static int blit_index(float a0, float d, int texsize, int x) { return clampidx(blit_coord(a0, d, texsize, x) >> 16, texsize); }
static int lerp8(int a, int b, int w) { return a + (((b - a) * w) >> 8); }
/* does the 64-pixel target tile holding x need clamping along this axis? */
static int blit_tile_inside(float a0, float d, int texsize, int x, int extent) {
  const int j = x & ~63, last = (j + 63 < extent - 1) ? j + 63 : extent - 1;   /* caller rounds the x extent up to 4: spans are 4 pixels wide */
  const int lo = (blit_coord(a0, d, texsize, j) - 32768) >> 16, hi = (blit_coord(a0, d, texsize, last) - 32768) >> 16;
  return lo >= 0 && hi + 1 <= texsize - 1;
}

static void o_pass_stock_body(const o_pass_args* a) {
  const int W = a->out_w, H = a->out_h;
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  if (a->flags & O_FLAG_STOCK_NO_BLIT) {   /* a mip level: drawn by the GL's blitter, whose quad is a triangle fan */
    tu = o_varying_setup_fan(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt);
    tv = o_varying_setup_fan(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  }
  if (a->in->fmt == O_FMT_RGBA8 && a->in->wrap == O_WRAP_EDGE && a->out_fmt == O_FMT_RGBA8 && !(a->flags & O_FLAG_STOCK_NO_BLIT)) {
    const uint8_t* tex = (const uint8_t*)a->in->data;
    const int TW = a->in->w, TH = a->in->h;
    for (int y = a->y0; y < a->y1; ++y)
      for (int x = 0; x < W; ++x) {
        uint8_t* dst = (uint8_t*)a->dst + ((size_t)y * W + x) * 4;
        if (!a->in->linear) {
          const uint8_t* t = tex + ((size_t)blit_index(tv.a0_lo, tv.dy_lo, TH, y) * TW + blit_index(tu.a0_lo, tu.dx_lo, TW, x)) * 4;
          for (int c = 0; c < 4; ++c) dst[c] = t[c];
          continue;
        }
        const int sx = blit_coord(tu.a0_lo, tu.dx_lo, TW, x) - 32768, sy = blit_coord(tv.a0_lo, tv.dy_lo, TH, y) - 32768;
        const int x0 = clampidx(sx >> 16, TW), x1 = clampidx((sx >> 16) + 1, TW), wx = (sx >> 8) & 255;
        const int y0 = clampidx(sy >> 16, TH), y1 = clampidx((sy >> 16) + 1, TH), wy = (sy >> 8) & 255;
        const int inside = blit_tile_inside(tu.a0_lo, tu.dx_lo, TW, x, (W + 3) & ~3) && blit_tile_inside(tv.a0_lo, tv.dy_lo, TH, y, H);
        for (int c = 0; c < 4; ++c) {
          const int A = tex[((size_t)y0 * TW + x0) * 4 + c], B = tex[((size_t)y0 * TW + x1) * 4 + c];
          const int Cc = tex[((size_t)y1 * TW + x0) * 4 + c], D = tex[((size_t)y1 * TW + x1) * 4 + c];
          dst[c] = (uint8_t)(inside ? lerp8(lerp8(A, B, wx), lerp8(Cc, D, wx), wy) : lerp8(lerp8(A, Cc, wy), lerp8(B, D, wy), wx));
        }
      }
    return;
  }
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      int lo = o_lower_tri(x, y, W, H);
      store_px(a, x, y, o_sample(a->in, o_varying_at(&tu, x, y, lo), o_varying_at(&tv, x, y, lo)));
    }
}

/* params: SCANLINE_BASE_BRIGHTNESS, SCANLINE_SINE_COMP_A, SCANLINE_SINE_COMP_B, size */
static void o_pass_scanline_body(const o_pass_args* a);
void o_pass_scanline(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  o_pass_scanline_body(a);
  o_fp_leave(csr);
}
static void o_pass_scanline_body(const o_pass_args* a) {
  const int W = a->out_w, H = a->out_h;
  const float base = a->params[0], comp_a = a->params[1], comp_b = a->params[2], size = a->params[3];
  const float pi = 3.141592654f;
  /* VS: omega = vec2(pi * size * OutputSize.x, 2.0 * pi * TextureSize.y); same at all 4
   * vertices, so the interpolated value is the vertex value. */
  const float omega_x = (pi * size) * (float)W;
  const float omega_y = (2.0f * pi) * (float)a->in->h;
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      int lo = o_lower_tri(x, y, W, H);
      float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      o_vec4 res = o_sample(a->in, u, v);
      float d = comp_a * o_sin(u * omega_x) + comp_b * o_sin(v * omega_y);
      float k = base + d;
      o_vec4 c = {res.x * k, res.y * k, res.z * k, 1.0f};
      store_px(a, x, y, c);
    }
}

/* params: CURVATURE_X, CURVATURE_Y, MASK_BRIGHTNESS, SCANLINE_WEIGHT,
 *         SCANLINE_GAP_BRIGHTNESS, BLOOM_FACTOR, INPUT_GAMMA, OUTPUT_GAMMA */
static inline float crtpi_weight(float dist, float sw, float gap) {
  float w = 1.0f - (dist * dist) * sw;
  return w > gap ? w : gap; /* max(w, gap) */
}

static void o_pass_crt_pi_body(const o_pass_args* a);
void o_pass_crt_pi(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  o_pass_crt_pi_body(a);
  o_fp_leave(csr);
}
static void o_pass_crt_pi_body(const o_pass_args* a) {
  const int W = a->out_w, H = a->out_h;
  const float mask_b = a->params[2], sw = a->params[3], gap = a->params[4], bloom = a->params[5];
  const float in_gamma = a->params[6], out_gamma = a->params[7];
  const float tsy = (float)a->in->h; /* TextureSize == InputSize */
  /* VS: filterWidth = (InputSize.y / OutputSize.y) / 3.0;  TEX0 = TexCoord * 1.0001 */
  const float filter_width = (tsy / (float)H) / 3.0f;
  const float k1 = 1.0001f;
  o_varying tu = o_varying_setup(0.f * k1, 1.f * k1, 1.f * k1, 0.f * k1, W, H, a->out_fmt);
  o_varying tv = o_varying_setup(0.f * k1, 0.f * k1, 1.f * k1, 1.f * k1, W, H, a->out_fmt);
  const float inv_out_gamma = 1.0f / out_gamma;
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      int lo = o_lower_tri(x, y, W, H);
      float tcx = o_varying_at(&tu, x, y, lo), tcy = o_varying_at(&tv, x, y, lo);
      float pix_y = tcy * tsy;
      float temp_y = floorf(pix_y) + 0.5f;
      float y_coord = temp_y / tsy;
      float dy = pix_y - temp_y;
      float slw = crtpi_weight(dy, sw, gap);
      slw += crtpi_weight(dy - filter_width, sw, gap);
      slw += crtpi_weight(dy + filter_width, sw, gap);
      slw *= 0.3333333f;
      float sign_y = dy > 0.f ? 1.f : (dy < 0.f ? -1.f : 0.f);
      dy = dy * dy;
      dy = dy * dy;
      dy *= 8.0f;
      dy /= tsy;
      dy *= sign_y;
      o_vec4 c = o_sample(a->in, tcx, y_coord + dy);
      float r = o_pow(c.x, in_gamma), g = o_pow(c.y, in_gamma), b = o_pow(c.z, in_gamma);
      slw *= bloom;
      r *= slw; g *= slw; b *= slw;
      r = o_pow(r, inv_out_gamma); g = o_pow(g, inv_out_gamma); b = o_pow(b, inv_out_gamma);
      float fx = ((float)x + 0.5f) * 1.0001f * 0.5f;
      float which = fx - floorf(fx);
      o_vec4 o;
      if (which < 0.5f) { o.x = r * mask_b; o.y = g * 1.0f; o.z = b * mask_b; }
      else { o.x = r * 1.0f; o.y = g * mask_b; o.z = b * 1.0f; }
      o.w = 1.0f;
      store_px(a, x, y, o);
    }
}

/* mix_frames  shaders/shaders_glsl/motionblur/shaders/mix_frames.glsl (VS :51-55, FS :94-107):
 * 50:50 blend of the input with PrevTexture (extra[0]).  mix with the constant weight 0.5 is
 * a*(1-0.5) + b*0.5.  TEX0 = TexCoord * 1.0001. */
static void o_pass_mix_frames_body(const o_pass_args* a) {
  const int W = a->out_w, H = a->out_h;
  const float k1 = 1.0001f;
  o_varying tu = o_varying_setup(0.f * k1, 1.f * k1, 1.f * k1, 0.f * k1, W, H, a->out_fmt);
  o_varying tv = o_varying_setup(0.f * k1, 0.f * k1, 1.f * k1, 1.f * k1, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      int lo = o_lower_tri(x, y, W, H);
      float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      o_vec4 c = o_sample(a->in, u, v), p = o_sample(a->extra[0], u, v);
      o_vec4 o = {c.x * 0.5f + p.x * 0.5f, c.y * 0.5f + p.y * 0.5f, c.z * 0.5f + p.z * 0.5f, 1.0f};
      store_px(a, x, y, o);
    }
}
void o_pass_mix_frames(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  o_pass_mix_frames_body(a);
  o_fp_leave(csr);
}

/* The other frame-history shaders of motionblur/ (each a one-pass preset, NEAREST input).  extra[] = the history
 * textures in the order given per function; every one of them samples all textures at the same coordinate
 * (motionblur-simple's PrevNTexCoord attributes alias TexCoord's location in the reference, ShaderEngine.cpp:712-718).
 *
 * motionblur-simple.glsl (VS 81-97, FS 178-199): extra[0..6] = Prev6, Prev5, ..., Prev1, PrevTexture;
 *   c = P6; c = (c + P5)/2; ... c = (c + Prev)/2; c = (c + Texture)/2, all four components. */
static void o_pass_motionblur_simple_body(const o_pass_args* a) {
  const int W = a->out_w, H = a->out_h;
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      o_vec4 c = o_sample(a->extra[0], u, v);
      for (int k = 1; k <= 7; ++k) {
        const o_vec4 t = k < 7 ? o_sample(a->extra[k], u, v) : o_sample(a->in, u, v);
        c.x = (c.x + t.x) / 2.0f; c.y = (c.y + t.y) / 2.0f; c.z = (c.z + t.z) / 2.0f; c.w = (c.w + t.w) / 2.0f;
      }
      store_px(a, x, y, c);
    }
}
void o_pass_motionblur_simple(const o_pass_args* a) { unsigned csr = o_fp_enter(); o_pass_motionblur_simple_body(a); o_fp_leave(csr); }

/* braid-rewind.glsl (VS 50-66, FS 135-160): the blend with the seven history frames applies only while rewinding
 * (FrameDirection < 0); the reference always sets FrameDirection = 1 (ShaderEngine.cpp:2164-2170), so the pass outputs
 * the current frame's sample, all four components.  (The history samplers are still declared, hence bound.) */
static void o_pass_braid_rewind_body(const o_pass_args* a) {
  const int W = a->out_w, H = a->out_h;
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      store_px(a, x, y, o_sample(a->in, o_varying_at(&tu, x, y, lo), o_varying_at(&tv, x, y, lo)));
    }
}
void o_pass_braid_rewind(const o_pass_args* a) { unsigned csr = o_fp_enter(); o_pass_braid_rewind_body(a); o_fp_leave(csr); }

/* response-time.glsl (VS 52-56, FS 122-136): param response_time; extra[0..6] = PrevTexture, Prev1, ..., Prev6;
 *   rgb += (prev_k - rgb) * response_time^(k+1), alpha 0.  pow(x, 2.0) is lowered to x*x (and 4.0 to two squarings),
 *   the other powers run the exp2/log2 polynomials. */
static void o_pass_response_time_body(const o_pass_args* a) {
  const int W = a->out_w, H = a->out_h;
  const float rt = a->params[0];
  float k[7];
  k[0] = rt;
  k[1] = rt * rt;
  k[2] = o_pow(rt, 3.0f);
  k[3] = (rt * rt) * (rt * rt);
  k[4] = o_pow(rt, 5.0f);
  k[5] = o_pow(rt, 6.0f);
  k[6] = o_pow(rt, 7.0f);
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      o_vec4 c = o_sample(a->in, u, v);
      for (int q = 0; q < 7; ++q) {
        const o_vec4 p = o_sample(a->extra[q], u, v);
        c.x = c.x + (p.x - c.x) * k[q]; c.y = c.y + (p.y - c.y) * k[q]; c.z = c.z + (p.z - c.z) * k[q];
      }
      c.w = 0.0f;
      store_px(a, x, y, c);
    }
}
void o_pass_response_time(const o_pass_args* a) { unsigned csr = o_fp_enter(); o_pass_response_time_body(a); o_fp_leave(csr); }

/* mix_frames_smart.glsl (VS 41-45: TEX0 = TexCoord * 1.0001; FS 64-105): param DEFLICKER_EMPHASIS;
 * extra[0..4] = PrevTexture, Prev1, ..., Prev4.  Mixes the current and the previous frame 50:50 where alternate frames
 * repeat and adjacent ones differ (a flicker pattern); all the tests are exact comparisons or a step(). */
static void o_pass_mix_frames_smart_body(const o_pass_args* a) {
  const int W = a->out_w, H = a->out_h;
  const float edge = 0.000001f + a->params[0];
  const float k1 = 1.0001f;
  o_varying tu = o_varying_setup(0.f * k1, 1.f * k1, 1.f * k1, 0.f * k1, W, H, a->out_fmt);
  o_varying tv = o_varying_setup(0.f * k1, 0.f * k1, 1.f * k1, 1.f * k1, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      o_vec4 c[6];
      c[0] = o_sample(a->in, u, v);
      for (int q = 0; q < 5; ++q) c[q + 1] = o_sample(a->extra[q], u, v);
#define IS_EQ(i, j) ((c[i].x == c[j].x && c[i].y == c[j].y && c[i].z == c[j].z) ? 1.0f : 0.0f)
#define IS_AEQ(i, j) ((!(fabsf(c[i].x - c[j].x) >= edge) && !(fabsf(c[i].y - c[j].y) >= edge) && !(fabsf(c[i].z - c[j].z) >= edge)) ? 1.0f : 0.0f)
      const float alt = IS_AEQ(0, 2) * IS_AEQ(2, 4) + IS_AEQ(1, 3) * IS_AEQ(3, 5);
      float m = (1.0f - IS_EQ(0, 3)) * (1.0f - IS_EQ(0, 5)) * (1.0f - IS_EQ(1, 2)) * (1.0f - IS_EQ(1, 4)) * (1.0f - IS_EQ(2, 3)) * (1.0f - IS_EQ(2, 5));
      m = m * (alt < 1.0f ? alt : 1.0f);
#undef IS_EQ
#undef IS_AEQ
      const float t = m * 0.5f;
      o_vec4 o;
      /* mix() with a per-pixel weight: a + t (b - a) */
      o.x = c[0].x + t * (c[1].x - c[0].x); o.y = c[0].y + t * (c[1].y - c[0].y); o.z = c[0].z + t * (c[1].z - c[0].z);
      o.w = 1.0f;
      store_px(a, x, y, o);
    }
}
void o_pass_mix_frames_smart(const o_pass_args* a) { unsigned csr = o_fp_enter(); o_pass_mix_frames_smart_body(a); o_fp_leave(csr); }

/* Conformance fixture tests/fixtures/conformance/feedback-persist.glsl (this repository's own shader,
 * FS main): max(cur*0.75 + old0*0.25, old1*PERSIST); extra[0] = PassFeedback0, extra[1] = PassFeedback1.
 * params: PERSIST */
static void o_pass_feedback_persist_body(const o_pass_args* a) {
  const int W = a->out_w, H = a->out_h;
  const float persist = a->params[0];
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      int lo = o_lower_tri(x, y, W, H);
      float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      o_vec4 c = o_sample(a->in, u, v), p0 = o_sample(a->extra[0], u, v), p1 = o_sample(a->extra[1], u, v);
      float m0 = c.x * 0.75f + p0.x * 0.25f, m1 = c.y * 0.75f + p0.y * 0.25f, m2 = c.z * 0.75f + p0.z * 0.25f;
      float q0 = p1.x * persist, q1 = p1.y * persist, q2 = p1.z * persist;
      o_vec4 o = {m0 < q0 ? q0 : m0, m1 < q1 ? q1 : m1, m2 < q2 ? q2 : m2, 1.0f};
      store_px(a, x, y, o);
    }
}
void o_pass_feedback_persist(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  o_pass_feedback_persist_body(a);
  o_fp_leave(csr);
}

/* Conformance fixture tests/fixtures/conformance/history-size.glsl (this repository's own shader): a frame-history shader
 * that reads TextureSize.x (a one-texel offset), OutputSize.y (a two-row pattern) and InputSize.y / TextureSize.y, to pin
 * the stale size uniforms of the reference's history re-draw (o_pass_args::uni_*).  Operation order from llvmpipe's NIR:
 * now = cur * 0.75 + right * 0.25; t = (old0 * 0.625 - now) + old1 * 0.375; (now + t * HS_MIX) * dim * (InputSize.y /
 * TextureSize.y), no contraction.  extra[0] = PrevTexture, extra[1] = Prev1Texture.  params: HS_MIX */
static void o_pass_history_size_body(const o_pass_args* a) {
  const int W = a->out_w, H = a->out_h;
  const float mixw = a->params[0];
  const float ts_x = (float)(a->uni_tex_w ? a->uni_tex_w : a->in->w), ts_y = (float)(a->uni_tex_h ? a->uni_tex_h : a->in->h);
  const float os_y = (float)(a->uni_out_h ? a->uni_out_h : a->out_h);
  const float inv = 1.0f / ts_x, cover = ts_y / ts_y;   /* InputSize == TextureSize in the reference (ShaderEngine.cpp:2401-2437) */
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      int lo = o_lower_tri(x, y, W, H);
      float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      const float u2 = u + inv;
      o_vec4 c = o_sample(a->in, u, v), r = o_sample(a->in, u2, v), p0 = o_sample(a->extra[0], u, v), p1 = o_sample(a->extra[1], u2, v);
      const float row = floorf(v * os_y), hrow = row * 0.5f, fr = hrow + (-floorf(hrow));
      const float dim = fr < 0.25f ? 1.0f : 0.75f;
      const float cc[3] = {c.x, c.y, c.z}, rr[3] = {r.x, r.y, r.z}, a0[3] = {p0.x, p0.y, p0.z}, a1[3] = {p1.x, p1.y, p1.z};
      float o[3];
      for (int k = 0; k < 3; ++k) {
        const float now = cc[k] * 0.75f + rr[k] * 0.25f;
        const float t = (a0[k] * 0.625f + (-now)) + a1[k] * 0.375f;
        o[k] = ((now + t * mixw) * dim) * cover;
      }
      o_vec4 ov = {o[0], o[1], o[2], 1.0f};
      store_px(a, x, y, ov);
    }
}
void o_pass_history_size(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  o_pass_history_size_body(a);
  o_fp_leave(csr);
}

/* crt/shaders/zfast_crt.glsl (crt/zfast-crt.glslp), FINEMASK as the file defines it; VS 101-108, FS 168-198.
 * params: BLURSCALEX, LOWLUMSCAN, HILUMSCAN, BRIGHTBOOST, MASK_DARK, MASK_FADE - the six names the reference
 * overwrites with fixed values after the pragma parameters (ShaderEngine.cpp:2260-2294), so they always arrive as
 * 0.30, 6, 8, 1.25, 0.25, 0.8 whatever the user or the preset says. */
static void o_pass_zfast_crt_body(const o_pass_args* a) {
  const int W = a->out_w, H = a->out_h;
  const float blur = a->params[0], lowlum = a->params[1], hilum = a->params[2], boost = a->params[3], mdark = a->params[4], mfade = a->params[5];
  const float tsx = (float)a->in->w, tsy = (float)a->in->h;
  o_varying tu = o_varying_setup(0.f * 1.0001f, 1.f * 1.0001f, 1.f * 1.0001f, 0.f * 1.0001f, W, H, a->out_fmt);
  o_varying tv = o_varying_setup(0.f * 1.0001f, 0.f * 1.0001f, 1.f * 1.0001f, 1.f * 1.0001f, W, H, a->out_fmt);
  const float mask_fade = 0.3333f * mfade, idx = 1.0f / tsx, idy = 1.0f / tsy;
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      const float px = u * tsx, py = v * tsy;
      const float ix = floorf(px) + 0.5f, iy = floorf(py) + 0.5f;
      const float fx = px - ix, fy = py - iy;
      float qx = (ix + ((4.0f * fx) * fx) * fx) * idx;
      const float qy = (iy + ((4.0f * fy) * fy) * fy) * idy;
      qx = qx + blur * (u - qx);
      const float Y = fy * fy, YY = Y * Y;
      const float wm0 = floorf((u * (float)W) * -0.4999f);
      const float whichmask = wm0 - floorf(wm0);
      const float mask = 1.0f + (whichmask < 0.5f ? 1.0f : 0.0f) * -mdark;
      const o_vec4 c = o_sample(a->in, qx, qy);
      const float slw = boost - lowlum * (Y - 2.05f * YY);
      const float slwb = 1.0f - hilum * (YY - (2.8f * YY) * Y);
      /* dot(colour, vec3(maskFade)): the splat factor is pulled out of the sum, a*m + (b*m + c*m) -> (a + (b + c))*m */
      const float d = (c.x + (c.y + c.z)) * mask_fade;
      const float m0 = slw * mask;
      const float w = m0 + d * (slwb - m0);
      const o_vec4 o = {c.x * w, c.y * w, c.z * w, 1.0f};
      store_px(a, x, y, o);
    }
}
void o_pass_zfast_crt(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  o_pass_zfast_crt_body(a);
  o_fp_leave(csr);
}

/* crt/shaders/crt-easymode.glsl (crt/crt-easymode.glslp), ENABLE_LANCZOS 1; FS 159-268.  17 params in pragma order:
 * SHARPNESS_H, SHARPNESS_V, MASK_STRENGTH, MASK_DOT_WIDTH, MASK_DOT_HEIGHT, MASK_STAGGER, MASK_SIZE, SCANLINE_STRENGTH,
 * SCANLINE_BEAM_WIDTH_MIN, SCANLINE_BEAM_WIDTH_MAX, SCANLINE_BRIGHT_MIN, SCANLINE_BRIGHT_MAX, SCANLINE_CUTOFF,
 * GAMMA_INPUT, GAMMA_OUTPUT, BRIGHT_BOOST, DILATION */
static float em_curve(float x, float sharp) {
  const float x_step = x < 0.5f ? 0.0f : 1.0f;
  const float h = 0.5f - x;
  const float sg = h > 0.0f ? 1.0f : (h < 0.0f ? -1.0f : 0.0f);
  const float curve = 0.5f - sqrtf(0.25f - (x - x_step) * (x - x_step)) * sg;
  return x + sharp * (curve - x);
}
static o_vec4 em_tex(const o_tex* t, float u, float v, float dil) {
  const o_vec4 c = o_sample(t, u, v);
  /* dilate(): col * mix(1.0, col, DILATION) with a uniform weight: a*(1 - t) + b*t (in-situ float probe) */
  const float om = 1.0f - dil;
  const o_vec4 r = {c.x * (om + c.x * dil), c.y * (om + c.y * dil), c.z * (om + c.z * dil), c.w * (om + c.w * dil)};
  return r;
}
static void em_lanczos(const o_tex* t, float u, float v, float dx, const float* k, float dil, float* out3) {
  const o_vec4 m0 = em_tex(t, u - dx, v - 0.0f, dil), m1 = em_tex(t, u, v, dil), m2 = em_tex(t, u + dx, v + 0.0f, dil);
  const o_vec4 m3 = em_tex(t, u + 2.0f * dx, v + 2.0f * 0.0f, dil);
  const float* p0 = &m0.x; const float* p1 = &m1.x; const float* p2 = &m2.x; const float* p3 = &m3.x;
  for (int c = 0; c < 3; ++c) {
    const float col = ((p0[c] * k[0] + p1[c] * k[1]) + p2[c] * k[2]) + p3[c] * k[3];
    const float mn = p1[c] < p2[c] ? p1[c] : p2[c], mx = p1[c] > p2[c] ? p1[c] : p2[c];
    const float lo = col > mn ? col : mn;
    out3[c] = lo < mx ? lo : mx;
  }
}
static void o_pass_crt_easymode_body(const o_pass_args* a) {
  const int W = a->out_w, H = a->out_h;
  const float* P = a->params;
  const float sh = P[0], sv = P[1], mstr = P[2], mdw = P[3], mdh = P[4], mstag = P[5], msize = P[6], sstr = P[7];
  const float bwmin = P[8], bwmax = P[9], brmin = P[10], brmax = P[11], cutoff = P[12], gin = P[13], gout = P[14], boost = P[15], dil = P[16];
  const float tsx = (float)a->in->w, tsy = (float)a->in->h; /* TextureSize == InputSize */
  const float idx = 1.0f / tsx, idy = 1.0f / tsy;
  const float pi = 3.141592653589f;
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      const float pcx = u * tsx - 0.5f, pcy = v * tsy - 0.5f;
      const float tcx = (floorf(pcx) + 0.5f) * idx, tcy = (floorf(pcy) + 0.5f) * idy;
      const float dsx = pcx - floorf(pcx), dsy = pcy - floorf(pcy);
      const float cx = em_curve(dsx, sh * sh);
      float k[4] = {pi * (1.0f + cx), pi * cx, pi * (1.0f - cx), pi * (2.0f - cx)};
      for (int q = 0; q < 4; ++q) {
        float c = fabsf(k[q]);
        c = c > 1e-5f ? c : 1e-5f;
        k[q] = ((2.0f * o_sin(c)) * o_sin(c * 0.5f)) / (c * c);
      }
      const float ksum = k[0] + (k[1] + (k[2] + k[3]));   /* dot(coeffs, vec4(1.0)) */
      for (int q = 0; q < 4; ++q) k[q] = k[q] / ksum;
      float c1[3], c2[3], col[3];
      em_lanczos(a->in, tcx, tcy, idx, k, dil, c1);
      em_lanczos(a->in, tcx + 0.0f, tcy + idy, idx, k, dil, c2);
      const float cy = em_curve(dsy, sv);
      const float ge = gin / (dil + 1.0f);
      for (int c = 0; c < 3; ++c) col[c] = o_pow(c1[c] + cy * (c2[c] - c1[c]), ge);
      const float luma = 0.2126f * col[0] + (0.7152f * col[1] + 0.0722f * col[2]);
      const float gb = col[1] > col[2] ? col[1] : col[2];
      const float mxc = col[0] > gb ? col[0] : gb;
      const float bright = (mxc + luma) * 0.5f;
      float scan_bright = bright > brmin ? bright : brmin;
      scan_bright = scan_bright < brmax ? scan_bright : brmax;
      float scan_beam = bright * bwmax;
      scan_beam = scan_beam > bwmin ? scan_beam : bwmin;
      scan_beam = scan_beam < bwmax ? scan_beam : bwmax;
      const float ang = ((v * 2.0f) * pi) * tsy;
      float scan_weight = 1.0f - o_pow(o_cos(ang) * 0.5f + 0.5f, scan_beam) * sstr;
      const float mask = 1.0f - mstr;
      const float mfx = floorf(((u * (float)W) * tsx) / (tsx * msize)), mfy = floorf(((v * (float)H) * tsy) / (tsy * (mdh * msize)));
      const float m2 = mfy - 2.0f * floorf(mfy / 2.0f);
      const float q = (mfx + m2 * mstag) / mdw;
      const int dot_no = (int)(q - 3.0f * floorf(q / 3.0f));
      const float mw[3] = {dot_no == 0 ? 1.0f : mask, dot_no == 1 ? 1.0f : mask, (dot_no != 0 && dot_no != 1) ? 1.0f : mask};
      if (tsy >= cutoff) scan_weight = 1.0f;
      float out[3];
      for (int c = 0; c < 3; ++c) {
        const float c0 = col[c] * scan_weight;
        float r = c0 + scan_bright * (col[c] - c0);
        r = r * mw[c];
        out[c] = o_pow(r, 1.0f / gout) * boost;
      }
      const o_vec4 o = {out[0], out[1], out[2], 1.0f};
      store_px(a, x, y, o);
    }
}
void o_pass_crt_easymode(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  o_pass_crt_easymode_body(a);
  o_fp_leave(csr);
}

/* crt/shaders/crt-nes-mini.glsl (crt/crt-nes-mini.glslp), VS 38-43, FS 94-105.  params: SCANTHICK, INTENSITY, BRIGHTBOOST;
 * BRIGHTBOOST is one of the names the reference overwrites on every draw (ShaderEngine.cpp:2278-2282): it always
 * arrives as 1.25, not the shader's 0.15. */
static void o_pass_crt_nes_mini_body(const o_pass_args* a) {
  const int W = a->out_w, H = a->out_h;
  const float thick = a->params[0], inten = a->params[1], boost = a->params[2];
  const float tsy = (float)a->in->h;
  o_varying tu = o_varying_setup(0.f * 1.00001f, 1.f * 1.00001f, 1.f * 1.00001f, 0.f * 1.00001f, W, H, a->out_fmt);
  o_varying tv = o_varying_setup(0.f * 1.00001f, 0.f * 1.00001f, 1.f * 1.00001f, 1.f * 1.00001f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      const o_vec4 t = o_sample(a->in, u, v);
      const float sy0 = (v * thick) * tsy;
      const float sel = sy0 - 2.0f * floorf(sy0 / 2.0f);
      const float hi = sel < 1.0f ? 0.0f : 1.0f, lw = 1.0f - hi;
      const float t3[3] = {t.x, t.y, t.z};
      float out[3];
      for (int c = 0; c < 3; ++c) {
        const float ph = ((1.0f + boost) - 0.2f * t3[c]) * t3[c];
        const float pl = ((1.0f - inten) + 0.1f * t3[c]) * t3[c];
        out[c] = lw * pl + hi * ph;
      }
      const o_vec4 o = {out[0], out[1], out[2], 1.0f};
      store_px(a, x, y, o);
    }
}
void o_pass_crt_nes_mini(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  o_pass_crt_nes_mini_body(a);
  o_fp_leave(csr);
}

/* interpolation/shaders/quilez.glsl (FS 87-102) and interpolation/shaders/sharp-bilinear.glsl (FS 104-121; params
 * SHARP_BILINEAR_PRE_SCALE, AUTO_PRESCALE): a modified coordinate, then one sample with the input's own filter. */
static void o_pass_interp_body(const o_pass_args* a, int sharp) {
  const int W = a->out_w, H = a->out_h;
  const float tsx = (float)a->in->w, tsy = (float)a->in->h, idx = 1.0f / tsx, idy = 1.0f / tsy;
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  float scale = 1.0f, range = 0.0f;
  if (sharp == 1) {
    scale = a->params[1] > 0.5f ? floorf((float)H / tsy + 0.01f) : a->params[0];   /* InputSize.y == TextureSize.y */
    range = 0.5f - 0.5f / scale;
  }
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      float qx, qy;
      if (sharp != 1) {
        const float px = u * tsx + 0.5f, py = v * tsy + 0.5f;
        const float ix = floorf(px), iy = floorf(py);
        float fx = px - ix, fy = py - iy;
        if (sharp == 2) {   /* smootheststep.glsl FS 104: f*f*f*f*(f*(f*(-20 f + 70) - 84) + 35) */
          fx = (((fx * fx) * fx) * fx) * (fx * (fx * (-20.0f * fx + 70.0f) - 84.0f) + 35.0f);
          fy = (((fy * fy) * fy) * fy) * (fy * (fy * (-20.0f * fy + 70.0f) - 84.0f) + 35.0f);
        } else {
          fx = ((fx * fx) * fx) * (fx * (fx * 6.0f - 15.0f) + 10.0f);
          fy = ((fy * fy) * fy) * (fy * (fy * 6.0f - 15.0f) + 10.0f);
        }
        qx = ((ix + fx) - 0.5f) * idx;
        qy = ((iy + fy) - 0.5f) * idy;
      } else {
        const float tx = u * tsx, ty = v * tsy;
        const float flx = floorf(tx), fly = floorf(ty);
        const float cdx = (tx - flx) - 0.5f, cdy = (ty - fly) - 0.5f;
        const float clx = fminf(fmaxf(cdx, -range), range), cly = fminf(fmaxf(cdy, -range), range);
        const float fx = (cdx - clx) * scale + 0.5f, fy = (cdy - cly) * scale + 0.5f;
        qx = (flx + fx) / tsx;
        qy = (fly + fy) / tsy;
      }
      o_vec4 c = o_sample(a->in, qx, qy);
      if (sharp != 0) c.w = 1.0f;   /* quilez writes vec4(texture), the other two vec4(rgb, 1.0) */
      store_px(a, x, y, c);
    }
}
void o_pass_quilez(const o_pass_args* a) { unsigned csr = o_fp_enter(); o_pass_interp_body(a, 0); o_fp_leave(csr); }
void o_pass_smootheststep(const o_pass_args* a) { unsigned csr = o_fp_enter(); o_pass_interp_body(a, 2); o_fp_leave(csr); }
void o_pass_sharp_bilinear(const o_pass_args* a) { unsigned csr = o_fp_enter(); o_pass_interp_body(a, 1); o_fp_leave(csr); }

/* scalenx/shaders/epx.glsl (scalenx/epx.glslp: NEAREST, source x 2), FS 97-136: EPX / Scale2x selection rules. */
static int epx_same(o_vec4 a, o_vec4 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
static void o_pass_epx_body(const o_pass_args* a) {
  const int W = a->out_w, H = a->out_h;
  const float tsx = (float)a->in->w, tsy = (float)a->in->h, idx = 1.0f / tsx, idy = 1.0f / tsy;
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      const o_vec4 P = o_sample(a->in, u + 0.0f * idx, v + 0.0f * idy);
      const o_vec4 A = o_sample(a->in, u + 0.0f * idx, v + 1.0f * idy), B = o_sample(a->in, u + 1.0f * idx, v + 0.0f * idy);
      const o_vec4 D = o_sample(a->in, u + 0.0f * idx, v + -1.0f * idy), C = o_sample(a->in, u + -1.0f * idx, v + 0.0f * idy);
      const o_vec4 one = (epx_same(C, D) && !epx_same(C, A) && !epx_same(C, B)) ? C : P;
      const o_vec4 two = (epx_same(D, B) && !epx_same(D, C) && !epx_same(D, A)) ? D : P;
      const o_vec4 three = (epx_same(A, C) && !epx_same(A, B) && !epx_same(A, D)) ? A : P;
      const o_vec4 four = (epx_same(B, A) && !epx_same(B, D) && !epx_same(B, C)) ? B : P;
      float pxx = u * tsx, pxy = v * tsy;
      pxx = pxx - floorf(pxx);
      pxy = pxy - floorf(pxy);
      o_vec4 o = pxx < 0.5f ? (pxy < 0.5f ? one : three) : (pxy < 0.5f ? two : four);
      o.w = 1.0f;
      store_px(a, x, y, o);
    }
}
void o_pass_epx(const o_pass_args* a) { unsigned csr = o_fp_enter(); o_pass_epx_body(a); o_fp_leave(csr); }

/* handheld/shaders/lcd3x.glsl (handheld/lcd3x.glslp), FS 95-110.  params: brighten_scanlines, brighten_lcd */
static void o_pass_lcd3x_body(const o_pass_args* a) {
  const int W = a->out_w, H = a->out_h;
  const float bs = a->params[0], bl = a->params[1];
  const float pi = 3.141592654f;
  const float off[3] = {pi * (1.0f / 2.0f), pi * (1.0f / 2.0f - 2.0f / 3.0f), pi * (1.0f / 2.0f - 4.0f / 3.0f)};
  const float omx = (pi * 2.0f) * (float)a->in->w, omy = (pi * 2.0f) * (float)a->in->h;
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      const o_vec4 r = o_sample(a->in, u, v);
      const float ax = u * omx, ay = v * omy;
      const float yf = (bs + o_sin(ay)) / (bs + 1.0f);
      const float r3[3] = {r.x, r.y, r.z};
      float out[3];
      for (int c = 0; c < 3; ++c) out[c] = (yf * ((bl + o_sin(ax + off[c])) / (bl + 1.0f))) * r3[c];
      const o_vec4 o = {out[0], out[1], out[2], 1.0f};
      store_px(a, x, y, o);
    }
}
void o_pass_lcd3x(const o_pass_args* a) { unsigned csr = o_fp_enter(); o_pass_lcd3x_body(a); o_fp_leave(csr); }

/* dithering/shaders/bayer-matrix-dithering.glsl (dithering/bayer-matrix-dithering.glslp), FS 99-141: 8x8 ordered
 * dithering of every channel to 0 / 1.  params: animate, dither_size; FrameCount is an int uniform. */
static const int k_bayer8[8][8] = {{0, 32, 8, 40, 2, 34, 10, 42}, {48, 16, 56, 24, 50, 18, 58, 26}, {12, 44, 4, 36, 14, 46, 6, 38},
                                   {60, 28, 52, 20, 62, 30, 54, 22}, {3, 35, 11, 43, 1, 33, 9, 41}, {51, 19, 59, 27, 49, 17, 57, 25},
                                   {15, 47, 7, 39, 13, 45, 5, 37}, {63, 31, 55, 23, 61, 29, 53, 21}};
static void o_pass_bayer_body(const o_pass_args* a) {
  const int W = a->out_w, H = a->out_h;
  const float animate = a->params[0], dsize = a->params[1];
  const float fc2 = 2.0f * (float)a->frame_count;
  const float scale = (3.0f + (fc2 - 32.0f * floorf(fc2 / 32.0f)) * animate) + dsize;
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      const o_vec4 c = o_sample(a->in, u, v);
      const float xx = (u * (float)W) * scale, yy = (v * (float)H) * scale;
      const int ix = (int)(xx - 8.0f * floorf(xx / 8.0f)), iy = (int)(yy - 8.0f * floorf(yy / 8.0f));
      float limit = 0.0f;
      if (ix < 8) limit = (float)(k_bayer8[ix & 7][iy & 7] + 1) / 64.0f;
      const o_vec4 o = {c.x < limit ? 0.0f : 1.0f, c.y < limit ? 0.0f : 1.0f, c.z < limit ? 0.0f : 1.0f, 1.0f};
      store_px(a, x, y, o);
    }
}
void o_pass_bayer(const o_pass_args* a) { unsigned csr = o_fp_enter(); o_pass_bayer_body(a); o_fp_leave(csr); }

/* handheld/shaders/lcd1x.glsl (handheld/lcd1x.glslp: NEAREST, clamp_to_border), FS 104-121.
 * params: BRIGHTEN_SCANLINES, BRIGHTEN_LCD */
static void o_pass_lcd1x_body(const o_pass_args* a) {
  const int W = a->out_w, H = a->out_h;
  const float bs = a->params[0], bl = a->params[1];
  const float two_pi = 2.0f * 3.141592654f;
  const float tsx = (float)a->in->w, tsy = (float)a->in->h;
  o_varying tu = o_varying_setup(0.f * 1.0001f, 1.f * 1.0001f, 1.f * 1.0001f, 0.f * 1.0001f, W, H, a->out_fmt);
  o_varying tv = o_varying_setup(0.f * 1.0001f, 0.f * 1.0001f, 1.f * 1.0001f, 1.f * 1.0001f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      const float ax = two_pi * (u * tsx - 0.25f), ay = two_pi * (v * tsy - 0.25f);
      const float yf = (bs + o_sin(ay)) / (bs + 1.0f), xf = (bl + o_sin(ax)) / (bl + 1.0f);
      const o_vec4 c = o_sample(a->in, u, v);
      const float k = yf * xf;
      const o_vec4 o = {k * c.x, k * c.y, k * c.z, 1.0f};
      store_px(a, x, y, o);
    }
}
void o_pass_lcd1x(const o_pass_args* a) { unsigned csr = o_fp_enter(); o_pass_lcd1x_body(a); o_fp_leave(csr); }

/* stereoscopic-3d/shaders/shutter-3d.glsl (stereoscopic-3d/shutter-to-side-by-side.glslp): VS 61-73, FS 123-143; operation
 * order from the GL's final instruction listing (LP_DEBUG=fs / GALLIVM_DEBUG=tgsi).  params: ZOOM, vert_pos, horz_pos,
 * separation, flicker, height_mod, swap_eye; extra[0] = PrevTexture.  The left / right eye coordinates are varyings; timer
 * = |swap_eye - mod(FrameCount, 2)| is the same at every vertex. */
static void shutter_3d_vertex(const float* P, float isx, float isy, float tsx, float tsy, float tcx, float tcy, float* lr) {
  const float hx = (0.5f * isx) / tsx, hy = (0.5f * isy) / tsy;
  const float tx = tcx + -hx, ty = tcy + -hy;
  float x = (tx * 2.0f) * P[0] + P[2], y = (ty * P[0]) * (1.0f / P[5]) + P[1];
  x = x + hx;
  y = y + hy;
  const float sx = ((0.5f + P[3]) * isx) / tsx, sy = 0.0f / tsy;
  lr[0] = x + -sx; lr[1] = y + -sy; lr[2] = x + sx; lr[3] = y + sy;
}
void o_pass_shutter_3d(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  const float* P = a->params;
  const float isx = (float)a->in->w, isy = (float)a->in->h, tsx = isx, tsy = isy;
  float v[4][4];   /* BL, BR, TR, TL */
  static const float tc[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
  for (int k = 0; k < 4; ++k) shutter_3d_vertex(P, isx, isy, tsx, tsy, tc[k][0], tc[k][1], v[k]);
  o_varying pl[4];
  for (int c = 0; c < 4; ++c) pl[c] = o_varying_setup(v[0][c], v[1][c], v[2][c], v[3][c], W, H, a->out_fmt);
  const float fc = (float)a->frame_count;
  const float timer = fabsf(P[6] + -(fc + -(2.0f * floorf(fc / 2.0f))));
  const float omt = 1.0f + -timer, flicker = P[4];
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float lx = o_varying_at(&pl[0], x, y, lo), ly = o_varying_at(&pl[1], x, y, lo);
      const float rx = o_varying_at(&pl[2], x, y, lo), ry = o_varying_at(&pl[3], x, y, lo);
      const o_vec4 L = o_sample(a->in, lx, ly), R = o_sample(a->in, rx, ry), Rh = o_sample(a->extra[0], rx, ry), Lh = o_sample(a->extra[0], lx, ly);
      const float l4[4] = {L.x, L.y, L.z, L.w}, lh4[4] = {Lh.x, Lh.y, Lh.z, Lh.w}, r4[4] = {R.x, R.y, R.z, R.w}, rh4[4] = {Rh.x, Rh.y, Rh.z, Rh.w};
      const float lcx = (lx * isx) / tsx, lcy = (ly * isy) / tsy, rcx = (rx * isx) / tsx, rcy = (ry * isy) / tsy;
      const float lm = (lcy != lcy) ? lcx : (lcx < lcy ? lcx : lcy), rm = (rcy != rcy) ? rcx : (rcx < rcy ? rcx : rcy);
      const float ml = (0.0001f < lm && lcx < 0.9999f && lcy < 0.9999f) ? 1.0f : 0.0f;
      const float mr = (0.0001f < rm && rcx < 0.9999f && rcy < 0.9999f) ? 1.0f : 0.0f;
      float o[4];
      for (int c = 0; c < 4; ++c) {
        const float lc = l4[c] * timer + (omt * lh4[c]) * flicker;
        const float rc = r4[c] * omt + (rh4[c] * timer) * flicker;
        o[c] = lc * ml + rc * mr;
      }
      const o_vec4 out = {o[0], o[1], o[2], o[3]};
      store_px(a, x, y, out);
    }
  o_fp_leave(csr);
}

/* misc/anti-flicker.glsl FS 99-127 (no preset of the reference's tree names it; loaded as a one-pass chain).  params:
 * lum_diff_thresh; extra[0] = PrevTexture, extra[1] = Prev1Texture.  YIQ dot products run blue, green, then red. */
void o_pass_anti_flicker(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  const float th = a->params[0];
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      const o_vec4 c = o_sample(a->in, u, v), p0 = o_sample(a->extra[0], u, v), p1 = o_sample(a->extra[1], u, v);
      const float cy = (c.z * 0.114f + c.y * 0.587f) + c.x * 0.2989f, ci = (c.z * -0.3216f + c.y * -0.2744f) + c.x * 0.5959f;
      const float cq = (c.z * 0.3114f + c.y * -0.5229f) + c.x * 0.2115f;
      const float py = (p0.z * 0.114f + p0.y * 0.587f) + p0.x * 0.2989f, pi = (p0.z * -0.3216f + p0.y * -0.2744f) + p0.x * 0.5959f;
      const float pq = (p0.z * 0.3114f + p0.y * -0.5229f) + p0.x * 0.2115f;
      const float p1y = (p1.z * 0.114f + p1.y * 0.587f) + p1.x * 0.2989f;
      const int blend = (th < fabsf(cy + -py)) && (fabsf(cy + -p1y) < 1.0f + -th);
      const float Y = blend ? (py + cy) / 2.0f : cy, I = blend ? (pi + ci) / 2.0f : ci, Q = blend ? (pq + cq) / 2.0f : cq;
      const o_vec4 out = {(Q * 0.621f + Y) + I * 0.956f, (Q * -0.6474f + Y) + I * -0.272f, (Q * 1.7046f + Y) + I * -1.106f, 1.0f};
      store_px(a, x, y, out);
    }
  o_fp_leave(csr);
}

/* handheld/shaders/color/{gba,gbc,nds,palm,psp,vba}-color.glsl (handheld/<name>-color.glslp and the lcd-grid-v2-* chains): one
 * structure, FS main: pow(texel, gamma_in) * lum, clamp, a constant 3x3 matrix (the shader's `color * adjust` with sat = 1 and
 * contrast = 1 folds to the plain coefficients), pow(.., 1 / display_gamma), alpha 0.  What differs, from the GL's final
 * instruction listings: gamma_in = target_gamma + darken_screen (gba), - lighten_screen (gbc), + darken_screen * 1.7 (vba),
 * a constant elsewhere; a zero coefficient drops its term (vba red); psp's blue row has two equal coefficients and runs
 * as 0.01 * (R + G) + 0.98 * B. */
typedef struct { float ga, gs, lum, m[3][3], inv; int factored_blue; } color_spec;
static void color_body(const o_pass_args* a, const color_spec* s, float user) {
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  const float gin = s->ga + user * s->gs;
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const o_vec4 t = o_sample(a->in, o_varying_at(&tu, x, y, lo), o_varying_at(&tv, x, y, lo));
      float c[3] = {o_pow(t.x, gin) * s->lum, o_pow(t.y, gin) * s->lum, o_pow(t.z, gin) * s->lum}, o[3];
      for (int k = 0; k < 3; ++k) {
        c[k] = c[k] > 0.0f ? c[k] : 0.0f;      /* fmax(x, 0) then fmin(., 1): MAXPS / MINPS */
        c[k] = c[k] < 1.0f ? c[k] : 1.0f;
      }
      for (int k = 0; k < 3; ++k) {
        float v;
        if (k == 2 && s->factored_blue) v = s->m[2][0] * (c[0] + c[1]);
        else v = s->m[k][0] * c[0] + s->m[k][1] * c[1];
        if (s->m[k][2] != 0.0f) v = v + s->m[k][2] * c[2];
        o[k] = o_pow(v, s->inv);
      }
      const o_vec4 out = {o[0], o[1], o[2], 0.0f};
      store_px(a, x, y, out);
    }
  o_fp_leave(csr);
}
static const color_spec k_gba_color = {2.2f, 1.0f, 0.94f, {{0.82f, 0.24f, -0.06f}, {0.125f, 0.665f, 0.21f}, {0.195f, 0.075f, 0.73f}}, 1.0f / 2.2f, 0};
static const color_spec k_gbc_color = {2.2f, -1.0f, 0.94f, {{0.82f, 0.24f, -0.06f}, {0.125f, 0.665f, 0.21f}, {0.195f, 0.075f, 0.73f}}, 1.0f / 2.2f, 0};
static const color_spec k_nds_color = {1.91f, 0.0f, 0.89f, {{0.87f, 0.255f, -0.125f}, {0.10f, 0.645f, 0.255f}, {0.10f, 0.17f, 0.73f}}, 1.0f / 1.91f, 0};
static const color_spec k_palm_color = {2.2f, 0.0f, 1.0f, {{0.83f, 0.26f, -0.09f}, {0.073f, 0.677f, 0.25f}, {0.085f, 0.12f, 0.795f}}, 1.0f / 2.2f, 0};
static const color_spec k_psp_color = {2.21f, 0.0f, 1.0f, {{0.98f, 0.20f, -0.18f}, {0.04f, 0.795f, 0.165f}, {0.01f, 0.01f, 0.98f}}, 1.0f / 2.2f, 1};
static const color_spec k_vba_color = {1.45f, 1.7f, 1.0f, {{0.73f, 0.27f, 0.0f}, {0.085f, 0.675f, 0.24f}, {0.085f, 0.24f, 0.675f}}, 1.0f / 1.45f, 0};
void o_pass_gba_color(const o_pass_args* a) { color_body(a, &k_gba_color, a->params[0]); }   /* 1 param: darken_screen */
void o_pass_gbc_color(const o_pass_args* a) { color_body(a, &k_gbc_color, a->params[0]); }   /* 1 param: lighten_screen */
void o_pass_vba_color(const o_pass_args* a) { color_body(a, &k_vba_color, a->params[0]); }   /* 1 param: darken_screen */
void o_pass_nds_color(const o_pass_args* a) { color_body(a, &k_nds_color, 0.0f); }
void o_pass_palm_color(const o_pass_args* a) { color_body(a, &k_palm_color, 0.0f); }
void o_pass_psp_color(const o_pass_args* a) { color_body(a, &k_psp_color, 0.0f); }

/* handheld/shaders/color/gbc-gambatte-color.glsl FS main: a fixed matrix on the texel as sampled, products taken blue, green, red;
 * alpha passes through. */
void o_pass_gbc_gambatte_color(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const o_vec4 t = o_sample(a->in, o_varying_at(&tu, x, y, lo), o_varying_at(&tv, x, y, lo));
      const float g8 = t.y * 0.125f;
      const o_vec4 out = {(t.z * 0.0625f + g8) + t.x * 0.8125f, t.z * 0.25f + t.y * 0.75f, (t.z * 0.6875f + g8) + t.x * 0.1875f, t.w};
      store_px(a, x, y, out);
    }
  o_fp_leave(csr);
}

/* handheld/shaders/retro-v2.glsl FS main (handheld/retro-v2.glslp, presets/retro-v2+<console>-color.glslp): pow(texel, 2.4), a
 * border of width depending on RETRO_PIXEL_SIZE darkens each source pixel's edge, pow(.., 1/2.2), clamp.  Operation order from
 * the GL's final instruction listing.  params: RETRO_PIXEL_SIZE. */
static inline float nmin(float a, float b) { return b != b ? a : (a < b ? a : b); }   /* gallivm's fmin / fmax: the operand that is not NaN */
static inline float nmax(float a, float b) { return b != b ? a : (a > b ? a : b); }
void o_pass_retro_v2(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  const float rps = a->params[0], tsx = (float)a->in->w, tsy = (float)a->in->h;
  const float px = tsx * (1.0f / (float)W), py = tsy * (1.0f / (float)H);   /* InputSize * (1 / OutputSize) */
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      const o_vec4 t = o_sample(a->in, u, v);
      const float su = u * tsx, sv = v * tsy, fx = su - floorf(su), fy = sv - floorf(sv);
      const float ax = nmin(nmax(fx + 0.5f * px, 0.0f), 1.0f), ay = nmin(nmax(fy + 0.5f * py, 0.0f), 1.0f);
      const float cx = nmin(nmax(ax + -rps, 0.0f), px) / px, cy = nmin(nmax(ay + -rps, 0.0f), py) / py;
      const float m = nmax(cx, cy);
      const float k = (1.04f + fx * fy) * (1.0f + -m) + 0.36f * m;
      const o_vec4 out = {nmin(nmax(o_pow(k * o_pow(t.x, 2.4f), 1.0f / 2.2f), 0.0f), 1.0f), nmin(nmax(o_pow(k * o_pow(t.y, 2.4f), 1.0f / 2.2f), 0.0f), 1.0f),
                          nmin(nmax(o_pow(k * o_pow(t.z, 2.4f), 1.0f / 2.2f), 0.0f), 1.0f), 1.0f};
      store_px(a, x, y, out);
    }
  o_fp_leave(csr);
}

/* handheld/shaders/mgba/agb001.glsl FS main (handheld/agb001.glslp, agb001-gba-color-motionblur.glslp): pow(texel * 0.8, 1.8) + 0.16,
 * a 4x4 subpixel pattern per source texel - column 0 / 1 / 2 keeps red / green / blue and takes the others to 0.2, column 3 takes
 * all to 0.4, row 3 another 0.8 - alpha 0.5.  The pattern index is int(mod(coord * size * 4, 4)) with mod as a - 4 floor(a / 4). */
void o_pass_agb001(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  const float tsx = (float)a->in->w, tsy = (float)a->in->h;
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      const o_vec4 t = o_sample(a->in, u, v);
      float c[3] = {o_pow(t.x * 0.8f, 1.8f) + 0.16f, o_pow(t.y * 0.8f, 1.8f) + 0.16f, o_pow(t.z * 0.8f, 1.8f) + 0.16f};
      const float ax = (u * tsx) * 4.0f, ay = (v * tsy) * 4.0f;
      const float mx = ax + -(4.0f * floorf(ax / 4.0f)), my = ay + -(4.0f * floorf(ay / 4.0f));
      const int ix = mx != mx ? (-2147483647 - 1) : (int)mx, iy = my != my ? (-2147483647 - 1) : (int)my;
      for (int k = 0; k < 3; ++k) {
        if (ix >= 0 && ix <= 2) { if (k != ix) c[k] = c[k] * 0.2f; }
        else c[k] = c[k] * 0.4f;
        if (!((unsigned)iy <= 2u)) c[k] = c[k] * 0.8f;
      }
      const o_vec4 out = {c[0], c[1], c[2], 0.5f};
      store_px(a, x, y, out);
    }
  o_fp_leave(csr);
}

/* handheld/console-border/shader-files/gb-pass-5.glsl (the last pass of the 75 console-border presets): the frame, scaled about its
 * centre, under a border image that is blended in by its own alpha.  VS 46-58 in the GL's operation order; FS: frame + a (border - frame).
 * params: SCALE, OUT_X, OUT_Y; extra[0] = BORDER. */
void o_pass_gb_pass_5(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  const float* P = a->params;
  const float osx = (float)W, osy = (float)H, isx = (float)a->in->w, isy = (float)a->in->h, tsx = isx;
  /* the pass sits at index 3 in the 4-pass presets, where the reference hands TextureSize.y the TARGET's height (ShaderEngine.cpp:2418-2421) */
  const float tsy = (a->pass_index == 3 && H != a->in->h) ? (float)H : isy;
  const float scx = (osx / isx) / P[0], scy = (osy / isy) / P[0];
  const float mx = (0.5f * isx) / tsx, my = (0.5f * isy) / tsy;
  float v[4][4];
  static const float tc[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
  for (int k = 0; k < 4; ++k) {
    const float tx = mx + (tc[k][0] + -mx) * scx, ty = my + (tc[k][1] + -my) * scy;
    const float bx = tx * (tsx / isx) + -0.5f, by = ty * (tsy / isy) + -0.5f;
    v[k][0] = tx; v[k][1] = ty;
    v[k][2] = 0.5f + ((bx * osx) / P[1]) / scx;
    v[k][3] = 0.5f + ((by * osy) / P[2]) / scy;
  }
  o_varying pl[4];
  for (int c = 0; c < 4; ++c) pl[c] = o_varying_setup(v[0][c], v[1][c], v[2][c], v[3][c], W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const o_vec4 b = o_sample(a->extra[0], o_varying_at(&pl[2], x, y, lo), o_varying_at(&pl[3], x, y, lo));
      const o_vec4 f = o_sample(a->in, o_varying_at(&pl[0], x, y, lo), o_varying_at(&pl[1], x, y, lo));
      const o_vec4 out = {f.x + b.w * (b.x + -f.x), f.y + b.w * (b.y + -f.y), f.z + b.w * (b.z + -f.z), f.w + b.w * (b.w + -f.w)};
      store_px(a, x, y, out);
    }
  o_fp_leave(csr);
}

/* borders/resources/imgborder-{sgb,gameboy-player,sgba}.glsl (one text, three sets of #pragma defaults; borders/sgb/, gameboy-player/,
 * sgba/ presets): the frame placed inside a border image, which covers it by its own alpha - or not at all inside the viewport when
 * border_on_top is set.  VS 93-103 / FS 165-173 in the GL's operation order.
 * params: box_scale, location_x, location_y, in_res_x, in_res_y, border_on_top, border_zoom_x, border_zoom_y, OS_MASK_TOP, OS_MASK_BOTTOM,
 * OS_MASK_LEFT, OS_MASK_RIGHT; extra[0] = BORDER. */
void o_pass_imgborder(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  const float* P = a->params;
  const float osx = (float)W, osy = (float)H, isx = (float)a->in->w, isy = (float)a->in->h, tsx = isx;
  const float tsy = (a->pass_index == 3 && H != a->in->h) ? (float)H : isy;   /* the reference's TextureSize.y rule for pass index 3 */
  const float mx = (P[1] * isx) / tsx, my = (P[2] * isy) / tsy;
  const float scx = (osx / P[3]) / P[0], scy = (osy / P[4]) / P[0];
  const float rx = tsx / isx, ry = tsy / isy;
  float v[4][4];
  static const float tc[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
  for (int k = 0; k < 4; ++k) {
    v[k][0] = mx + (tc[k][0] + -mx) * scx;                       /* screen_coord */
    v[k][1] = my + (tc[k][1] + -my) * scy;
    v[k][2] = 0.4999f + (tc[k][0] * rx + -0.4999f) * P[6];       /* TEX0: the border image, zoomed about its centre */
    v[k][3] = 0.4999f + (tc[k][1] * ry + -0.4999f) * P[7];
  }
  o_varying pl[4];
  for (int c = 0; c < 4; ++c) pl[c] = o_varying_setup(v[0][c], v[1][c], v[2][c], v[3][c], W, H, a->out_fmt);
  const float x_hi = 0.9999f + -P[11], x_lo = 0.0001f + P[10], y_hi = 0.9999f + -P[9], y_lo = 0.0001f + P[8];
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float sx = o_varying_at(&pl[0], x, y, lo), sy = o_varying_at(&pl[1], x, y, lo);
      const o_vec4 f = o_sample(a->in, sx, sy);
      const o_vec4 b = o_sample(a->extra[0], o_varying_at(&pl[2], x, y, lo), o_varying_at(&pl[3], x, y, lo));
      const int inside = sx < x_hi && x_lo < sx && sy < y_hi && y_lo < sy && 0.5f < P[5];
      const float al = inside ? 0.0f : b.w;
      const o_vec4 out = {f.x + al * (b.x + -f.x), f.y + al * (b.y + -f.y), f.z + al * (b.z + -f.z), f.w + al * (al + -f.w)};
      store_px(a, x, y, out);
    }
  o_fp_leave(csr);
}

/* reshade/shaders/LUT/LUT.glsl FS main (reshade/{lut,gba,nds,vba,bsnes-gamma-ramp,spfft}.glslp): a colour LUT of LUT_Size slices laid side
 * by side, sampled twice (LINEAR) and mixed along blue.  As written, `ceil(b + 0.000001 * (LUT_Size - 1.0))` rounds the colour itself up,
 * so the second slice is slice 1 (or 0 for b = 0) - kept.  The mix runs only where the first sample's blue is below 1.
 * params: LUT_Size; extra[0] = SamplerLUT. */
void o_pass_lut(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  const float S = a->params[0], k = S + -1.0f;
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const o_vec4 c = o_sample(a->in, o_varying_at(&tu, x, y, lo), o_varying_at(&tv, x, y, lo));
      const float red = (c.x * k + 0.4999f) / (S * S), green = (c.y * k + 0.4999f) / S;
      const float b1 = floorf(c.z * k) / S + red, b2 = ceilf(c.z + 0.000001f * k) / S + red;
      const o_vec4 c1 = o_sample(a->extra[0], b1, green), c2 = o_sample(a->extra[0], b2, green);
      float m = (c.z + -b1) / (b2 + -b1);
      m = nmin(nmax(m, 0.0f), 32.0f);
      o_vec4 out = c1;
      if (c1.z < 1.0f) {
        out.x = c1.x + m * (c2.x + -c1.x); out.y = c1.y + m * (c2.y + -c1.y); out.z = c1.z + m * (c2.z + -c1.z); out.w = c1.w + m * (c2.w + -c1.w);
      }
      store_px(a, x, y, out);
    }
  o_fp_leave(csr);
}

/* handheld/console-border/shader-files/border.glsl: imgborder without the four OS_MASK parameters (its viewport test compares with the
 * bare 0.9999 / 0.0001, which is what 0.9999 - 0 and 0.0001 + 0 are).  params: box_scale, location_x, location_y, in_res_x, in_res_y,
 * border_on_top, border_zoom_x, border_zoom_y; extra[0] = BORDER. */
void o_pass_console_border(const o_pass_args* a) {
  float p12[12] = {0};
  for (int k = 0; k < 8; ++k) p12[k] = a->params[k];
  o_pass_args b = *a;
  b.params = p12;
  o_pass_imgborder(&b);
}

/* handheld/shaders/gb-palette/gb-palette.glsl FS main (handheld/gb-palette-{dmg,light,pocket}.glslp): the red channel as a grey level picks
 * a row of a palette image; alpha = ceil(|1 - r|).  extra[0] = COLOR_PALETTE. */
void o_pass_gb_palette(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const o_vec4 c = o_sample(a->in, o_varying_at(&tu, x, y, lo), o_varying_at(&tv, x, y, lo));
      const float g = fabsf(1.0f + -c.x);
      const o_vec4 p = o_sample(a->extra[0], 0.5f, g * 0.75f + 0.125f);
      const o_vec4 out = {p.x, p.y, p.z, ceilf(g)};
      store_px(a, x, y, out);
    }
  o_fp_leave(csr);
}

/* crt/shaders/crt-potato/shader-files/crt-potato.glsl FS main (crt/crt-potato-{cool,warm}.glslp): the frame times a small mask image tiled
 * every 2 target pixels across and every floor(OutputSize.y / InputSize.y + 0.000001) lines down (gl_FragCoord = pixel + 0.5).
 * extra[0] = MASK. */
void o_pass_crt_potato(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  const float scale = floorf((float)H / (float)a->in->h + 0.000001f);
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float fx = ((float)x + 0.5f) / 2.0f, fy = (((float)y + 0.5f) * 1.0f + 0.0f) / scale;
      const o_vec4 m = o_sample(a->extra[0], fx - floorf(fx), fy - floorf(fy));
      const o_vec4 c = o_sample(a->in, o_varying_at(&tu, x, y, lo), o_varying_at(&tv, x, y, lo));
      const o_vec4 out = {m.x * c.x, m.y * c.y, m.z * c.z, m.w * c.w};
      store_px(a, x, y, out);
    }
  o_fp_leave(csr);
}

/* misc/interlacing.glsl FS main (24 presets): every other line of the frame dimmed to `percent`; above 400 source lines the field alternates
 * with FrameCount (enable_480i) and top_field_first shifts it.  The shader the reference's pass-index-3 TextureSize.y rule was written for
 * (ShaderEngine.cpp:2418-2421).  Operation order from the GL's instruction listing.  params: percent, enable_480i, top_field_first. */
void o_pass_interlacing(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  const float* P = a->params;
  const float isy = (float)a->in->h, tsy = (a->pass_index == 3 && H != a->in->h) ? (float)H : isy, fc = (float)a->frame_count;
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      const o_vec4 c = o_sample(a->in, u, v);
      const float line = 400.0f < isy ? (tsy * v + P[2]) + fc * P[1] : (2.000001f * tsy) * v + P[2];
      const float m = line + -(1.99999f * floorf(line / 1.99999f));
      const int keep = 0.99999f < m;
      const o_vec4 out = {keep ? c.x : P[0] * c.x, keep ? c.y : P[0] * c.y, keep ? c.z : P[0] * c.z, keep ? c.w : P[0] * c.w};
      store_px(a, x, y, out);
    }
  o_fp_leave(csr);
}
