/* TEST INFRASTRUCTURE - see rc_oracle.h.
 *
 * Hand restatement of the live arithmetic of the crt-royale preset's twelve passes
 * (reference shaders/shaders_glsl/crt/crt-royale.glslp; GLSL files under
 * crt/shaders/crt-royale/src/ and blurs/blur9fast-{vertical,horizontal}.glsl), for the
 * configuration the reference engine gives them: passes 0-10 have no #pragma parameter, so
 * PARAMETER_UNIFORM is undefined and they run on the static user-settings constants
 * (first-pass file lines 104-473, bind-shader-params #else branch ~1300-1345); pass 11
 * receives its 44 pragma parameters as uniforms.
 *
 * Rules followed (all measured on the GL, see oracle/probes and DESIGN.md):
 *  - operation order = GLSL expression order; nothing fused except inside o_pow/o_exp/...;
 *  - compile-time constant sub-expressions are folded in float with libm (exp(x) const ->
 *    exp2f(x * log2e), sqrt -> sqrtf, plain + - * / in float);
 *  - mix(a,b,t) = a + t*(b-a) for a run-time t, a*(1-t) + b*t for a constant t;
 *  - x/const stays a division; min/max return their second operand when one is NaN;
 *  - vertex-shader outputs are plane-interpolated (rc_varying.c) from the four per-vertex
 *    values, which are computed here in float exactly as the vertex shader computes them.
 */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "rc_oracle.h"

/* ------------------------------------------------------------------ small helpers ---- */
typedef struct { float x, y; } v2;
typedef struct { float x, y, z; } v3;

static inline float minps(float a, float b) { return a < b ? a : b; } /* SSE minps: NaN -> b */
static inline float maxps(float a, float b) { return a > b ? a : b; }
static inline float clampf(float x, float lo, float hi) { return minps(maxps(x, lo), hi); }
static inline float fractf(float x) { return x - floorf(x); }
static inline float modf_glsl(float x, float y) { return x - y * floorf(x / y); }
static inline float mix_rt(float a, float b, float t) { return a + t * (b - a); }
static inline v3 v3s(float s) { v3 r = {s, s, s}; return r; }
static inline v3 rgb(o_vec4 c) { v3 r = {c.x, c.y, c.z}; return r; }

static const float under_half = 0.4995f;
static const float kLog2e = 1.4426950408889634f;
/* exp() of a compile-time constant, as the GL's compiler folds it */
static float const_exp(float x) { return exp2f(x * kLog2e); }

/* is_interlaced() (first-pass file 4723-4750) with interlace_detect = true,
 * interlace_1080i = false */
static int is_interlaced(float num_lines) { return (num_lines > 288.5f) && (num_lines < 576.5f); }

typedef struct { o_varying u, v; } uvplanes;
static uvplanes texcoord_planes(float k, int W, int H, int fmt) {
  uvplanes p;
  p.u = o_varying_setup(0.f * k, 1.f * k, 1.f * k, 0.f * k, W, H, fmt);
  p.v = o_varying_setup(0.f * k, 0.f * k, 1.f * k, 1.f * k, W, H, fmt);
  return p;
}
/* varying whose vertex value is k*TexCoord.x (or .y) with k itself a float computed by the VS */
static o_varying plane_u(float at0, float at1, int W, int H, int fmt) { return o_varying_setup(at0, at1, at1, at0, W, H, fmt); }
static o_varying plane_v(float at0, float at1, int W, int H, int fmt) { return o_varying_setup(at0, at0, at1, at1, W, H, fmt); }

#define ENTER unsigned csr_ = o_fp_enter()
#define LEAVE o_fp_leave(csr_)

/* =========================================================================== P0 ====== */
/* first-pass-linearize-crt-gamma-bob-fields.glsl: VS 4801-4811, FS 4850-4884.
 * FIRST_PASS + SIMULATE_CRT_ON_LCD: decode_input = pow(c, crt_gamma = 2.5); no output gamma. */
void o_pass_royale_first(const o_pass_args* a) {
  ENTER;
  const int W = a->out_w, H = a->out_h;
  const float tsy = (float)a->in->h; /* texture_size == video_size */
  uvplanes tc = texcoord_planes(1.00001f, W, H, a->out_fmt);
  const float uv_step_y = 1.0f / tsy;
  const float interlaced = is_interlaced(tsy) ? 1.0f : 0.0f;
  const float crt_gamma = 2.5f;
  const float interlace_bff = 0.0f;
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      int lo = o_lower_tri(x, y, W, H);
      float u = o_varying_at(&tc.u, x, y, lo), v = o_varying_at(&tc.v, x, y, lo);
      o_vec4 c = o_sample(a->in, u, v), l = o_sample(a->in, u, v - uv_step_y), n = o_sample(a->in, u, v + uv_step_y);
      v3 cur = {o_pow(c.x, crt_gamma), o_pow(c.y, crt_gamma), o_pow(c.z, crt_gamma)};
      v3 last = {o_pow(l.x, crt_gamma), o_pow(l.y, crt_gamma), o_pow(l.z, crt_gamma)};
      v3 next = {o_pow(n.x, crt_gamma), o_pow(n.y, crt_gamma), o_pow(n.z, crt_gamma)};
      v3 interp = {0.5f * (last.x + next.x), 0.5f * (last.y + next.y), 0.5f * (last.z + next.z)};
      float modulus = interlaced + 1.0f;
      float field_offset = modf_glsl((float)a->frame_count + interlace_bff, modulus);
      float curr_line_texel = v * tsy;
      float line_num_last = floorf(curr_line_texel - under_half);
      float wrong_field = modf_glsl(line_num_last + field_offset, modulus);
      o_vec4 o = {mix_rt(cur.x, interp.x, wrong_field), mix_rt(cur.y, interp.y, wrong_field),
                  mix_rt(cur.z, interp.z, wrong_field), 1.0f};
      o_store_pixel(a, x, y, o);
    }
  LEAVE;
}

/* =========================================================================== P1 ====== */
/* scanlines-vertical-interlacing.glsl: VS 5912-5938, FS 5982-6141; beam functions 4775-4998,
 * gamma_impl 3907-3918, get_last_scanline_uv (same text as first-pass file 4693-4721). */
static const float beam_min_sigma = 0.02f, beam_max_sigma = 0.3f;
static const float beam_min_shape = 2.0f, beam_max_shape = 4.0f;

static v2 last_scanline_uv(float u, float v, float tsx, float tsy, float tix, float tiy, float il_y,
                           float frame_count, float* dist) {
  const float interlace_bff = 0.0f;
  float field_offset = floorf(il_y * 0.75f) * modf_glsl(frame_count + interlace_bff, 2.0f);
  float ctx = u * tsx, cty = v * tsy;
  float ptx = floorf(ctx - under_half), pty = floorf(cty - under_half);
  float wrong_field = modf_glsl(pty + field_offset, il_y);
  float snx = ptx - 0.0f, sny = pty - wrong_field;
  float stx = snx + 0.5f, sty = sny + 0.5f;
  v2 uv = {stx * tix, sty * tiy};
  *dist = (cty - sty) / il_y;
  return uv;
}

static float gamma_impl1(float s, float s_inv) {
  const float g = 1.12906830989f, c0 = 0.8109119309638332633713423362694399653724431f;
  const float c1 = 0.4808354605142681877121661197951496120000040f, e = 2.71828182845904523536028747135266249775724709f;
  float sph = s + 0.5f;
  float lanczos_sum = c0 + c1 / (s + 1.0f);
  /* (s + 0.5 + g) / e: the two additive constants fold, (s + (0.5 + g)) / e (in-situ float probe, bit-identical) */
  float base = (s + (0.5f + g)) / e;
  return (o_pow(base, sph) * lanczos_sum) * s_inv;
}

/* scanline_generalized_gaussian_sampled_contrib, one channel, beam_antialias_level = 1 */
static float beam_contrib(float dist, float color, float ph, float sigma_range, float shape_range) {
  const float beam_spot_power = 1.0f / 3.0f, beam_shape_power = 1.0f / 4.0f;
  float sigma = beam_min_sigma + sigma_range * o_pow(color, beam_spot_power);
  float alpha = sqrtf(2.0f) * sigma;
  float beta = beam_min_shape + shape_range * o_pow(color, beam_shape_power);
  float alpha_inv = 1.0f / alpha, beta_inv = 1.0f / beta;
  float scale = color * beta * 0.5f * alpha_inv / gamma_impl1(beta_inv, beta);
  float off = ph / 3.0f;
  float d2 = dist + off, d3 = fabsf(dist - off);
  float w1 = o_exp(-o_pow(fabsf(dist * alpha_inv), beta));
  float w2 = o_exp(-o_pow(fabsf(d2 * alpha_inv), beta));
  float w3 = o_exp(-o_pow(fabsf(d3 * alpha_inv), beta));
  return scale / 3.0f * (w1 + w2 + w3);
}

/* beam_contrib on arrays (calibration of the scanline kernel's table form and its tests): out[i] =
 * beam_contrib(dist[i], color[i], ph) with the pass's static sigma / shape ranges */
void o_royale_beam_array(const float* dist, const float* color, float ph, float* out, size_t n) {
  ENTER;
  const float sigma_range = maxps(beam_max_sigma, beam_min_sigma) - beam_min_sigma;
  const float shape_range = maxps(beam_max_shape, beam_min_shape) - beam_min_shape;
  for (size_t i = 0; i < n; ++i) out[i] = beam_contrib(dist[i], color[i], ph, sigma_range, shape_range);
  LEAVE;
}

void o_pass_royale_scan_v(const o_pass_args* a) {
  ENTER;
  const int W = a->out_w, H = a->out_h;
  /* texture_size = TextureSize, video_size = InputSize (glsl :38-39).  They differ in one case: the reference
   * hands pass index 3 TextureSize.y = the target's height when that is not the input's (ShaderEngine.cpp:
   * 2418-2421) - which is where this shader sits in crt/crt-royale-ntsc-*.glslp */
  const float vsy = (float)a->in->h;
  const float tsx = (float)a->in->w, tsy = (a->pass_index == 3 && H != a->in->h) ? (float)H : vsy;
  uvplanes tc = texcoord_planes(1.0f, W, H, a->out_fmt);
  const float y_step = 1.0f + (is_interlaced(vsy) ? 1.0f : 0.0f);
  const float uv_step_y = y_step / tsy;
  const float ph = (vsy / (float)H) / y_step;
  const float tix = 1.0f / tsx, tiy = 1.0f / tsy;
  const float sigma_range = maxps(beam_max_sigma, beam_min_sigma) - beam_min_sigma;
  const float shape_range = maxps(beam_max_shape, beam_min_shape) - beam_min_shape;
  const float conv_y[3] = {0.2f, 0.4f, 0.6f};
  const float autodim = 0.5f;
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      int lo = o_lower_tri(x, y, W, H);
      float u = o_varying_at(&tc.u, x, y, lo), v = o_varying_at(&tc.v, x, y, lo);
      float dist;
      v2 suv = last_scanline_uv(u, v, tsx, tsy, tix, tiy, y_step, (float)a->frame_count, &dist);
      o_vec4 s2 = o_sample(a->in, suv.x, suv.y);
      o_vec4 s3 = o_sample(a->in, suv.x + 0.0f, suv.y + uv_step_y);
      /* beam_num_scanlines = 3: one more scanline, 1 or 4 */
      float dist_round = rintf(dist);
      float off_x = mix_rt(-0.0f, 2.0f * 0.0f, dist_round);
      float off_y = mix_rt(-uv_step_y, 2.0f * uv_step_y, dist_round);
      o_vec4 so = o_sample(a->in, suv.x + off_x, suv.y + off_y);
      float c2[3] = {s2.x, s2.y, s2.z}, c3[3] = {s3.x, s3.y, s3.z}, co[3] = {so.x, so.y, so.z};
      float out[3];
      for (int ch = 0; ch < 3; ++ch) {
        float d2 = dist - conv_y[ch];
        float k2 = beam_contrib(d2, c2[ch], ph, sigma_range, shape_range);
        /* The GL's compiler re-associates the additive constants of these expressions (measured on
         * a float target): 1.0 - (dist - c) is evaluated as (1.0 + c) - dist, etc. */
        float k3 = beam_contrib(fabsf((1.0f + conv_y[ch]) - dist), c3[ch], ph, sigma_range, shape_range);
        float inten = k2 + k3;
        float d14 = mix_rt(dist + (1.0f - conv_y[ch]), (2.0f + conv_y[ch]) - dist, dist_round);
        inten += beam_contrib(d14, co[ch], ph, sigma_range, shape_range);
        out[ch] = inten * autodim;
      }
      o_vec4 o = {out[0], out[1], out[2], 1.0f};
      o_store_pixel(a, x, y, o);
    }
  LEAVE;
}

/* =========================================================================== P2 ====== */
/* bloom-approx.glsl FS 14053-14184: every path above the last statement is dead; the output is
 * one tap of ORIG_LINEARIZED (= PassPrev2Texture, extra[0]) at tex_uv (VS 5926-5932). */
void o_pass_royale_bloom_approx(const o_pass_args* a) {
  ENTER;
  const int W = a->out_w, H = a->out_h;
  /* VS: video_uv = TexCoord * texture_size / video_size; tex_uv = video_uv * PassPrev2InputSize
   * / PassPrev2TextureSize.  pass_index - 2 is pass 0: InputSize = source, TextureSize = its
   * output; per vertex this is (0 or 1) * T / T * S / S. */
  const float tsx = (float)a->in->w, tsy = (float)a->in->h;
  const int p = a->pass_index - 2;
  const float pisx = p == 0 ? (float)a->src_w : (float)a->chain_w[p - 1], pisy = p == 0 ? (float)a->src_h : (float)a->chain_h[p - 1];
  const float ptsx = (float)a->chain_w[p], ptsy = (float)a->chain_h[p];
  const float u1 = ((1.0f * tsx) / tsx) * pisx / ptsx, v1 = ((1.0f * tsy) / tsy) * pisy / ptsy;
  const float u0 = ((0.0f * tsx) / tsx) * pisx / ptsx, v0 = ((0.0f * tsy) / tsy) * pisy / ptsy;
  o_varying pu = plane_u(u0, u1, W, H, a->out_fmt), pv = plane_v(v0, v1, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      int lo = o_lower_tri(x, y, W, H);
      o_store_pixel(a, x, y, o_sample(a->extra[0], o_varying_at(&pu, x, y, lo), o_varying_at(&pv, x, y, lo)));
    }
  LEAVE;
}

/* ====================================================================== P3 / P4 ====== */
/* blurs/blur9fast-{vertical,horizontal}.glsl: VS lines 2040-2048, tex2Dblur9fast 1496-1524 with
 * the compile-time sigma blur9_std_dev = 1.7533203125 (line 394). */
static void blur9(const o_pass_args* a, int horizontal) {
  const int W = a->out_w, H = a->out_h;
  const float sigma = 1.7533203125f;
  const float denom_inv = 0.5f / (sigma * sigma);
  const float w0 = 1.0f, w1 = const_exp(-1.0f * denom_inv), w2 = const_exp(-4.0f * denom_inv);
  const float w3 = const_exp(-9.0f * denom_inv), w4 = const_exp(-16.0f * denom_inv);
  /* the weights fold at compile time; the four-term sum is rebalanced by the GLSL compiler (measured:
   * weight_sum_inv read back from the GL is 1/(1 + 2*((w1+w2)+(w3+w4)))) */
  const float weight_sum_inv = 1.0f / (w0 + 2.0f * ((w1 + w2) + (w3 + w4)));
  const float w12 = w1 + w2, w34 = w3 + w4;
  const float w12_ratio = w2 / w12, w34_ratio = w4 / w34;
  const float k34 = 3.0f + w34_ratio, k12 = 1.0f + w12_ratio;
  /* VS: dxdy = (InputSize / OutputSize) / TextureSize, one axis zeroed */
  const float tsx = (float)a->in->w, tsy = (float)a->in->h;
  const float dx = horizontal ? (tsx / (float)W) / tsx : 0.0f;
  const float dy = horizontal ? 0.0f : (tsy / (float)H) / tsy;
  uvplanes tc = texcoord_planes(1.0f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      int lo = o_lower_tri(x, y, W, H);
      float u = o_varying_at(&tc.u, x, y, lo), v = o_varying_at(&tc.v, x, y, lo);
      o_vec4 s0 = o_sample(a->in, u - k34 * dx, v - k34 * dy);
      o_vec4 s1 = o_sample(a->in, u - k12 * dx, v - k12 * dy);
      o_vec4 s2 = o_sample(a->in, u, v);
      o_vec4 s3 = o_sample(a->in, u + k12 * dx, v + k12 * dy);
      o_vec4 s4 = o_sample(a->in, u + k34 * dx, v + k34 * dy);
      /* sum = w34*s0 + w12*s1 + 1.0*s2 + w12*s3 + w34*s4 as the GL evaluates it: every "+ product" is
       * fused to a multiply-add, a plain addend met by a pending multiply-add is added to ITS addend
       * first (fadd(x, ffma(a,b,c)) -> ffma(a,b, x + c)), then the fused operations are split again:
       * ((((A + s2) + B) + D) + E) - the only association of all 5-leaf trees that matches the float
       * goldens */
      v3 sum;
      sum.x = (((w34 * s0.x + s2.x) + w12 * s1.x) + w12 * s3.x) + w34 * s4.x;
      sum.y = (((w34 * s0.y + s2.y) + w12 * s1.y) + w12 * s3.y) + w34 * s4.y;
      sum.z = (((w34 * s0.z + s2.z) + w12 * s1.z) + w12 * s3.z) + w34 * s4.z;
      o_vec4 o = {sum.x * weight_sum_inv, sum.y * weight_sum_inv, sum.z * weight_sum_inv, 1.0f};
      o_store_pixel(a, x, y, o);
    }
}
void o_pass_blur9_v(const o_pass_args* a) { ENTER; blur9(a, 0); LEAVE; }
void o_pass_blur9_h(const o_pass_args* a) { ENTER; blur9(a, 1); LEAVE; }

/* ====================================================================== P5 / P6 ====== */
/* mask-resize-vertical.glsl (VS 3280-3305, FS 3372-3442) and mask-resize-horizontal.glsl
 * (VS 3276-3300, FS 3374-3404); Lanczos-windowed sinc resampling, phosphor-mask-resizing
 * functions 2624-2994.  Derived constants (file lines 1060-1160, 2603-2622) under the defines
 * that survive the driver-capability #ifdefs: PHOSPHOR_MASK_MANUALLY_RESIZE,
 * PHOSPHOR_MASK_RESIZE_LANCZOS_WINDOW, ANISOTROPIC_TILING_COMPAT_TILE_FLAT_TWICE,
 * USE_SINGLE_STATIC_LOOP (24 samples). */
static const float mask_resize_num_tiles = 2.0f;
static const float mask_triads_per_tile = 8.0f;
static const float mask_lut = 64.0f; /* mask_resize_src_lut_size = mask_texture_small_size */
static const float geom_aspect_ratio_static = 1.313069909f;
#define FIX_ZERO0 0.0000152587890625f /* FIX_ZERO(0.0) = max(abs(0), 2^-16) */

/* get_resized_mask_tile_size (2995-3046) with mask_sample_mode 0, mask_specify_num_triads 0,
 * mask_triad_size_desired 3 and the caller's `false` flag; tile aspect is 1. */
static v2 resized_mask_tile_size(float out_x, float out_y) {
  const float desired_tile_size_x = mask_triads_per_tile * 3.0f;
  const float temp = minps(desired_tile_size_x, mask_lut);
  const float min_tile = 16.0f; /* ceil(mask_min_allowed_triad_size * mask_triads_per_tile) */
  const float max_x = out_x / mask_resize_num_tiles, max_y = out_y / mask_resize_num_tiles;
  const float cx = clampf(temp * 1.0f, min_tile * 1.0f, max_x), cy = clampf(temp * 1.0f, min_tile * 1.0f, max_y);
  const float x_from_y = cy * 1.0f, y_from_x = cy; /* lerp(cy, cx, 0.0) */
  v2 r = {floorf(minps(cx, x_from_y) + FIX_ZERO0), floorf(minps(cy, y_from_x) + FIX_ZERO0)};
  return r;
}

typedef struct { float tile_uv, dist; } first_texel;
static first_texel first_texel_tile_uv_and_dist(float tex_uv, float tex_size, float dr, float tiles_per_tex, float samples) {
  float curr = tex_uv * tex_size;
  float prev = floorf(curr - under_half) + 0.5f;
  float first = prev - (samples / 2.0f - 1.0f);
  float uv_wrap = first * dr;
  first_texel r;
  r.dist = curr - first;
  float tile_uv_wrap = uv_wrap * tiles_per_tex;
  float neg = tile_uv_wrap < 0.0f ? 1.0f : 0.0f;
  r.tile_uv = fractf(tile_uv_wrap) + neg;
  return r;
}

/* downsample_{vertical,horizontal}_sinc_tiled: 24 taps in 6 groups of 4 */
static v3 sinc_tiled(const o_tex* t, float fixed_coord, float r_coord, float r_size, float dr, float magnification,
                     float tile_size_uv_r, int vertical) {
  const float pi = 3.141592653589f, pi_over_lobes = pi / 3.0f;
  const float samples = 24.0f;
  const float tiles_per_tex = 1.0f / tile_size_uv_r;
  first_texel ft = first_texel_tile_uv_and_dist(r_coord, r_size, dr, tiles_per_tex, samples);
  const float tile_dr = dr * tiles_per_tex;
  float wsum[4] = {0.f, 0.f, 0.f, 0.f};
  v3 color = {0.f, 0.f, 0.f};
  for (int i = 0; i < 24; i += 4) {
    float w[4];
    o_vec4 s[4];
    for (int k = 0; k < 4; ++k) {
      float true_i = (float)(0 + i) + (float)k;
      float tile_uv_r = fractf(ft.tile_uv + true_i * tile_dr);
      float tex_uv_r = tile_uv_r * tile_size_uv_r;
      s[k] = vertical ? o_sample(t, fixed_coord, tex_uv_r) : o_sample(t, tex_uv_r, fixed_coord);
      float dist = magnification * fabsf(ft.dist - true_i);
      float pi_dist = pi * dist;
      float pi_dist_over_lobes = pi_over_lobes * dist;
      w[k] = minps(o_sin(pi_dist) * o_sin(pi_dist_over_lobes) / (pi_dist * pi_dist_over_lobes), 1.0f);
    }
    for (int k = 0; k < 4; ++k) {
      color.x += s[k].x * w[k]; color.y += s[k].y * w[k]; color.z += s[k].z * w[k];
      wsum[k] += w[k];
    }
  }
  float rx = wsum[0] + wsum[2], ry = wsum[1] + wsum[3];
  float total = rx + ry;
  v3 r = {color.x / total, color.y / total, color.z / total};
  return r;
}

void o_pass_royale_mask_v(const o_pass_args* a) {
  ENTER;
  const int W = a->out_w, H = a->out_h;
  const float ox = (float)W, oy = (float)H;
  const float tsx = (float)a->in->w, tsy = (float)a->in->h;
  /* VS */
  const float viewport_y = oy / 0.0625f;
  const float aspect_ratio = geom_aspect_ratio_static / 1.0f;
  (void)viewport_y;
  v2 tile = resized_mask_tile_size(oy * aspect_ratio, oy); /* estimated_mask_resize_output_size */
  const float pots_x = minps(mask_lut, ox), pots_y = tile.y; /* pass_output_tile_size */
  const float tiles_x = ox / pots_x, tiles_y = oy / pots_y;
  /* src_tex_uv_wrap = (TexCoord * texture_size / video_size) * output_tiles_this_pass */
  o_varying pu = plane_u(((0.0f * tsx) / tsx) * tiles_x, ((1.0f * tsx) / tsx) * tiles_x, W, H, a->out_fmt);
  o_varying pv = plane_v(((0.0f * tsy) / tsy) * tiles_y, ((1.0f * tsy) / tsy) * tiles_y, W, H, a->out_fmt);
  const float mag_y = pots_y / mask_lut;
  const float src_dy = 1.0f / mask_lut;
  const o_tex* lut = a->extra[0]; /* mask_type 1 -> mask_slot_texture_small */
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      int lo = o_lower_tri(x, y, W, H);
      float wu = o_varying_at(&pu, x, y, lo), wv = o_varying_at(&pv, x, y, lo);
      o_vec4 o = {0.f, 0.f, 0.f, 0.f}; /* discard keeps the cleared target */
      if (wv <= mask_resize_num_tiles) {
        float su = fractf(wu), sv = fractf(wv);
        v3 c = sinc_tiled(lut, su, sv, mask_lut, src_dy, mag_y, 1.0f, 1);
        o.x = c.x; o.y = c.y; o.z = c.z; o.w = 1.0f;
        o_store_pixel(a, x, y, o);
      } else {
        size_t i = ((size_t)y * W + x) * 4;
        uint8_t* d = (uint8_t*)a->dst + i; /* RGBA8 target */
        d[0] = d[1] = d[2] = d[3] = 0;
      }
    }
  LEAVE;
}

void o_pass_royale_mask_h(const o_pass_args* a) {
  ENTER;
  const int W = a->out_w, H = a->out_h;
  const float ox = (float)W, oy = (float)H;
  const float tsx = (float)a->in->w, tsy = (float)a->in->h;
  /* VS */
  v2 tile = resized_mask_tile_size(ox, oy); /* (estimated_viewport_size is unused for mode 0) */
  const float tiles_x = ox / tile.x, tiles_y = oy / tile.y;
  const float its_x = minps(mask_lut, tsx), its_y = tile.y; /* input_tile_size */
  const float tsuv_x = its_x / tsx, tsuv_y = its_y / tsy;   /* tile_size_uv */
  /* src_tex_uv_wrap = ((TexCoord * T / V) * output_tiles_this_pass) * tile_size_uv */
  o_varying pu = plane_u((((0.0f * tsx) / tsx) * tiles_x) * tsuv_x, (((1.0f * tsx) / tsx) * tiles_x) * tsuv_x, W, H, a->out_fmt);
  o_varying pv = plane_v((((0.0f * tsy) / tsy) * tiles_y) * tsuv_y, (((1.0f * tsy) / tsy) * tiles_y) * tsuv_y, W, H, a->out_fmt);
  const float mag_x = tile.x / its_x;
  const float src_dx = 1.0f / tsx;
  const int render = (a->flags & O_FLAG_ROYALE_UNDEF_VARYING_ZERO) != 0;
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      size_t i = ((size_t)y * W + x) * 4;
      uint8_t* d = (uint8_t*)a->dst + i;
      if (!render) { /* the FS's test reads an undefined varying: every fragment is discarded */
        d[0] = d[1] = d[2] = d[3] = 0;
        continue;
      }
      /* undefined varying reads 0: max(0, 0) <= mask_resize_num_tiles holds everywhere */
      int lo = o_lower_tri(x, y, W, H);
      float wu = o_varying_at(&pu, x, y, lo), wv = o_varying_at(&pv, x, y, lo);
      float su = fractf(wu), sv = fractf(wv);
      v3 c = sinc_tiled(a->in, sv, su, tsx, src_dx, mag_x, tsuv_x, 0);
      o_vec4 o = {c.x, c.y, c.z, 1.0f};
      o_store_pixel(a, x, y, o);
    }
  LEAVE;
}

/* =========================================================================== P7 ====== */
/* scanlines-horizontal-apply-mask.glsl: VS 6103-6135, FS 10877-11030;
 * sample_single_scanline_horizontal 5198-5241 (beam_horiz_filter 0: Quilez weights),
 * get_interpolated_linear_color 5080-5163 (linear RGB weight 1 -> linear_mixed_color),
 * get_mask_sampling_parameters 5875-5908, convert_phosphor_tile_uv_wrap_to_tex_uv 5965-5992.
 * extra[0] = PassPrev6Texture (VERTICAL_SCANLINES), extra[1] = PassPrev3Texture (HALATION_BLUR). */
static void prev_pass_sizes(const o_pass_args* a, int n, float* in_w, float* in_h, float* tex_w, float* tex_h) {
  /* PassPrev<n>InputSize / TextureSize as the reference engine sets them
   * (ShaderEngine.cpp:1191-1227): target pass p = pass_index - n; TextureSize = p's output
   * size; InputSize = what p received = output of p-1, or the source frame for p = 0. */
  int p = a->pass_index - n;
  *tex_w = (float)a->chain_w[p];
  *tex_h = (float)a->chain_h[p];
  *in_w = p == 0 ? (float)a->src_w : (float)a->chain_w[p - 1];
  *in_h = p == 0 ? (float)a->src_h : (float)a->chain_h[p - 1];
}

static float scanline_horizontal_1ch(const o_tex* t, float u, float v, float tsx, float tsy, float tix, float tiy, int ch) {
  float ctx = u * tsx, cty = v * tsy;
  float ptx = floorf(ctx - under_half) + 0.5f;
  float phx = ptx, phy = cty;
  float puv_x = phx * tix, puv_y = phy * tiy;
  float prev_dist = ctx - phx;
  float x = prev_dist;
  float w2 = x * x * x * (x * (x * 6.0f - 15.0f) + 10.0f);
  float wx = 0.0f, wy = 1.0f - w2, wz = w2, ww = 0.0f;
  float dot = ((wx * 1.0f + wy * 1.0f) + wz * 1.0f) + ww * 1.0f;
  float fx = wx / dot, fy = wy / dot, fz = wz / dot, fw = ww / dot;
  o_vec4 c1 = o_sample(t, puv_x, puv_y);
  o_vec4 c2 = o_sample(t, puv_x + tix, puv_y + 0.0f);
  float a1 = ch == 0 ? c1.x : (ch == 1 ? c1.y : c1.z);
  float a2 = ch == 0 ? c2.x : (ch == 1 ? c2.y : c2.z);
  float m = ((0.0f * fx + a1 * fy) + a2 * fz) + 0.0f * fw;
  return maxps(m, 0.0f);
}

static const float mask_amplify = 1.0f / (46.0f / 255.0f); /* mask_type 1: 1/mask_slot_avg_color */
/* FS 10948-11027 with PHOSPHOR_BLOOM_FAKE (and PHOSPHOR_BLOOM_FAKE_WITH_SIMPLE_BLEND): every parameter is a
 * compile-time constant in this file (no #pragma parameter).  scan = electron_intensity_dim (halation_weight 0). */
static float fake_bloom_tail(float scan, float mask, float soft, float hal) {
  const float undim = 1.0f / 0.5f, under = 0.8f, diffusion = 0.075f, contrast = 1.05f;
  const float ped = scan * mask;
  const float pe = ped * (undim * mask_amplify);
  const float ei = scan * undim;
  const float lerped = soft * (1.0f - 0.1f) + ei * 0.1f;
  const float approx = lerped * contrast;
  /* the GL's compiler gathers the constant factors of each product chain (float goldens): the
   * underestimates are products of the lerp itself, not of `approx` */
  const float pbu = lerped * (contrast * under);
  const float amu = lerped * ((contrast * under) * mask_amplify);
  const float rt = (amu - 1.0f) / (amu - pbu);
  const float ratio = maxps(clampf(rt, 0.0f, 1.0f), 0.0f);
  const float unclipped = pe + ratio * (approx - pe);   /* non-constant weight: x + t*(y - x) */
  return unclipped * (1.0f - diffusion) + hal * diffusion;
}
static void royale_scan_h_body(const o_pass_args* a, int fake) {
  const int W = a->out_w, H = a->out_h;
  const float ox = (float)W, oy = (float)H;
  const float tsx = (float)a->in->w, tsy = (float)a->in->h; /* MASK_RESIZE texture/video size */
  float p6iw, p6ih, p6tw, p6th;
  prev_pass_sizes(a, 6, &p6iw, &p6ih, &p6tw, &p6th);
  const float stix = 1.0f / p6tw, stiy = 1.0f / p6th; /* scanline_texture_size_inv */
  /* vertex values */
  const float vu1 = (1.0f * tsx) / tsx, vv1 = (1.0f * tsy) / tsy, vu0 = (0.0f * tsx) / tsx, vv0 = (0.0f * tsy) / tsy;
  o_varying p_vu = plane_u(vu0, vu1, W, H, a->out_fmt), p_vv = plane_v(vv0, vv1, W, H, a->out_fmt);
  o_varying p_su = plane_u(vu0 * p6iw * stix, vu1 * p6iw * stix, W, H, a->out_fmt);
  o_varying p_sv = plane_v(vv0 * p6ih * stiy, vv1 * p6ih * stiy, W, H, a->out_fmt);
  /* get_mask_sampling_parameters(MASK_RESIZE texture size, video size, output_size) */
  v2 tile = resized_mask_tile_size(tsx, tsy);
  const float uvs_x = tile.x / tsx, uvs_y = tile.y / tsy;                 /* mask_tile_uv_size */
  const float start_x = (0.0f / tile.x) * uvs_x, start_y = (0.0f / tile.y) * uvs_y; /* mask_start_texels = 0 */
  const float tps_x = ox / tile.x, tps_y = oy / tile.y;                    /* mask_tiles_per_screen */
  const float conv_x[3] = {0.1f, 0.3f, 0.5f};
  const o_tex* scan = a->extra[0];
  /* PHOSPHOR_BLOOM_FAKE: blur3x3_tex_uv / halation_tex_uv = video_uv * <pass>video_size / <pass>texture_size
   * (VS 6117-6120) for BLOOM_APPROX (PassPrev5) and HALATION_BLUR (PassPrev3) */
  o_varying p_bu = p_vu, p_bv = p_vv, p_hu = p_vu, p_hv = p_vv;
  if (fake) {
    float iw, ih, tw, th;
    prev_pass_sizes(a, 5, &iw, &ih, &tw, &th);
    p_bu = plane_u(vu0 * iw / tw, vu1 * iw / tw, W, H, a->out_fmt);
    p_bv = plane_v(vv0 * ih / th, vv1 * ih / th, W, H, a->out_fmt);
    prev_pass_sizes(a, 3, &iw, &ih, &tw, &th);
    p_hu = plane_u(vu0 * iw / tw, vu1 * iw / tw, W, H, a->out_fmt);
    p_hv = plane_v(vv0 * ih / th, vv1 * ih / th, W, H, a->out_fmt);
  }
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      int lo = o_lower_tri(x, y, W, H);
      float vu = o_varying_at(&p_vu, x, y, lo), vv = o_varying_at(&p_vv, x, y, lo);
      float su = o_varying_at(&p_su, x, y, lo), sv = o_varying_at(&p_sv, x, y, lo);
      float scanc[3];
      for (int ch = 0; ch < 3; ++ch) {
        float off = conv_x[ch] * stix;
        scanc[ch] = scanline_horizontal_1ch(scan, su - off, sv - 0.0f, p6tw, p6th, stix, stiy, ch);
      }
      float twx = vu * tps_x, twy = vv * tps_y;
      float tux = fractf(twx * 0.5f) * 2.0f, tuy = fractf(twy * 0.5f) * 2.0f;
      float mu = start_x + tux * uvs_x, mv = start_y + tuy * uvs_y;
      o_vec4 mask = o_sample(a->in, mu, mv);
      /* electron_intensity_dim = lerp(scanline, halation_intensity, halation_weight = 0) */
      o_vec4 o = {scanc[0] * mask.x, scanc[1] * mask.y, scanc[2] * mask.z, 1.0f};
      if (fake) {
        const o_vec4 soft = o_sample(a->extra[1], o_varying_at(&p_bu, x, y, lo), o_varying_at(&p_bv, x, y, lo));
        const o_vec4 hal = o_sample(a->extra[2], o_varying_at(&p_hu, x, y, lo), o_varying_at(&p_hv, x, y, lo));
        const float m3[3] = {mask.x, mask.y, mask.z}, s3[3] = {soft.x, soft.y, soft.z}, h3[3] = {hal.x, hal.y, hal.z};
        float out[3];
        for (int c = 0; c < 3; ++c) out[c] = fake_bloom_tail(scanc[c], m3[c], s3[c], h3[c]);
        o.x = out[0]; o.y = out[1]; o.z = out[2];
      }
      o_store_pixel(a, x, y, o);
    }
}
void o_pass_royale_scan_h(const o_pass_args* a) {
  ENTER;
  royale_scan_h_body(a, 0);
  LEAVE;
}
/* scanlines-horizontal-apply-mask-fake-bloom.glsl: the same file with PHOSPHOR_BLOOM_FAKE defined;
 * extra = PassPrev6 (VERTICAL_SCANLINES), PassPrev5 (BLOOM_APPROX), PassPrev3 (HALATION_BLUR) */
void o_pass_royale_scan_h_fake(const o_pass_args* a) {
  ENTER;
  royale_scan_h_body(a, 1);
  LEAVE;
}

/* ===================================================================== P8 - P10 ====== */
/* bloom sigma: VS of brightpass / bloom-vertical / bloom-horizontal-reconstitute
 * (e.g. brightpass.glsl 6616-6649): get_min_sigma_to_blur_triad(tile_x / 8, 1/256). */
static float bloom_sigma_runtime(float ox, float oy) {
  v2 tile = resized_mask_tile_size(ox * 0.0625f, oy * 0.0625f);
  float triad = tile.x / mask_triads_per_tile;
  const float thresh = 1.0f / 256.0f;
  return -0.05168f + 0.6113f * triad - 1.122f * triad * sqrtf(0.000416f + thresh);
}
/* get_fast_gaussian_weight_sum_inv (bloom-vertical.glsl 6624-6628), evaluated per fragment */
static float center_weight(float sigma) {
  return minps(o_exp(o_exp(0.348348412457428f / (sigma - 0.0860587260734721f))), 0.399334576340352f / sigma);
}

/* brightpass.glsl FS 14610-14663; extra[0] = PassPrev4Texture */
void o_pass_royale_brightpass(const o_pass_args* a) {
  ENTER;
  const int W = a->out_w, H = a->out_h;
  const float tsx = (float)a->in->w, tsy = (float)a->in->h;
  float qiw, qih, qtw, qth;
  prev_pass_sizes(a, 4, &qiw, &qih, &qtw, &qth);
  const float vu1 = (1.0f * tsx) / tsx, vv1 = (1.0f * tsy) / tsy, vu0 = (0.0f * tsx) / tsx, vv0 = (0.0f * tsy) / tsy;
  o_varying p_su = plane_u(vu0 * tsx / tsx, vu1 * tsx / tsx, W, H, a->out_fmt), p_sv = plane_v(vv0 * tsy / tsy, vv1 * tsy / tsy, W, H, a->out_fmt);
  o_varying p_bu = plane_u(vu0 * qiw / qtw, vu1 * qiw / qtw, W, H, a->out_fmt), p_bv = plane_v(vv0 * qih / qth, vv1 * qih / qth, W, H, a->out_fmt);
  const float sigma = bloom_sigma_runtime((float)W, (float)H);
  const float undim = 1.0f / 0.5f;
  const float under = 0.8f; /* bloom_underestimate_levels */
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      int lo = o_lower_tri(x, y, W, H);
      o_vec4 idim = o_sample(a->in, o_varying_at(&p_su, x, y, lo), o_varying_at(&p_sv, x, y, lo));
      o_vec4 blur = o_sample(a->extra[0], o_varying_at(&p_bu, x, y, lo), o_varying_at(&p_bv, x, y, lo));
      const float cw = center_weight(sigma);
      float in3[3] = {idim.x, idim.y, idim.z}, bl3[3] = {blur.x, blur.y, blur.z}, out[3];
      for (int c = 0; c < 3; ++c) {
        float intensity = in3[c] * undim * mask_amplify * 1.0f;
        float pba = 1.0f * bl3[c];
        float max_area = maxps(pba - cw * intensity, 0.0f);
        float area_under = under * max_area;
        /* bloom_underestimate_levels * (intensity_dim * undim * mask_amplify * levels_contrast): every
         * factor but the sample is a compile-time constant in this file (no #pragma parameter, so
         * PARAMETER_UNIFORM is undefined, brightpass.glsl ~2355), and the GL's compiler gathers constant
         * factors of a product chain into one: in * (undim*mask_amplify*under) (float goldens) */
        float int_under = in3[c] * ((undim * mask_amplify) * under);
        float ratio_temp = ((1.0f - area_under) / int_under - 1.0f) / (cw - 1.0f);
        float ratio = clampf(ratio_temp, 0.0f, 1.0f);
        out[c] = in3[c] * ratio; /* lerp(blur_ratio, 1, bloom_excess = 0) = blur_ratio */
      }
      o_vec4 o = {out[0], out[1], out[2], 1.0f};
      o_store_pixel(a, x, y, o);
    }
  LEAVE;
}

/* tex2Dblur17fast (bloom-vertical.glsl 7132-7176) with a run-time sigma */
typedef struct { float w0, w12, w34, w56, w78, k12, k34, k56, k78, sum_inv; } blur17_w;
static blur17_w blur17_weights(float sigma) {
  blur17_w b;
  float denom_inv = 0.5f / (sigma * sigma);
  float w1 = o_exp(-1.0f * denom_inv), w2 = o_exp(-4.0f * denom_inv), w3 = o_exp(-9.0f * denom_inv);
  float w4 = o_exp(-16.0f * denom_inv), w5 = o_exp(-25.0f * denom_inv), w6 = o_exp(-36.0f * denom_inv);
  float w7 = o_exp(-49.0f * denom_inv), w8 = o_exp(-64.0f * denom_inv);
  b.w0 = 1.0f;
  b.sum_inv = center_weight(sigma);
  b.w12 = w1 + w2; b.w34 = w3 + w4; b.w56 = w5 + w6; b.w78 = w7 + w8;
  b.k12 = 1.0f + w2 / b.w12; b.k34 = 3.0f + w4 / b.w34; b.k56 = 5.0f + w6 / b.w56; b.k78 = 7.0f + w8 / b.w78;
  return b;
}
static v3 blur17(const o_tex* t, float u, float v, float dx, float dy, const blur17_w* b) {
  const float ks[9] = {-b->k78, -b->k56, -b->k34, -b->k12, 0.0f, b->k12, b->k34, b->k56, b->k78};
  const float ws[9] = {b->w78, b->w56, b->w34, b->w12, b->w0, b->w12, b->w34, b->w56, b->w78};
  /* the nine terms are added in source order, except that the centre term - weight 1.0, so a plain
   * addend, not a product - is added before the product that precedes it (same mechanism as blur9:
   * fadd(x, ffma(a,b,c)) -> ffma(a,b, x + c)): (((A+B)+C) + centre) + D, then + E + F + G + H */
  o_vec4 smp[9];
  for (int i = 0; i < 9; ++i) {
    if (i < 4) smp[i] = o_sample(t, u - (-ks[i]) * dx, v - (-ks[i]) * dy);
    else if (i == 4) smp[i] = o_sample(t, u, v);
    else smp[i] = o_sample(t, u + ks[i] * dx, v + ks[i] * dy);
  }
  static const int order[9] = {0, 1, 2, 4, 3, 5, 6, 7, 8};
  v3 sum = {0.f, 0.f, 0.f};
  for (int j = 0; j < 9; ++j) {
    const int i = order[j];
    sum.x += ws[i] * smp[i].x; sum.y += ws[i] * smp[i].y; sum.z += ws[i] * smp[i].z;
  }
  v3 r = {sum.x * b->sum_inv, sum.y * b->sum_inv, sum.z * b->sum_inv};
  return r;
}

/* bloom-vertical.glsl: VS 3851-3861, FS 8605-8613 */
void o_pass_royale_bloom_v(const o_pass_args* a) {
  ENTER;
  const int W = a->out_w, H = a->out_h;
  const float tsx = (float)a->in->w, tsy = (float)a->in->h;
  uvplanes tc = texcoord_planes(1.0001f, W, H, a->out_fmt);
  const float dy = (tsy / (float)H) / tsy;
  (void)tsx;
  const float sigma = bloom_sigma_runtime((float)W, (float)H);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      int lo = o_lower_tri(x, y, W, H);
      blur17_w b = blur17_weights(sigma);
      v3 c = blur17(a->in, o_varying_at(&tc.u, x, y, lo), o_varying_at(&tc.v, x, y, lo), 0.0f, dy, &b);
      o_vec4 o = {c.x, c.y, c.z, 1.0f};
      o_store_pixel(a, x, y, o);
    }
  LEAVE;
}

/* bloom-horizontal-reconstitute.glsl: VS 6641-6660, FS 11407-11439.
 * extra[0] = PassPrev3Texture (MASKED_SCANLINES), extra[1] = PassPrev2Texture (BRIGHTPASS),
 * extra[2] = PassPrev6Texture (HALATION_BLUR). */
void o_pass_royale_bloom_h(const o_pass_args* a) {
  ENTER;
  const int W = a->out_w, H = a->out_h;
  const float tsx = (float)a->in->w, tsy = (float)a->in->h;
  float m_iw, m_ih, m_tw, m_th, b_iw, b_ih, b_tw, b_th, h_iw, h_ih, h_tw, h_th;
  prev_pass_sizes(a, 3, &m_iw, &m_ih, &m_tw, &m_th);
  prev_pass_sizes(a, 2, &b_iw, &b_ih, &b_tw, &b_th);
  prev_pass_sizes(a, 6, &h_iw, &h_ih, &h_tw, &h_th);
  const float vu1 = (1.0f * tsx) / tsx, vv1 = (1.0f * tsy) / tsy, vu0 = (0.0f * tsx) / tsx, vv0 = (0.0f * tsy) / tsy;
  o_varying p_mu = plane_u(vu0 * m_iw / m_tw, vu1 * m_iw / m_tw, W, H, a->out_fmt), p_mv = plane_v(vv0 * m_ih / m_th, vv1 * m_ih / m_th, W, H, a->out_fmt);
  o_varying p_hu = plane_u(vu0 * h_iw / h_tw, vu1 * h_iw / h_tw, W, H, a->out_fmt), p_hv = plane_v(vv0 * h_ih / h_th, vv1 * h_ih / h_th, W, H, a->out_fmt);
  o_varying p_bu = plane_u(vu0 * b_iw / b_tw, vu1 * b_iw / b_tw, W, H, a->out_fmt), p_bv = plane_v(vv0 * b_ih / b_th, vv1 * b_ih / b_th, W, H, a->out_fmt);
  uvplanes tc = texcoord_planes(1.0f, W, H, a->out_fmt);
  const float dx = 1.0f / tsx;
  const float sigma = bloom_sigma_runtime((float)W, (float)H);
  const float undim = 1.0f / 0.5f;
  const float diffusion = 0.075f;
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      int lo = o_lower_tri(x, y, W, H);
      blur17_w b = blur17_weights(sigma);
      v3 blurred = blur17(a->in, o_varying_at(&tc.u, x, y, lo), o_varying_at(&tc.v, x, y, lo), dx, 0.0f, &b);
      o_vec4 idim = o_sample(a->extra[0], o_varying_at(&p_mu, x, y, lo), o_varying_at(&p_mv, x, y, lo));
      o_vec4 bright = o_sample(a->extra[1], o_varying_at(&p_bu, x, y, lo), o_varying_at(&p_bv, x, y, lo));
      o_vec4 hal = o_sample(a->extra[2], o_varying_at(&p_hu, x, y, lo), o_varying_at(&p_hv, x, y, lo));
      float i3[3] = {idim.x, idim.y, idim.z}, b3[3] = {bright.x, bright.y, bright.z}, bl[3] = {blurred.x, blurred.y, blurred.z};
      float h3[3] = {hal.x, hal.y, hal.z}, out[3];
      for (int c = 0; c < 3; ++c) {
        float dimpass = i3[c] - b3[c];
        /* lerp((dimpass + blurred) * mask_amplify * undim * contrast, contrast * halation, diffusion_weight):
         * all parameters are compile-time constants here, the lerp is a*(1-t) + b*t and the constant
         * factors of a*(1-t) are gathered into one: X * (mask_amplify*undim*(1-t)) + h * t (float goldens) */
        out[c] = (dimpass + bl[c]) * ((mask_amplify * undim) * (1.0f - diffusion)) + h3[c] * diffusion;
      }
      o_vec4 o = {out[0], out[1], out[2], 1.0f};
      o_store_pixel(a, x, y, o);
    }
  LEAVE;
}

/* ========================================================================== P11 ====== */
/* geometry-aa-last-pass.glsl: VS 5337-5400, FS 5451-5531, get_border_dim_factor 5250-5261,
 * get_aspect_vector 2512-2520.  LAST_PASS + SIMULATE_CRT_ON_LCD: output gamma = lcd_gamma.
 * params: the file's 44 #pragma parameters in declaration order (lines 21-64).
 * The flat path (geom_mode_runtime <= 0.5 and overscan == 1) is restated here; the tex2Daa / curved-geometry path is
 * rc_passes_royale_last.c. */
enum { RP_LCD_GAMMA = 1, RP_GEOM_MODE = 30, RP_OVERSCAN_X = 37, RP_OVERSCAN_Y = 38, RP_BORDER_SIZE = 39,
       RP_BORDER_DARKNESS = 40, RP_BORDER_COMPRESS = 41 };

void o_pass_royale_last(const o_pass_args* a) {
  if (o_royale_last_is_general(a->params)) {
    o_pass_royale_last_general(a);
    return;
  }
  ENTER;
  const int W = a->out_w, H = a->out_h;
  const float tsx = (float)a->in->w, tsy = (float)a->in->h; /* texture_size == video_size */
  const float* P = a->params;
  const float lcd_gamma = P[RP_LCD_GAMMA], osx = P[RP_OVERSCAN_X], osy = P[RP_OVERSCAN_Y];
  const float border_size = P[RP_BORDER_SIZE], border_darkness = P[RP_BORDER_DARKNESS], border_compress = P[RP_BORDER_COMPRESS];
  /* VS */
  const float vsix = 1.0f / tsx, vsiy = 1.0f / tsy, tsix = 1.0f / tsx, tsiy = 1.0f / tsy;
  const float ar = (float)W / (float)H;
  const float gx = minps(ar, 4.0f / 3.0f), gy = 1.0f;
  const float rs = 1.0f / sqrtf(gx * gx + gy * gy);
  const float geom_aspect_x = gx * rs, geom_aspect_y = gy * rs;
  /* This pass's vertex shader emits eye_pos_local, which is NaN in the flat geometry mode;
   * llvmpipe only takes its single-plane rectangle path when every varying is consistent
   * across the quad, so this pass is rasterised as two triangles even on an RGBA8 target
   * (measured: per-pixel intermediates differ across the BL-TR diagonal). */
  uvplanes tc = texcoord_planes(1.0f, W, H, O_FMT_SRGB8 /* = two-triangle planes */);
  const float inv_gamma = 1.0f / lcd_gamma;
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      int lo = o_lower_tri(x, y, W, H);
      float u = o_varying_at(&tc.u, x, y, lo), v = o_varying_at(&tc.v, x, y, lo);
      float fu = u * (tsx * vsix), fv = v * (tsy * vsiy);          /* flat_video_uv */
      float vu = (fu - 0.5f) / osx + 0.5f, vv = (fv - 0.5f) / osy + 0.5f; /* video_uv */
      float tu = vu * (tsx * tsix), tv = vv * (tsy * tsiy);
      o_vec4 c;
      if (a->in->n_levels > 1) {
        /* mipmap_input (crt-royale-fake-bloom's last pass): the sample coordinate of the quad's pixels as this
         * pixel's triangle extrapolates them (rc_sampler.c, o_sample_quad) */
        const int x0 = x & ~1, y0 = y & ~1;
        float qu[4], qv[4];
        const int qx[4] = {x0, x0 + 1, x, x}, qy[4] = {y, y, y0, y0 + 1};
        for (int k = 0; k < 4; ++k) {
          const float uu = o_varying_at(&tc.u, qx[k], qy[k], lo), vq = o_varying_at(&tc.v, qx[k], qy[k], lo);
          qu[k] = ((uu * (tsx * vsix) - 0.5f) / osx + 0.5f) * (tsx * tsix);
          qv[k] = ((vq * (tsy * vsiy) - 0.5f) / osy + 0.5f) * (tsy * tsiy);
        }
        c = o_sample_quad(a->in, tu, tv, qu[0], qu[1], qv[0], qv[1], qu[2], qu[3], qv[2], qv[3]);
      } else {
        c = o_sample(a->in, tu, tv);
      }
      /* get_border_dim_factor */
      float ex = minps(vu, 1.0f - vu) * geom_aspect_x, ey = minps(vv, 1.0f - vv) * geom_aspect_y;
      float bx = maxps(border_size - ex, 0.0f), by = maxps(border_size - ey, 0.0f);
      float pen = sqrtf(bx * bx + by * by) / border_size;
      float esc = maxps(1.0f - pen, 0.0f);
      float f = minps(o_pow(esc, border_darkness) * maxps(1.0f, border_compress), 1.0f);
      o_vec4 o = {o_pow(c.x * f, inv_gamma), o_pow(c.y * f, inv_gamma), o_pow(c.z * f, inv_gamma), 1.0f};
      o_store_pixel(a, x, y, o);
    }
  LEAVE;
}
