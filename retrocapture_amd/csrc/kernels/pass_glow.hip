// crt/crt-hyllian-glow.glslp - the reference's smoke-test default preset - 6 passes:
//   P0 crt/shaders/glow/linearize.glsl                              FS 88-93
//   P1 crt/shaders/hyllian/crt-hyllian-glow/crt-hyllian-glow.glsl   FS 146-247
//   P2 crt/shaders/glow/threshold.glsl                              FS 90-96
//   P3 crt/shaders/glow/blur_horiz.glsl (mip-mapped input)          FS 84-98
//   P4 crt/shaders/glow/blur_vert.glsl                              FS 84-98
//   P5 crt/shaders/hyllian/crt-hyllian-glow/resolve2.glsl           FS 129-189, 408-424
// Operation order as in oracle/rc_passes_glow.c (pinned by the float-precision goldens).  These
// passes use the run-time selected samplers: they are small next to crt-royale's and HBM-light
// (P3 / P4 work at 1/16 of the pixels).
#include "pass_launch.h"

using namespace rcd;

namespace {

__device__ __forceinline__ float minps(float a, float b) { return a < b ? a : b; }  // SSE minps: NaN -> b
__device__ __forceinline__ float maxps(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float mix_rt_(float a, float b, float t) { return a + t * (b - a); }
__device__ __forceinline__ float clamp_ps(float x, float lo, float hi) { return minps(maxps(x, lo), hi); }

// params: INPUT_GAMMA
__global__ void __launch_bounds__(256) k_glow_linearize(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float g = L.params[0];
  const float4 c = sample_rt(L.in, frame_ptr(L.in, z), vary(L.plane[0], x, y, lo), vary(L.plane[1], x, y, lo), &lds);
  store_rt(L, z, x, y, make_float4(pow_(c.x, g), pow_(c.y, g), pow_(c.z, g), 1.0f), &lds);
  RC_TILE_LOOP_END
}

// 8-bit NEAREST clamp-to-edge input and an 8-bit target (the shipped preset): both passes are per-channel
// functions of the source byte - a 256-entry byte map applied to the nearest texel (as crt-royale's first
// pass, pass_royale.hip).  OP 0: linearize, OP 1: threshold.  Same bytes as the general kernels (tested).
template <int OP>
__global__ void __launch_bounds__(256) k_glow_bytemap(const PassLaunch L) {
    __shared__ uint32_t map[256];
  RC_SRGB_LDS(lds, L);
  {
    const int t = threadIdx.y * 64 + threadIdx.x;
    const float c = L.in.fmt == FMT_SRGB8 ? lds.dec[t] : (float)t * (1.0f / 255.0f);
    const float r = OP == 0 ? pow_(c, L.params[0]) : pow_(clamp_ps((1.15f * c) / L.params[0], 0.0f, 1.0f), L.params[1]);
    map[t] = L.out_fmt == FMT_SRGB8 ? srgb8(r, &lds) : unorm8(r);
  }
  __syncthreads();
  RC_TILE_LOOP_BEGIN
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const int sx = clampi((int)__builtin_floorf(u * (float)L.in.w), 0, L.in.w - 1);
  const int sy = clampi((int)__builtin_floorf(v * (float)L.in.h), 0, L.in.h - 1);
  const uint32_t p = *reinterpret_cast<const uint32_t*>(frame_ptr(L.in, z) + texel_off(L.in.w, sx, sy, 4u));
  const uint32_t o = map[p & 255u] | (map[(p >> 8) & 255u] << 8) | (map[(p >> 16) & 255u] << 16) | 0xff000000u;
  *reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(L.out) + L.out_frame_stride * (uint64_t)z + texel_off(L.out_w, x, y, 4u)) = o;
  RC_TILE_LOOP_END
}

// params: GLOW_WHITEPOINT, GLOW_ROLLOFF
__global__ void __launch_bounds__(256) k_glow_threshold(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float wp = L.params[0], roll = L.params[1];
  const float4 c = sample_rt(L.in, frame_ptr(L.in, z), vary(L.plane[0], x, y, lo), vary(L.plane[1], x, y, lo), &lds);
  store_rt(L, z, x, y,
           make_float4(pow_(clamp_ps((1.15f * c.x) / wp, 0.0f, 1.0f), roll), pow_(clamp_ps((1.15f * c.y) / wp, 0.0f, 1.0f), roll),
                       pow_(clamp_ps((1.15f * c.z) / wp, 0.0f, 1.0f), roll), 1.0f),
           &lds);
  RC_TILE_LOOP_END
}

// 9 taps, weights exp(-0.35 i^2) folded by the GL's compiler with a correctly rounded exp (params[0..8]) and
// their sum (params[9]); HORIZ steps 4 texels on a mip-mapped input, else 1 texel vertically
template <bool HORIZ>
__global__ void __launch_bounds__(256) k_glow_blur(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float step = HORIZ ? 4.0f * (1.0f / (float)L.in.w) : 1.0f / (float)L.in.h;
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const bool mip = HORIZ && L.in.n_levels > 1;
  const int x0 = x & ~1, y0 = y & ~1;
  // the quad's coordinates as this pixel's triangle extrapolates them
  const float ux0 = vary(L.plane[0], x0, y, lo), ux1 = vary(L.plane[0], x0 + 1, y, lo);
  const float vx0 = vary(L.plane[1], x0, y, lo), vx1 = vary(L.plane[1], x0 + 1, y, lo);
  const float uy0 = vary(L.plane[0], x, y0, lo), uy1 = vary(L.plane[0], x, y0 + 1, lo);
  const float vy0 = vary(L.plane[1], x, y0, lo), vy1 = vary(L.plane[1], x, y0 + 1, lo);
  const uint8_t* img = frame_ptr(L.in, z);
  float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
  for (int i = -4; i <= 4; ++i) {
    const float off = (float)i * step;
    const float su = HORIZ ? u + off : u + 0.0f, sv = HORIZ ? v + 0.0f : v + off;
    float4 c;
    if (mip) {
      const float lod = lod_from_quad(L.in, ux0 + off, ux1 + off, vx0 + 0.0f, vx1 + 0.0f, uy0 + off, uy1 + off, vy0 + 0.0f, vy1 + 0.0f);
      c = sample_mip(L.in, z, su, sv, lod, &lds);
    } else {
      c = sample_rt(L.in, img, su, sv, &lds);
    }
    const float k = L.params[i + 4];
    c0 += k * c.x;
    c1 += k * c.y;
    c2 += k * c.z;
  }
  const float kt = L.params[9];
  store_rt(L, z, x, y, make_float4(c0 / kt, c1 / kt, c2 / kt, 1.0f), &lds);
  RC_TILE_LOOP_END
}

// params[0..9]: BEAM_PROFILE, BEAM_MIN_WIDTH, BEAM_MAX_WIDTH, SCANLINES_STRENGTH, COLOR_BOOST, HFILTER_SHARPNESS,
// CRT_ANTI_RINGING, InputGamma, OutputGamma, VSCANLINES; params[16..31]: invX (column c, row r at 16 + 4c + r),
// params[32..35]: the beam profile in effect (scanlines strength, min width, max width, colour boost) (host)
// TEXEL_LUT: 8-bit NEAREST clamp-to-edge input - GAMMA_IN(texel) is a function of the texel's bytes, so the 32
// pows per pixel become lookups in two 256-entry tables (colour channels, alpha) built once per workgroup.
template <bool TEXEL_LUT>
__global__ void __launch_bounds__(256) k_crt_hyllian_glow(const PassLaunch L) {
    __shared__ float gin_rgb[TEXEL_LUT ? 256 : 1], gin_a[TEXEL_LUT ? 256 : 1];
  RC_SRGB_LDS(lds, L);
  if (TEXEL_LUT) {
    const int t = threadIdx.y * 64 + threadIdx.x;
    const float k = (float)t * (1.0f / 255.0f);
    gin_rgb[t] = pow_(L.in.fmt == FMT_SRGB8 ? lds.dec[t] : k, L.params[7]);
    gin_a[t] = pow_(L.in.fmt == FMT_RGBX8 ? 1.0f : k, L.params[7]);
    __syncthreads();
  }
  RC_TILE_LOOP_BEGIN
  const float anti = L.params[6], gin = L.params[7], gout = L.params[8], vs = L.params[9];
  const float* m = &L.params[16];
  const float scan = 4.0f * L.params[32], bmin = L.params[33], bmax = L.params[34], boost = L.params[35];
  const float tsx = (float)L.in.w, tsy = (float)L.in.h;
  const float dxx = mix_rt_(1.0f / tsx, 0.0f, vs), dxy = mix_rt_(0.0f, 1.0f / tsy, vs);
  const float dyx = mix_rt_(0.0f, 1.0f / tsx, vs), dyy = mix_rt_(1.0f / tsy, 0.0f, vs);
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float pcx = u * tsx + -0.5f, pcy = v * tsy + 0.5f;
  const float flx = __builtin_floorf(pcx), fly = __builtin_floorf(pcy);
  const float tcx = mix_rt_((flx + 0.5f) / tsx, (flx + 1.0f) / tsx, vs);
  const float tcy = mix_rt_((fly + 0.5f) / tsy, (fly + -0.5f) / tsy, vs);
  const float frx = pcx - flx, fry = pcy - fly;
  const float fpx = mix_rt_(frx, fry, vs), fpy = mix_rt_(fry, frx, vs);
  const uint8_t* img = frame_ptr(L.in, z);
  float c[2][4][4];
  for (int r = 0; r < 2; ++r)
    for (int k = 0; k < 4; ++k) {
      const float kx = (float)(k - 1);
      float su = tcx + kx * dxx, sv = tcy + kx * dxy;
      if (r == 0) {
        su -= dyx;
        sv -= dyy;
      }
      if (TEXEL_LUT) {
        const int sx = clampi((int)__builtin_floorf(su * tsx), 0, L.in.w - 1), sy = clampi((int)__builtin_floorf(sv * tsy), 0, L.in.h - 1);
        const uint32_t p = *reinterpret_cast<const uint32_t*>(img + texel_off(L.in.w, sx, sy, 4u));
        c[r][k][0] = gin_rgb[p & 255u];
        c[r][k][1] = gin_rgb[(p >> 8) & 255u];
        c[r][k][2] = gin_rgb[(p >> 16) & 255u];
        c[r][k][3] = gin_a[p >> 24];
      } else {
        const float4 t = sample_rt(L.in, img, su, sv, &lds);
        c[r][k][0] = pow_(t.x, gin);
        c[r][k][1] = pow_(t.y, gin);
        c[r][k][2] = pow_(t.z, gin);
        c[r][k][3] = pow_(t.w, gin);
      }
    }
  const float l0 = fpx * fpx * fpx, l1 = fpx * fpx, l2 = fpx;
  float ip[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) ip[r] = ((m[r] * l0 + m[4 + r] * l1) + m[8 + r] * l2) + m[12 + r] * 1.0f;
  // row 1 has a literal 0 in column 2 and the last lobe is the literal 1: two products and a plain addend, which joins
  // the inner product (oracle/rc_passes_glow.c: in-situ float probe, bit-identical)
  ip[1] = m[4 + 1] * l1 + (m[1] * l0 + m[12 + 1]);
  const float pos0 = fpy, pos1 = 1.0f - fpy;
  float out[4];
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) {
    float col[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float v0 = ((c[r][0][ch] * ip[0] + c[r][1][ch] * ip[1]) + c[r][2][ch] * ip[2]) + c[r][3][ch] * ip[3];
      const float mn = minps(c[r][1][ch], c[r][2][ch]), mx = maxps(c[r][1][ch], c[r][2][ch]);
      col[r] = mix_rt_(v0, clamp_ps(v0, mn, mx), anti);
    }
    const float lum0 = mix_rt_(bmin, bmax, col[0]), lum1 = mix_rt_(bmin, bmax, col[1]);
    float d0 = (scan * pos0) / (lum0 + 0.0000001f), d1 = (scan * pos1) / (lum1 + 0.0000001f);
    // exp(-d*d): the constant of exp(x) = exp2(x * log2e) moves onto one factor (oracle)
    d0 = exp2_(d0 * (d0 * -1.4426950408889634f));
    d1 = exp2_(d1 * (d1 * -1.4426950408889634f));
    out[ch] = pow_(boost * (col[0] * d0 + col[1] * d1), 1.0f / gout);
  }
  store_rt(L, z, x, y, make_float4(out[0], out[1], out[2], out[3]), &lds);
  RC_TILE_LOOP_END
}

// resolve2.glsl mask_weights (129-400), layouts 3 and 6..19: cell [w][z] of the layout's table, w = floor(mod(coord.y, ny)),
// z = floor(mod(coord.x, nx)) at coord = pixel + 0.5 (the float mod equals the integer one: the quotient stays 0.03 away
// from every integer).  The shader's ternary chains test w == 1, w == 2, ... and leave the last row to w == 0; the rows
// here are indexed by w.  A cell is three bits - red, green, blue: 1 where the colour has the channel, 1 - MASK_INTENSITY
// elsewhere - and a row packs up to 14 cells.  Layout 12 compares a never-written `w`: the GL takes its second table on
// every row (llvmpipe golden, tests/golden/crt_hyllian_glow_layouts_*).
constexpr uint64_t mask_row(const char* r) {
  uint64_t v = 0;
  for (int i = 0; r[i]; ++i) {
    const char c = r[i];
    const uint64_t b = c == 'R' ? 1 : c == 'G' ? 2 : c == 'B' ? 4 : c == 'M' ? 5 : c == 'Y' ? 3 : c == 'C' ? 6 : 0;
    v |= b << (3 * i);
  }
  return v;
}
struct MaskLayout {
  int nx, ny;
  uint64_t rows[6];
};
__constant__ MaskLayout k_mask_layouts[20] = {
    {0, 0, {}}, {0, 0, {}}, {0, 0, {}},
    {4, 3, {mask_row("KKMG"), mask_row("MGKK"), mask_row("MGMG")}},
    {0, 0, {}}, {0, 0, {}},
    {4, 1, {mask_row("RGBK")}},
    {5, 1, {mask_row("RMBGG")}},
    {7, 1, {mask_row("RRYGCBB")}},
    {4, 1, {mask_row("RYCB")}},
    {4, 1, {mask_row("RMCG")}},
    {4, 2, {mask_row("BKRG"), mask_row("RGBK")}},
    {4, 1, {mask_row("CBRY")}},
    {4, 4, {mask_row("CBRY"), mask_row("RYCB"), mask_row("RYCB"), mask_row("CBRY")}},
    {6, 3, {mask_row("KKKMGK"), mask_row("MGKKKK"), mask_row("MGKMGK")}},
    {8, 4, {mask_row("KKKKRYCB"), mask_row("RYCBRYCB"), mask_row("RYCBKKKK"), mask_row("RYCBRYCB")}},
    {4, 3, {mask_row("KKYB"), mask_row("YBKK"), mask_row("YBYB")}},
    {10, 4, {mask_row("RRKKKKBBGG"), mask_row("RMBGGRMBGG"), mask_row("KBBGGRRKKK"), mask_row("RMBGGRMBGG")}},
    {10, 4, {mask_row("RRKKKKGGBB"), mask_row("RYGBBRYGBB"), mask_row("KGGBBRRKKK"), mask_row("RYGBBRYGBB")}},
    {14, 6, {mask_row("KKKKKKKKRRYGCB"), mask_row("RRYGCBBRRYGCBB"), mask_row("RRYGCBBRRYGCBB"), mask_row("RRYGCBBKKKKKKK"), mask_row("RRYGCBBRRYGCBB"),
             mask_row("RRYGCBBRRYGCBB")}},
};

// params: BLOOM_STRENGTH, OUTPUT_GAMMA, PHOSPHOR_LAYOUT, MASK_INTENSITY; extra[0] = PassPrev4Texture.
__global__ void __launch_bounds__(256) k_hyllian_resolve2(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float strength = L.params[0], gamma = L.params[1], intensity = L.params[3];
  const int layout = (int)L.params[2];
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float4 s = sample_rt(L.extra[0], frame_ptr(L.extra[0], z), u, v, &lds);
  const float4 b = sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds);
  const float fx = (float)x + 0.5f, fy = (float)y + 0.5f;
  const float mx = __builtin_floorf(fx - 2.0f * __builtin_floorf(fx / 2.0f)), my = __builtin_floorf(fy - 2.0f * __builtin_floorf(fy / 2.0f));
  const float on = 1.0f, off = 1.0f - intensity;
  float w3[3] = {1.0f, 1.0f, 1.0f};
  if (layout == 3 || (layout >= 6 && layout <= 19)) {
    const MaskLayout& m = k_mask_layouts[layout];
    const uint32_t bits = (uint32_t)(m.rows[y % m.ny] >> (3 * (x % m.nx)));
#pragma unroll
    for (int k = 0; k < 3; ++k) w3[k] = ((bits >> k) & 1u) ? on : off;
  } else if (layout == 1 || layout == 2 || layout == 4 || layout == 5) {
    const bool rgb = layout == 1 || layout == 2;   // magenta / green columns; else yellow / blue
    const float a3[3] = {on, rgb ? off : on, rgb ? on : off}, b3[3] = {off, rgb ? on : off, rgb ? off : on};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float ap = a3[k] + mx * (b3[k] - a3[k]);
      w3[k] = ap;
      if (layout == 2 || layout == 5) {
        const float inv = b3[k] + mx * (a3[k] - b3[k]);
        w3[k] = ap + my * (inv - ap);
      }
    }
  }
  const float s3[3] = {s.x, s.y, s.z}, bl[3] = {b.x, b.y, b.z};
  float out[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) out[k] = pow_(clamp_ps((1.15f * s3[k] + strength * bl[k]) * w3[k], 0.0f, 1.0f), 1.0f / gamma);
  store_rt(L, z, x, y, make_float4(out[0], out[1], out[2], 1.0f), &lds);
  RC_TILE_LOOP_END
}

}  // namespace

namespace rck {
#define GOK(K)                                                         \
  hipLaunchKernelGGL(K, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);              \
  return hipGetLastError()
static bool bytes_nearest_edge(const rcd::Tex& t) {
  return (t.fmt == FMT_SRGB8 || t.fmt == FMT_RGBA8 || t.fmt == FMT_RGBX8) && !t.linear && t.wrap == WRAP_EDGE && t.n_levels <= 1;
}
static bool bytemap_ok(const PassLaunch& L) {
  return bytes_nearest_edge(L.in) && (L.out_fmt == FMT_SRGB8 || L.out_fmt == FMT_RGBA8) && !(L.flags & RC_FLAG_GENERAL_ONLY);
}
hipError_t launch_glow_linearize(const PassLaunch& L, hipStream_t s) {
  if (bytemap_ok(L)) { GOK(k_glow_bytemap<0>); }
  GOK(k_glow_linearize);
}
hipError_t launch_glow_threshold(const PassLaunch& L, hipStream_t s) {
  if (bytemap_ok(L)) { GOK(k_glow_bytemap<1>); }
  GOK(k_glow_threshold);
}
hipError_t launch_glow_blur_h(const PassLaunch& L, hipStream_t s) { GOK(k_glow_blur<true>); }
hipError_t launch_glow_blur_v(const PassLaunch& L, hipStream_t s) { GOK(k_glow_blur<false>); }
hipError_t launch_crt_hyllian_glow(const PassLaunch& L, hipStream_t s) {
  if (bytes_nearest_edge(L.in) && !(L.flags & RC_FLAG_GENERAL_ONLY)) { GOK(k_crt_hyllian_glow<true>); }
  GOK(k_crt_hyllian_glow<false>);
}
hipError_t launch_hyllian_resolve2(const PassLaunch& L, hipStream_t s) { GOK(k_hyllian_resolve2); }
}  // namespace rck
