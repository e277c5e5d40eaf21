// Pass kernels for three reference shader assets (arithmetic spec = the GLSL text):
//   stock.glsl                         shaders/shaders_glsl/stock.glsl
//   scanlines/shaders/scanline.glsl    VS line 50, FS lines 107-113
//   motionblur/shaders/mix_frames.glsl VS lines 51-55, FS lines 94-107 (samples PrevTexture)
//   (crt/shaders/crt-pi.glsl: pass_crt_pi.hip)
// One thread per target pixel; blockIdx.z = frame of the batch.
#include "pass_launch.h"

using namespace rcd;

namespace {

// llvmpipe's blit fast path for a pure copy of an RGBA8 texture to a plain RGBA8 target with clamp to
// edge: the texture coordinate is stepped in 16.16 fixed point, re-anchored every 64 target pixels;
// LINEAR takes the texel pair and an 8-bit weight from the same coordinate minus half a texel, lerps
// on bytes (a + (((b-a)*w) >> 8)), horizontally first in 64x64 target tiles that need no clamping and
// vertically first in the others (formulas fitted on the GL, see oracle/rc_passes_basic.c).
__device__ __forceinline__ int blit_coord(float a0, float d, int texsize, int x) {
  const float T = (float)texsize, K = 65536.0f;
  const float fd = d * T;
  const int D = (int)(fd * K);
  const int j = x & ~63;
  const float s0 = fd * (float)j + a0 * T;
  const int S0 = (int)(s0 * K);
  return S0 + (x - j) * D;
}
__device__ __forceinline__ bool blit_tile_inside(float a0, float d, int texsize, int x, int extent) {
  const int j = x & ~63, last = min(j + 63, extent - 1);
  const int lo = (blit_coord(a0, d, texsize, j) - 32768) >> 16, hi = (blit_coord(a0, d, texsize, last) - 32768) >> 16;
  return lo >= 0 && hi + 1 <= texsize - 1;
}
__device__ __forceinline__ int lerp8(int a, int b, int w) { return a + (((b - a) * w) >> 8); }
template <bool LINEAR>
__global__ void __launch_bounds__(256) k_stock_blit(const PassLaunch L) {
  RC_TILE_LOOP_BEGIN
  (void)lo;
  const uint8_t* img = frame_ptr(L.in, z);
  uint32_t* dst = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(L.out) + L.out_frame_stride * (uint64_t)z + texel_off(L.out_w, x, y, 4u));
  const int cx = blit_coord(L.plane[0].a0_lo, L.plane[0].dx_lo, L.in.w, x), cy = blit_coord(L.plane[1].a0_lo, L.plane[1].dy_lo, L.in.h, y);
  if (!LINEAR) {
    *dst = *reinterpret_cast<const uint32_t*>(img + texel_off(L.in.w, clampi(cx >> 16, 0, L.in.w - 1), clampi(cy >> 16, 0, L.in.h - 1), 4u));
    continue;
  }
  const int sx = cx - 32768, sy = cy - 32768;
  const int x0 = clampi(sx >> 16, 0, L.in.w - 1), x1 = clampi((sx >> 16) + 1, 0, L.in.w - 1), wx = (sx >> 8) & 255;
  const int y0 = clampi(sy >> 16, 0, L.in.h - 1), y1 = clampi((sy >> 16) + 1, 0, L.in.h - 1), wy = (sy >> 8) & 255;
  const bool inside = blit_tile_inside(L.plane[0].a0_lo, L.plane[0].dx_lo, L.in.w, x, (L.out_w + 3) & ~3) &&   // spans are 4 px wide
                      blit_tile_inside(L.plane[1].a0_lo, L.plane[1].dy_lo, L.in.h, y, L.out_h);
  const uint32_t pa = *reinterpret_cast<const uint32_t*>(img + texel_off(L.in.w, x0, y0, 4u));
  const uint32_t pb = *reinterpret_cast<const uint32_t*>(img + texel_off(L.in.w, x1, y0, 4u));
  const uint32_t pc = *reinterpret_cast<const uint32_t*>(img + texel_off(L.in.w, x0, y1, 4u));
  const uint32_t pd = *reinterpret_cast<const uint32_t*>(img + texel_off(L.in.w, x1, y1, 4u));
  uint32_t o = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int A = (pa >> (8 * c)) & 255, B = (pb >> (8 * c)) & 255, Cc = (pc >> (8 * c)) & 255, D = (pd >> (8 * c)) & 255;
    const int v = inside ? lerp8(lerp8(A, B, wx), lerp8(Cc, D, wx), wy) : lerp8(lerp8(A, Cc, wy), lerp8(B, D, wy), wx);
    o |= (uint32_t)(v & 255) << (8 * c);
  }
  *dst = o;
  RC_TILE_LOOP_END
}

__global__ void __launch_bounds__(256) k_stock(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  store_rt(L, z, x, y, sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds), &lds);
  RC_TILE_LOOP_END
}

// params: SCANLINE_BASE_BRIGHTNESS, SCANLINE_SINE_COMP_A, SCANLINE_SINE_COMP_B, size
__global__ void __launch_bounds__(256) k_scanline(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float base = L.params[0], comp_a = L.params[1], comp_b = L.params[2], size = L.params[3];
  const float pi = 3.141592654f;
  const float omega_x = (pi * size) * (float)L.out_w;
  const float omega_y = (2.0f * pi) * (float)L.in.h;
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float4 res = sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds);
  const float d = comp_a * sin_(u * omega_x) + comp_b * sin_(v * omega_y);
  const float k = base + d;
  store_rt(L, z, x, y, make_float4(res.x * k, res.y * k, res.z * k, 1.0f), &lds);
  RC_TILE_LOOP_END
}

// extra[0] = PrevTexture; mix(colour, colourPrev, 0.5) with a constant weight = a*(1-0.5) + b*0.5
// plane[0], plane[1]: TEX0 = TexCoord * 1.0001
__global__ void __launch_bounds__(256) k_mix_frames(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float4 c = sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds);
  const float4 p = sample_rt(L.extra[0], frame_ptr(L.extra[0], z), u, v, &lds);
  store_rt(L, z, x, y, make_float4(c.x * 0.5f + p.x * 0.5f, c.y * 0.5f + p.y * 0.5f, c.z * 0.5f + p.z * 0.5f, 1.0f), &lds);
  RC_TILE_LOOP_END
}

// handheld/shaders/color/{gba,gbc,nds,palm,psp,vba}-color.glsl FS main (handheld/<name>-color.glslp and the lcd-grid-v2-* chains):
// pow(texel, gamma_in) * lum, clamp, a constant 3x3 matrix, pow(.., 1 / display_gamma), alpha 0.  The constants and the three
// places where the GL's compiled form differs between the six files come from the registry (kernel_registry.cpp, setupColor*):
// params[8] gamma_in, [9] lum, [10..18] the matrix by output channel, [19] 1 / display_gamma, [20] != 0: psp's blue row,
// 0.01 * (R + G) + 0.98 * B.  A zero coefficient of the third column drops its term (vba red), as in the GL.
__global__ void __launch_bounds__(256) k_color_matrix(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  const float* P = L.params;
  const float gin = P[8], lum = P[9], inv = P[19];
  const bool factored_blue = P[20] != 0.0f;
  RC_TILE_LOOP_BEGIN
  const float4 t = sample_rt(L.in, frame_ptr(L.in, z), vary(L.plane[0], x, y, lo), vary(L.plane[1], x, y, lo), &lds);
  float c[3] = {pow_(t.x, gin) * lum, pow_(t.y, gin) * lum, pow_(t.z, gin) * lum}, o[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    c[k] = c[k] > 0.0f ? c[k] : 0.0f;   // fmax(x, 0) then fmin(., 1) as MAXPS / MINPS
    c[k] = c[k] < 1.0f ? c[k] : 1.0f;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float v = (k == 2 && factored_blue) ? P[16] * (c[0] + c[1]) : P[10 + 3 * k] * c[0] + P[11 + 3 * k] * c[1];
    if (P[12 + 3 * k] != 0.0f) v = v + P[12 + 3 * k] * c[2];
    o[k] = v != v ? 0.0f : pow_(v, inv);   // llvmpipe's pow of a NaN base is 0
  }
  store_rt(L, z, x, y, make_float4(o[0], o[1], o[2], 0.0f), &lds);
  RC_TILE_LOOP_END
}

// handheld/shaders/retro-v2.glsl FS main (handheld/retro-v2.glslp, presets/retro-v2+<console>-color.glslp): pow(texel, 2.4); the edge
// of every source pixel is darkened over a width set by RETRO_PIXEL_SIZE; pow(.., 1 / 2.2), clamp.  params[0] = RETRO_PIXEL_SIZE.
// Operation order: the GL's instruction listing (oracle/rc_passes_basic.c).  fmin / fmax keep the operand that is not NaN.
__global__ void __launch_bounds__(256) k_retro_v2(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  const float rps = L.params[0], tsx = (float)L.in.w, tsy = (float)L.in.h;
  const float px = tsx * (1.0f / (float)L.out_w), py = tsy * (1.0f / (float)L.out_h);   // InputSize * (1 / OutputSize)
  auto nmin = [](float a, float b) { return b != b ? a : (a < b ? a : b); };
  auto nmax = [](float a, float b) { return b != b ? a : (a > b ? a : b); };
  RC_TILE_LOOP_BEGIN
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float4 t = sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds);
  const float su = u * tsx, sv = v * tsy, fx = su - __builtin_floorf(su), fy = sv - __builtin_floorf(sv);
  const float ax = nmin(nmax(fx + 0.5f * px, 0.0f), 1.0f), ay = nmin(nmax(fy + 0.5f * py, 0.0f), 1.0f);
  const float cx = nmin(nmax(ax + -rps, 0.0f), px) / px, cy = nmin(nmax(ay + -rps, 0.0f), py) / py;
  const float m = nmax(cx, cy);
  const float k = (1.04f + fx * fy) * (1.0f + -m) + 0.36f * m;
  const float c3[3] = {t.x, t.y, t.z};
  float o[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float q = k * pow_(c3[c], 2.4f);
    o[c] = nmin(nmax(q != q ? 0.0f : pow_(q, 1.0f / 2.2f), 0.0f), 1.0f);
  }
  store_rt(L, z, x, y, make_float4(o[0], o[1], o[2], 1.0f), &lds);
  RC_TILE_LOOP_END
}

// handheld/console-border/shader-files/gb-pass-5.glsl FS main (last pass of the console-border presets): the frame under a border image
// blended in by its own alpha, frame + a (border - frame) on all four channels.  plane[0..1] = TEX0 (the frame scaled about its
// centre), plane[2..3] = tex_border (registry: setupGbPass5); extra[0] = BORDER.
__global__ void __launch_bounds__(256) k_gb_pass_5(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float4 b = sample_rt(L.extra[0], frame_ptr(L.extra[0], z), vary(L.plane[2], x, y, lo), vary(L.plane[3], x, y, lo), &lds);
  const float4 f = sample_rt(L.in, frame_ptr(L.in, z), vary(L.plane[0], x, y, lo), vary(L.plane[1], x, y, lo), &lds);
  store_rt(L, z, x, y, make_float4(f.x + b.w * (b.x + -f.x), f.y + b.w * (b.y + -f.y), f.z + b.w * (b.z + -f.z), f.w + b.w * (b.w + -f.w)), &lds);
  RC_TILE_LOOP_END
}

// borders/resources/imgborder-{sgb,gameboy-player,sgba}.glsl FS 165-173 (one text, three sets of defaults): the frame inside a border image
// that covers it by its own alpha, or not at all inside the viewport when border_on_top is set.  plane[0..1] = screen_coord,
// plane[2..3] = TEX0 (registry: setupImgBorder); params[5] border_on_top, [8..11] OS_MASK_TOP / BOTTOM / LEFT / RIGHT; extra[0] = BORDER.
__global__ void __launch_bounds__(256) k_imgborder(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  const float* P = L.params;
  const float x_hi = 0.9999f + -P[11], x_lo = 0.0001f + P[10], y_hi = 0.9999f + -P[9], y_lo = 0.0001f + P[8];
  const bool on_top = 0.5f < P[5];
  RC_TILE_LOOP_BEGIN
  const float sx = vary(L.plane[0], x, y, lo), sy = vary(L.plane[1], x, y, lo);
  const float4 f = sample_rt(L.in, frame_ptr(L.in, z), sx, sy, &lds);
  const float4 b = sample_rt(L.extra[0], frame_ptr(L.extra[0], z), vary(L.plane[2], x, y, lo), vary(L.plane[3], x, y, lo), &lds);
  const bool inside = sx < x_hi && x_lo < sx && sy < y_hi && y_lo < sy && on_top;
  const float al = inside ? 0.0f : b.w;
  store_rt(L, z, x, y, make_float4(f.x + al * (b.x + -f.x), f.y + al * (b.y + -f.y), f.z + al * (b.z + -f.z), f.w + al * (al + -f.w)), &lds);
  RC_TILE_LOOP_END
}

// reshade/shaders/LUT/LUT.glsl FS main (reshade/{lut,gba,nds,vba,bsnes-gamma-ramp,spfft}.glslp): a colour LUT of LUT_Size slices side by
// side, sampled twice and mixed along blue - where the first sample's blue is below 1.  As the shader is written,
// ceil(b + 0.000001 * (LUT_Size - 1)) rounds the colour itself up: the second slice is slice 1 (kept).  params[0] = LUT_Size;
// extra[0] = SamplerLUT.  fmin / fmax keep the operand that is not NaN (0 / 0 where both slices coincide).
__global__ void __launch_bounds__(256) k_lut(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  const float S = L.params[0], k = S + -1.0f;
  auto nmin = [](float a, float b) { return b != b ? a : (a < b ? a : b); };
  auto nmax = [](float a, float b) { return b != b ? a : (a > b ? a : b); };
  RC_TILE_LOOP_BEGIN
  const float4 c = sample_rt(L.in, frame_ptr(L.in, z), vary(L.plane[0], x, y, lo), vary(L.plane[1], x, y, lo), &lds);
  const float red = (c.x * k + 0.4999f) / (S * S), green = (c.y * k + 0.4999f) / S;
  const float b1 = __builtin_floorf(c.z * k) / S + red, b2 = __builtin_ceilf(c.z + 0.000001f * k) / S + red;
  const uint8_t* lut = frame_ptr(L.extra[0], z);
  const float4 c1 = sample_rt(L.extra[0], lut, b1, green, &lds), c2 = sample_rt(L.extra[0], lut, b2, green, &lds);
  const float m = nmin(nmax((c.z + -b1) / (b2 + -b1), 0.0f), 32.0f);
  float4 o = c1;
  if (c1.z < 1.0f) o = make_float4(c1.x + m * (c2.x + -c1.x), c1.y + m * (c2.y + -c1.y), c1.z + m * (c2.z + -c1.z), c1.w + m * (c2.w + -c1.w));
  store_rt(L, z, x, y, o, &lds);
  RC_TILE_LOOP_END
}

// handheld/shaders/gb-palette/gb-palette.glsl FS main (handheld/gb-palette-{dmg,light,pocket}.glslp): the red channel as a grey level picks a
// row of the palette image (column 0.5); alpha = ceil(|1 - r|).  extra[0] = COLOR_PALETTE.
__global__ void __launch_bounds__(256) k_gb_palette(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float4 c = sample_rt(L.in, frame_ptr(L.in, z), vary(L.plane[0], x, y, lo), vary(L.plane[1], x, y, lo), &lds);
  const float g = __builtin_fabsf(1.0f + -c.x);
  const float4 p = sample_rt(L.extra[0], frame_ptr(L.extra[0], z), 0.5f, g * 0.75f + 0.125f, &lds);
  store_rt(L, z, x, y, make_float4(p.x, p.y, p.z, __builtin_ceilf(g)), &lds);
  RC_TILE_LOOP_END
}

// crt/shaders/crt-potato/shader-files/crt-potato.glsl FS main (crt/crt-potato-{cool,warm}.glslp): the frame times a small mask image tiled
// every 2 target pixels across and every floor(OutputSize.y / InputSize.y + 0.000001) lines down; gl_FragCoord = pixel + 0.5.  extra[0] = MASK.
__global__ void __launch_bounds__(256) k_crt_potato(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  const float scale = __builtin_floorf((float)L.out_h / (float)L.in.h + 0.000001f);
  RC_TILE_LOOP_BEGIN
  const float fx = ((float)x + 0.5f) / 2.0f, fy = ((float)y + 0.5f) / scale;
  const float4 m = sample_rt(L.extra[0], frame_ptr(L.extra[0], z), fx - __builtin_floorf(fx), fy - __builtin_floorf(fy), &lds);
  const float4 c = sample_rt(L.in, frame_ptr(L.in, z), vary(L.plane[0], x, y, lo), vary(L.plane[1], x, y, lo), &lds);
  store_rt(L, z, x, y, make_float4(m.x * c.x, m.y * c.y, m.z * c.z, m.w * c.w), &lds);
  RC_TILE_LOOP_END
}

// ntsc/shaders/ntsc-gauss-pass.glsl FS main (ntsc/ntsc-*-gauss-scanline.glslp, ntsc.glslp, ntsc-svideo.glslp): five source lines around the
// target pixel, pow(., NTSC_CRT_GAMMA), weighted with exp(-5 d^2) = exp2((-7.213475 d) d) of the line distance, x 1.15,
// pow(., 1 / NTSC_DISPLAY_GAMMA).  plane[2] = pix_no (TexCoord.y * TextureSize.y), plane[3] = one (1 / TextureSize.y): setupNtscGauss.
__global__ void __launch_bounds__(256) k_ntsc_gauss(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  const float crt = L.params[0], inv = 1.0f / L.params[1];
  RC_TILE_LOOP_BEGIN
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo), o1 = vary(L.plane[3], x, y, lo);
  const float pno = vary(L.plane[2], x, y, lo), fr = pno - __builtin_floorf(pno);
  const float off[5] = {-2.0f * o1, -o1, 0.0f, o1, 2.0f * o1}, d[5] = {1.5f + fr, 0.5f + fr, fr + -0.5f, -1.5f + fr, -2.5f + fr};
  const uint8_t* img = frame_ptr(L.in, z);
  float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const float4 t = sample_rt(L.in, img, u, k == 2 ? v : v + off[k], &lds);
    const float w = exp2_((-7.213475f * d[k]) * d[k]);
    const float c[3] = {pow_(t.x, crt) * w, pow_(t.y, crt) * w, pow_(t.z, crt) * w};
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] = k == 0 ? c[q] : acc[q] + c[q];
  }
  store_rt(L, z, x, y, make_float4(pow_(1.15f * acc[0], inv), pow_(1.15f * acc[1], inv), pow_(1.15f * acc[2], inv), 1.0f), &lds);
  RC_TILE_LOOP_END
}

// misc/interlacing.glsl FS main (24 presets): every other line dimmed to `percent`; above 400 source lines the field alternates with FrameCount
// (enable_480i), top_field_first shifts it.  params: percent, enable_480i, top_field_first, [8] = TextureSize.y as the reference hands it over
// (the pass-index-3 rule was written for this shader; registry: setupInterlacing).
__global__ void __launch_bounds__(256) k_interlacing(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  const float* P = L.params;
  const float isy = (float)L.in.h, tsy = P[8];
  RC_TILE_LOOP_BEGIN
  const float fc = (float)(L.frame_count0 + z);
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float4 c = sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds);
  const float line = 400.0f < isy ? (tsy * v + P[2]) + fc * P[1] : (2.000001f * tsy) * v + P[2];
  const float m = line + -(1.99999f * __builtin_floorf(line / 1.99999f));
  const bool keep = 0.99999f < m;
  store_rt(L, z, x, y, keep ? c : make_float4(P[0] * c.x, P[0] * c.y, P[0] * c.z, P[0] * c.w), &lds);
  RC_TILE_LOOP_END
}

// handheld/shaders/mgba/agb001.glsl FS main (handheld/agb001.glslp, agb001-gba-color-motionblur.glslp): pow(texel * 0.8, 1.8) + 0.16 under
// a 4x4 subpixel pattern per source texel - column 0 / 1 / 2 keeps red / green / blue and takes the other two to 0.2, column 3
// takes all to 0.4, row 3 another 0.8 - alpha 0.5.  Index: int(mod(coord * size * 4, 4)), mod as a - 4 floor(a / 4).
__global__ void __launch_bounds__(256) k_agb001(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  const float tsx = (float)L.in.w, tsy = (float)L.in.h;
  RC_TILE_LOOP_BEGIN
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float4 t = sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds);
  float c[3] = {pow_(t.x * 0.8f, 1.8f) + 0.16f, pow_(t.y * 0.8f, 1.8f) + 0.16f, pow_(t.z * 0.8f, 1.8f) + 0.16f};
  const float ax = (u * tsx) * 4.0f, ay = (v * tsy) * 4.0f;
  const float mx = ax + -(4.0f * __builtin_floorf(ax / 4.0f)), my = ay + -(4.0f * __builtin_floorf(ay / 4.0f));
  const int ix = mx != mx ? (-2147483647 - 1) : (int)mx, iy = my != my ? (-2147483647 - 1) : (int)my;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (ix >= 0 && ix <= 2) {
      if (k != ix) c[k] = c[k] * 0.2f;
    } else {
      c[k] = c[k] * 0.4f;
    }
    if (!((unsigned)iy <= 2u)) c[k] = c[k] * 0.8f;
  }
  store_rt(L, z, x, y, make_float4(c[0], c[1], c[2], 0.5f), &lds);
  RC_TILE_LOOP_END
}

// handheld/shaders/color/gbc-gambatte-color.glsl FS main: a fixed matrix on the texel as sampled (products blue, green, red), alpha kept
__global__ void __launch_bounds__(256) k_gbc_gambatte_color(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float4 t = sample_rt(L.in, frame_ptr(L.in, z), vary(L.plane[0], x, y, lo), vary(L.plane[1], x, y, lo), &lds);
  const float g8 = t.y * 0.125f;
  store_rt(L, z, x, y, make_float4((t.z * 0.0625f + g8) + t.x * 0.8125f, t.z * 0.25f + t.y * 0.75f, (t.z * 0.6875f + g8) + t.x * 0.1875f, t.w), &lds);
  RC_TILE_LOOP_END
}

// stereoscopic-3d/shaders/shutter-3d.glsl FS 123-143 (stereoscopic-3d/shutter-to-side-by-side.glslp): the left eye's frame
// and the right eye's side by side, alternating with FrameCount parity, the other eye held from PrevTexture x flicker.
// params: ZOOM, vert_pos, horz_pos, separation, flicker, height_mod, swap_eye; extra[0] = PrevTexture;
// plane[0..3] = left_coord.xy, right_coord.xy (registry: setupShutter3d).  Operation order: the GL's instruction listing.
__global__ void __launch_bounds__(256) k_shutter_3d(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  const float isx = (float)L.in.w, isy = (float)L.in.h, flicker = L.params[4];
  RC_TILE_LOOP_BEGIN
  const float fc = (float)(L.frame_count0 + z);
  const float timer = __builtin_fabsf(L.params[6] + -(fc + -(2.0f * __builtin_floorf(fc / 2.0f))));   // |swap_eye - mod(FrameCount, 2)|
  const float omt = 1.0f + -timer;
  const float lx = vary(L.plane[0], x, y, lo), ly = vary(L.plane[1], x, y, lo), rx = vary(L.plane[2], x, y, lo), ry = vary(L.plane[3], x, y, lo);
  const uint8_t* img = frame_ptr(L.in, z);
  const uint8_t* prev = frame_ptr(L.extra[0], z);
  const float4 l = sample_rt(L.in, img, lx, ly, &lds), r = sample_rt(L.in, img, rx, ry, &lds);
  const float4 lh = sample_rt(L.extra[0], prev, lx, ly, &lds), rh = sample_rt(L.extra[0], prev, rx, ry, &lds);
  const float lcx = (lx * isx) / isx, lcy = (ly * isy) / isy, rcx = (rx * isx) / isx, rcy = (ry * isy) / isy;   // coord * InputSize / TextureSize
  const float lm = lcy != lcy ? lcx : (lcx < lcy ? lcx : lcy), rm = rcy != rcy ? rcx : (rcx < rcy ? rcx : rcy);
  const float ml = (0.0001f < lm && lcx < 0.9999f && lcy < 0.9999f) ? 1.0f : 0.0f;
  const float mr = (0.0001f < rm && rcx < 0.9999f && rcy < 0.9999f) ? 1.0f : 0.0f;
  const float l4[4] = {l.x, l.y, l.z, l.w}, lh4[4] = {lh.x, lh.y, lh.z, lh.w}, r4[4] = {r.x, r.y, r.z, r.w}, rh4[4] = {rh.x, rh.y, rh.z, rh.w};
  float o[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) o[c] = (l4[c] * timer + (omt * lh4[c]) * flicker) * ml + (r4[c] * omt + (rh4[c] * timer) * flicker) * mr;
  store_rt(L, z, x, y, make_float4(o[0], o[1], o[2], o[3]), &lds);
  RC_TILE_LOOP_END
}

// misc/anti-flicker.glsl FS 99-127: blends the previous frame in where the luma jumps against it but not against the frame
// before.  params: lum_diff_thresh; extra[0] = PrevTexture, extra[1] = Prev1Texture.  The YIQ products run blue, green, red.
__global__ void __launch_bounds__(256) k_anti_flicker(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  const float th = L.params[0];
  RC_TILE_LOOP_BEGIN
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float4 c = sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds);
  const float4 p0 = sample_rt(L.extra[0], frame_ptr(L.extra[0], z), u, v, &lds), p1 = sample_rt(L.extra[1], frame_ptr(L.extra[1], z), u, v, &lds);
  const float cy = (c.z * 0.114f + c.y * 0.587f) + c.x * 0.2989f, ci = (c.z * -0.3216f + c.y * -0.2744f) + c.x * 0.5959f;
  const float cq = (c.z * 0.3114f + c.y * -0.5229f) + c.x * 0.2115f;
  const float py = (p0.z * 0.114f + p0.y * 0.587f) + p0.x * 0.2989f, pi = (p0.z * -0.3216f + p0.y * -0.2744f) + p0.x * 0.5959f;
  const float pq = (p0.z * 0.3114f + p0.y * -0.5229f) + p0.x * 0.2115f;
  const float p1y = (p1.z * 0.114f + p1.y * 0.587f) + p1.x * 0.2989f;
  const bool blend = (th < __builtin_fabsf(cy + -py)) && (__builtin_fabsf(cy + -p1y) < 1.0f + -th);
  const float Y = blend ? (py + cy) / 2.0f : cy, I = blend ? (pi + ci) / 2.0f : ci, Q = blend ? (pq + cq) / 2.0f : cq;
  store_rt(L, z, x, y, make_float4((Q * 0.621f + Y) + I * 0.956f, (Q * -0.6474f + Y) + I * -0.272f, (Q * 1.7046f + Y) + I * -1.106f, 1.0f), &lds);
  RC_TILE_LOOP_END
}

// The other frame-history shaders of motionblur/ (oracle/rc_passes_basic.c); every texture is sampled at the pass's one
// coordinate (motionblur-simple's PrevNTexCoord attributes alias TexCoord's location in the reference).
// motionblur-simple.glsl FS 178-199: extra[0..6] = Prev6 .. Prev1, PrevTexture; c = (c + next) / 2 down to the current frame
__global__ void __launch_bounds__(256) k_motionblur_simple(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  float4 c = sample_rt(L.extra[0], frame_ptr(L.extra[0], z), u, v, &lds);
#pragma unroll
  for (int k = 1; k <= 7; ++k) {
    const float4 t = k < 7 ? sample_rt(L.extra[k], frame_ptr(L.extra[k], z), u, v, &lds) : sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds);
    c = make_float4((c.x + t.x) / 2.0f, (c.y + t.y) / 2.0f, (c.z + t.z) / 2.0f, (c.w + t.w) / 2.0f);
  }
  store_rt(L, z, x, y, c, &lds);
  RC_TILE_LOOP_END
}
// braid-rewind.glsl FS 135-160: FrameDirection is always 1 in the reference, so the history blend never applies
__global__ void __launch_bounds__(256) k_braid_rewind(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  store_rt(L, z, x, y, sample_rt(L.in, frame_ptr(L.in, z), vary(L.plane[0], x, y, lo), vary(L.plane[1], x, y, lo), &lds), &lds);
  RC_TILE_LOOP_END
}
// response-time.glsl FS 122-136: extra[0..6] = PrevTexture, Prev1 .. Prev6; params[0] = response_time, params[1..7] = its
// powers 1..7 as the GL evaluates them (host, kernel_registry.cpp); alpha 0
__global__ void __launch_bounds__(256) k_response_time(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  float4 c = sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds);
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const float4 p = sample_rt(L.extra[q], frame_ptr(L.extra[q], z), u, v, &lds);
    const float k = L.params[1 + q];
    c.x = c.x + (p.x - c.x) * k;
    c.y = c.y + (p.y - c.y) * k;
    c.z = c.z + (p.z - c.z) * k;
  }
  c.w = 0.0f;
  store_rt(L, z, x, y, c, &lds);
  RC_TILE_LOOP_END
}
// mix_frames_smart.glsl FS 64-105: extra[0..4] = PrevTexture, Prev1 .. Prev4; params[0] = DEFLICKER_EMPHASIS
__global__ void __launch_bounds__(256) k_mix_frames_smart(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float edge = 0.000001f + L.params[0];
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  float4 c[6];
  c[0] = sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds);
#pragma unroll
  for (int q = 0; q < 5; ++q) c[q + 1] = sample_rt(L.extra[q], frame_ptr(L.extra[q], z), u, v, &lds);
  auto is_eq = [&](int i, int j) { return (c[i].x == c[j].x && c[i].y == c[j].y && c[i].z == c[j].z) ? 1.0f : 0.0f; };
  auto is_aeq = [&](int i, int j) {
    return (!(__builtin_fabsf(c[i].x - c[j].x) >= edge) && !(__builtin_fabsf(c[i].y - c[j].y) >= edge) && !(__builtin_fabsf(c[i].z - c[j].z) >= edge)) ? 1.0f : 0.0f;
  };
  const float alt = is_aeq(0, 2) * is_aeq(2, 4) + is_aeq(1, 3) * is_aeq(3, 5);
  float m = (1.0f - is_eq(0, 3)) * (1.0f - is_eq(0, 5)) * (1.0f - is_eq(1, 2)) * (1.0f - is_eq(1, 4)) * (1.0f - is_eq(2, 3)) * (1.0f - is_eq(2, 5));
  m = m * (alt < 1.0f ? alt : 1.0f);
  const float t = m * 0.5f;
  store_rt(L, z, x, y, make_float4(c[0].x + t * (c[1].x - c[0].x), c[0].y + t * (c[1].y - c[0].y), c[0].z + t * (c[1].z - c[0].z), 1.0f), &lds);
  RC_TILE_LOOP_END
}

// Conformance fixture tests/fixtures/conformance/feedback-persist.glsl (this repository's own shader):
// max(cur*0.75 + old0*0.25, old1*PERSIST); extra[0] = PassFeedback0, extra[1] = PassFeedback1
__global__ void __launch_bounds__(256) k_feedback_persist(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float persist = L.params[0];
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float4 c = sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds);
  const float4 p0 = sample_rt(L.extra[0], frame_ptr(L.extra[0], z), u, v, &lds);
  const float4 p1 = sample_rt(L.extra[1], frame_ptr(L.extra[1], z), u, v, &lds);
  const float m0 = c.x * 0.75f + p0.x * 0.25f, m1 = c.y * 0.75f + p0.y * 0.25f, m2 = c.z * 0.75f + p0.z * 0.25f;
  const float q0 = p1.x * persist, q1 = p1.y * persist, q2 = p1.z * persist;
  store_rt(L, z, x, y, make_float4(m0 < q0 ? q0 : m0, m1 < q1 ? q1 : m1, m2 < q2 ? q2 : m2, 1.0f), &lds);
  RC_TILE_LOOP_END
}

// Conformance fixture tests/fixtures/conformance/history-size.glsl (this repository's own shader): a frame-history shader that
// reads TextureSize.x, OutputSize.y and InputSize.y / TextureSize.y - through PassLaunch::uni_* where the history re-draw leaves
// them stale (oracle/rc_passes_basic.c o_pass_history_size: operation order from llvmpipe's NIR).
// extra[0] = PrevTexture, extra[1] = Prev1Texture; params: HS_MIX
__global__ void __launch_bounds__(256) k_history_size(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float mixw = L.params[0];
  const float ts_x = (float)(L.uni_tex_w ? L.uni_tex_w : L.in.w), ts_y = (float)(L.uni_tex_h ? L.uni_tex_h : L.in.h);
  const float os_y = (float)(L.uni_out_h ? L.uni_out_h : L.out_h);
  const float inv = 1.0f / ts_x, cover = ts_y / ts_y;
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float u2 = u + inv;
  const float4 c = sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds), r = sample_rt(L.in, frame_ptr(L.in, z), u2, v, &lds);
  const float4 p0 = sample_rt(L.extra[0], frame_ptr(L.extra[0], z), u, v, &lds), p1 = sample_rt(L.extra[1], frame_ptr(L.extra[1], z), u2, v, &lds);
  const float row = __builtin_floorf(v * os_y), hrow = row * 0.5f, fr = hrow + (-__builtin_floorf(hrow));
  const float dim = fr < 0.25f ? 1.0f : 0.75f;
  const float cc[3] = {c.x, c.y, c.z}, rr[3] = {r.x, r.y, r.z}, a0[3] = {p0.x, p0.y, p0.z}, a1[3] = {p1.x, p1.y, p1.z};
  float o[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float now = cc[k] * 0.75f + rr[k] * 0.25f;
    const float t = (a0[k] * 0.625f + (-now)) + a1[k] * 0.375f;
    o[k] = ((now + t * mixw) * dim) * cover;
  }
  store_rt(L, z, x, y, make_float4(o[0], o[1], o[2], 1.0f), &lds);
  RC_TILE_LOOP_END
}

// crt/shaders/zfast_crt.glsl (FINEMASK), VS 101-108, FS 168-198; plane[0], plane[1]: TEX0 = TexCoord * 1.0001.
// params: BLURSCALEX, LOWLUMSCAN, HILUMSCAN, BRIGHTBOOST, MASK_DARK, MASK_FADE (always the reference's fixed values,
// ShaderEngine.cpp:2260-2294)
// GENERIC false: GL_RGB source (RGBX8) LINEAR clamp-to-edge and a plain RGBA8 target - the shipped single-pass preset
template <bool GENERIC>
__global__ void __launch_bounds__(256) k_zfast_crt(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float blur = L.params[0], lowlum = L.params[1], hilum = L.params[2], boost = L.params[3], mdark = L.params[4];
  const float mask_fade = 0.3333f * L.params[5];
  const float tsx = (float)L.in.w, tsy = (float)L.in.h, idx = 1.0f / tsx, idy = 1.0f / tsy;
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float px = u * tsx, py = v * tsy;
  const float ix = __builtin_floorf(px) + 0.5f, iy = __builtin_floorf(py) + 0.5f;
  const float fx = px - ix, fy = py - iy;
  float qx = (ix + ((4.0f * fx) * fx) * fx) * idx;
  const float qy = (iy + ((4.0f * fy) * fy) * fy) * idy;
  qx = qx + blur * (u - qx);
  const float Y = fy * fy, YY = Y * Y;
  const float wm0 = __builtin_floorf((u * (float)L.out_w) * -0.4999f);
  const float whichmask = wm0 - __builtin_floorf(wm0);
  const float mask = 1.0f + (whichmask < 0.5f ? 1.0f : 0.0f) * -mdark;
  const float4 c = GENERIC ? sample_rt(L.in, frame_ptr(L.in, z), qx, qy, &lds)
                           : sample<FMT_RGBX8, 1, WRAP_EDGE>(L.in, frame_ptr(L.in, z), qx, qy, &lds);
  const float slw = boost - lowlum * (Y - 2.05f * YY);
  const float slwb = 1.0f - hilum * (YY - (2.8f * YY) * Y);
  const float d = (c.x + (c.y + c.z)) * mask_fade;
  const float m0 = slw * mask;
  const float w = m0 + d * (slwb - m0);
  if (GENERIC) store_rt(L, z, x, y, make_float4(c.x * w, c.y * w, c.z * w, 1.0f), &lds);
  else store<FMT_RGBA8>(L, z, x, y, make_float4(c.x * w, c.y * w, c.z * w, 1.0f), &lds);
  RC_TILE_LOOP_END
}

// handheld/shaders/lcd1x.glsl (handheld/lcd1x.glslp), FS 104-121; plane[0], plane[1]: TEX0 = TexCoord * 1.0001.
// params: BRIGHTEN_SCANLINES, BRIGHTEN_LCD
__global__ void __launch_bounds__(256) k_lcd1x(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float bs = L.params[0], bl = L.params[1];
  const float two_pi = 2.0f * 3.141592654f;
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float ax = two_pi * (u * (float)L.in.w - 0.25f), ay = two_pi * (v * (float)L.in.h - 0.25f);
  const float k = ((bs + sin_(ay)) / (bs + 1.0f)) * ((bl + sin_(ax)) / (bl + 1.0f));
  const float4 c = sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds);
  store_rt(L, z, x, y, make_float4(k * c.x, k * c.y, k * c.z, 1.0f), &lds);
  RC_TILE_LOOP_END
}

// dithering/shaders/bayer-matrix-dithering.glsl, FS 99-141: 8x8 ordered dithering of every channel to 0 / 1.
// params: animate, dither_size; FrameCount is an int uniform.
__constant__ int k_bayer8[64] = {0, 32, 8, 40, 2, 34, 10, 42, 48, 16, 56, 24, 50, 18, 58, 26, 12, 44, 4, 36, 14, 46, 6, 38, 60, 28, 52, 20, 62, 30, 54, 22,
                                 3, 35, 11, 43, 1, 33, 9, 41, 51, 19, 59, 27, 49, 17, 57, 25, 15, 47, 7, 39, 13, 45, 5, 37, 63, 31, 55, 23, 61, 29, 53, 21};
__global__ void __launch_bounds__(256) k_bayer(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float fc2 = 2.0f * (float)(L.frame_count0 + z);
  const float scale = (3.0f + (fc2 - 32.0f * __builtin_floorf(fc2 / 32.0f)) * L.params[0]) + L.params[1];
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float4 c = sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds);
  const float xx = (u * (float)L.out_w) * scale, yy = (v * (float)L.out_h) * scale;
  const int ix = (int)(xx - 8.0f * __builtin_floorf(xx / 8.0f)), iy = (int)(yy - 8.0f * __builtin_floorf(yy / 8.0f));
  float limit = 0.0f;
  if (ix < 8) limit = (float)(k_bayer8[(ix & 7) * 8 + (iy & 7)] + 1) / 64.0f;
  store_rt(L, z, x, y, make_float4(c.x < limit ? 0.0f : 1.0f, c.y < limit ? 0.0f : 1.0f, c.z < limit ? 0.0f : 1.0f, 1.0f), &lds);
  RC_TILE_LOOP_END
}

// handheld/shaders/lcd3x.glsl (handheld/lcd3x.glslp), FS 95-110.  params: brighten_scanlines, brighten_lcd
__global__ void __launch_bounds__(256) k_lcd3x(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float bs = L.params[0], bl = L.params[1];
  const float pi = 3.141592654f;
  const float off[3] = {pi * (1.0f / 2.0f), pi * (1.0f / 2.0f - 2.0f / 3.0f), pi * (1.0f / 2.0f - 4.0f / 3.0f)};
  const float omx = (pi * 2.0f) * (float)L.in.w, omy = (pi * 2.0f) * (float)L.in.h;
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float4 r = sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds);
  const float ax = u * omx, ay = v * omy;
  const float yf = (bs + sin_(ay)) / (bs + 1.0f);
  const float r3[3] = {r.x, r.y, r.z};
  float out[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) out[c] = (yf * ((bl + sin_(ax + off[c])) / (bl + 1.0f))) * r3[c];
  store_rt(L, z, x, y, make_float4(out[0], out[1], out[2], 1.0f), &lds);
  RC_TILE_LOOP_END
}

// scalenx/shaders/epx.glsl (scalenx/epx.glslp: NEAREST, source x 2), FS 97-136: EPX / Scale2x selection rules.
__device__ __forceinline__ bool epx_same(const float4 a, const float4 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
__global__ void __launch_bounds__(256) k_epx(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float tsx = (float)L.in.w, tsy = (float)L.in.h, idx = 1.0f / tsx, idy = 1.0f / tsy;
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const uint8_t* img = frame_ptr(L.in, z);
  const float4 P = sample_rt(L.in, img, u + 0.0f * idx, v + 0.0f * idy, &lds);
  const float4 A = sample_rt(L.in, img, u + 0.0f * idx, v + 1.0f * idy, &lds), B = sample_rt(L.in, img, u + 1.0f * idx, v + 0.0f * idy, &lds);
  const float4 D = sample_rt(L.in, img, u + 0.0f * idx, v + -1.0f * idy, &lds), C = sample_rt(L.in, img, u + -1.0f * idx, v + 0.0f * idy, &lds);
  const float4 one = (epx_same(C, D) && !epx_same(C, A) && !epx_same(C, B)) ? C : P;
  const float4 two = (epx_same(D, B) && !epx_same(D, C) && !epx_same(D, A)) ? D : P;
  const float4 three = (epx_same(A, C) && !epx_same(A, B) && !epx_same(A, D)) ? A : P;
  const float4 four = (epx_same(B, A) && !epx_same(B, D) && !epx_same(B, C)) ? B : P;
  float pxx = u * tsx, pxy = v * tsy;
  pxx = pxx - __builtin_floorf(pxx);
  pxy = pxy - __builtin_floorf(pxy);
  float4 o = pxx < 0.5f ? (pxy < 0.5f ? one : three) : (pxy < 0.5f ? two : four);
  o.w = 1.0f;
  store_rt(L, z, x, y, o, &lds);
  RC_TILE_LOOP_END
}

// interpolation/shaders/quilez.glsl (FS 87-102) and interpolation/shaders/sharp-bilinear.glsl (FS 104-121; params
// SHARP_BILINEAR_PRE_SCALE, AUTO_PRESCALE): a modified coordinate, then one sample with the input's own filter.
// MODE 0: quilez, 1: sharp-bilinear, 2: smootheststep (interpolation/shaders/smootheststep.glsl FS 87-112)
template <int MODE>
__global__ void __launch_bounds__(256) k_interp(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float tsx = (float)L.in.w, tsy = (float)L.in.h, idx = 1.0f / tsx, idy = 1.0f / tsy;
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  float qx, qy;
  if (MODE != 1) {
    const float px = u * tsx + 0.5f, py = v * tsy + 0.5f;
    const float ix = __builtin_floorf(px), iy = __builtin_floorf(py);
    float fx = px - ix, fy = py - iy;
    if (MODE == 2) {
      fx = (((fx * fx) * fx) * fx) * (fx * (fx * (-20.0f * fx + 70.0f) - 84.0f) + 35.0f);
      fy = (((fy * fy) * fy) * fy) * (fy * (fy * (-20.0f * fy + 70.0f) - 84.0f) + 35.0f);
    } else {
      fx = ((fx * fx) * fx) * (fx * (fx * 6.0f - 15.0f) + 10.0f);
      fy = ((fy * fy) * fy) * (fy * (fy * 6.0f - 15.0f) + 10.0f);
    }
    qx = ((ix + fx) - 0.5f) * idx;
    qy = ((iy + fy) - 0.5f) * idy;
  } else {
    const float scale = L.params[1] > 0.5f ? __builtin_floorf((float)L.out_h / tsy + 0.01f) : L.params[0];
    const float range = 0.5f - 0.5f / scale;
    const float tx = u * tsx, ty = v * tsy;
    const float flx = __builtin_floorf(tx), fly = __builtin_floorf(ty);
    const float cdx = (tx - flx) - 0.5f, cdy = (ty - fly) - 0.5f;
    const float clx = fminf(fmaxf(cdx, -range), range), cly = fminf(fmaxf(cdy, -range), range);
    qx = (flx + ((cdx - clx) * scale + 0.5f)) / tsx;
    qy = (fly + ((cdy - cly) * scale + 0.5f)) / tsy;
  }
  float4 c = sample_rt(L.in, frame_ptr(L.in, z), qx, qy, &lds);
  if (MODE != 0) c.w = 1.0f;   // quilez writes vec4(texture), the other two vec4(rgb, 1.0)
  store_rt(L, z, x, y, c, &lds);
  RC_TILE_LOOP_END
}

// crt/shaders/crt-nes-mini.glsl, VS 38-43, FS 94-105; plane[0], plane[1]: TEX0 = TexCoord * 1.00001.
// params: SCANTHICK, INTENSITY, BRIGHTBOOST (the last always 1.25: one of the uniforms the reference overwrites)
__global__ void __launch_bounds__(256) k_crt_nes_mini(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float thick = L.params[0], inten = L.params[1], boost = L.params[2];
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float4 t = sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds);
  const float sy0 = (v * thick) * (float)L.in.h;
  const float sel = sy0 - 2.0f * __builtin_floorf(sy0 / 2.0f);
  const float hi = sel < 1.0f ? 0.0f : 1.0f, lw = 1.0f - hi;
  const float t3[3] = {t.x, t.y, t.z};
  float out[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float ph = ((1.0f + boost) - 0.2f * t3[c]) * t3[c];
    const float pl = ((1.0f - inten) + 0.1f * t3[c]) * t3[c];
    out[c] = lw * pl + hi * ph;
  }
  store_rt(L, z, x, y, make_float4(out[0], out[1], out[2], 1.0f), &lds);
  RC_TILE_LOOP_END
}

// crt/shaders/crt-easymode.glsl (ENABLE_LANCZOS 1), FS 159-268; 17 params in pragma order (oracle/rc_passes_basic.c).
__device__ __forceinline__ float em_curve(float x, float sharp) {
  const float x_step = x < 0.5f ? 0.0f : 1.0f;
  const float h = 0.5f - x;
  const float sg = h > 0.0f ? 1.0f : (h < 0.0f ? -1.0f : 0.0f);
  const float curve = 0.5f - __builtin_sqrtf(0.25f - (x - x_step) * (x - x_step)) * sg;
  return x + sharp * (curve - x);
}
template <bool GENERIC>
__device__ __forceinline__ void em_lanczos(const Tex& t, const uint8_t* img, float u, float v, float dx, const float* k, float dil,
                                            const SrgbLds* lds, float* out3) {
  float m[4][3];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float su = q == 0 ? u - dx : (q == 1 ? u : (q == 2 ? u + dx : u + 2.0f * dx));
    const float sv = q == 0 ? v - 0.0f : (q == 1 ? v : (q == 2 ? v + 0.0f : v + 2.0f * 0.0f));
    const float4 c = GENERIC ? sample_rt(t, img, su, sv, lds) : sample<FMT_RGBX8, 0, WRAP_EDGE>(t, img, su, sv, lds);
    // dilate(): col * mix(1.0, col, DILATION) with a uniform weight: a*(1 - t) + b*t (oracle: in-situ float probe)
    const float om = 1.0f - dil;
    m[q][0] = c.x * (om + c.x * dil);
    m[q][1] = c.y * (om + c.y * dil);
    m[q][2] = c.z * (om + c.z * dil);
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float col = ((m[0][c] * k[0] + m[1][c] * k[1]) + m[2][c] * k[2]) + m[3][c] * k[3];
    const float mn = m[1][c] < m[2][c] ? m[1][c] : m[2][c], mx = m[1][c] > m[2][c] ? m[1][c] : m[2][c];
    const float lo = col > mn ? col : mn;
    out3[c] = lo < mx ? lo : mx;
  }
}
// GENERIC false: GL_RGB source (RGBX8) NEAREST clamp-to-edge and a plain RGBA8 target - the shipped single-pass preset
template <bool GENERIC>
__global__ void __launch_bounds__(256) k_crt_easymode(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float* P = L.params;
  const float sh = P[0], sv = P[1], mstr = P[2], mdw = P[3], mdh = P[4], mstag = P[5], msize = P[6], sstr = P[7];
  const float bwmin = P[8], bwmax = P[9], brmin = P[10], brmax = P[11], cutoff = P[12], gin = P[13], gout = P[14], boost = P[15], dil = P[16];
  const float tsx = (float)L.in.w, tsy = (float)L.in.h, idx = 1.0f / tsx, idy = 1.0f / tsy;
  const float pi = 3.141592653589f;
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float pcx = u * tsx - 0.5f, pcy = v * tsy - 0.5f;
  const float flx = __builtin_floorf(pcx), fly = __builtin_floorf(pcy);
  const float tcx = (flx + 0.5f) * idx, tcy = (fly + 0.5f) * idy;
  const float dsx = pcx - flx, dsy = pcy - fly;
  const float cx = em_curve(dsx, sh * sh);
  float k[4] = {pi * (1.0f + cx), pi * cx, pi * (1.0f - cx), pi * (2.0f - cx)};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float c = __builtin_fabsf(k[q]);
    c = c > 1e-5f ? c : 1e-5f;
    k[q] = ((2.0f * sin_(c)) * sin_(c * 0.5f)) / (c * c);
  }
  const float ksum = k[0] + (k[1] + (k[2] + k[3]));
#pragma unroll
  for (int q = 0; q < 4; ++q) k[q] = k[q] / ksum;
  const uint8_t* img = frame_ptr(L.in, z);
  float c1[3], c2[3], col[3];
  em_lanczos<GENERIC>(L.in, img, tcx, tcy, idx, k, dil, &lds, c1);
  em_lanczos<GENERIC>(L.in, img, tcx + 0.0f, tcy + idy, idx, k, dil, &lds, c2);
  const float cy = em_curve(dsy, sv);
  const float ge = gin / (dil + 1.0f);
#pragma unroll
  for (int c = 0; c < 3; ++c) col[c] = pow_(c1[c] + cy * (c2[c] - c1[c]), ge);
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
  float scan_weight = 1.0f - pow_(cos_(ang) * 0.5f + 0.5f, scan_beam) * sstr;
  const float mask = 1.0f - mstr;
  const float mfx = __builtin_floorf(((u * (float)L.out_w) * tsx) / (tsx * msize));
  const float mfy = __builtin_floorf(((v * (float)L.out_h) * tsy) / (tsy * (mdh * msize)));
  const float m2 = mfy - 2.0f * __builtin_floorf(mfy / 2.0f);
  const float qd = (mfx + m2 * mstag) / mdw;
  const int dot_no = (int)(qd - 3.0f * __builtin_floorf(qd / 3.0f));
  const float mw[3] = {dot_no == 0 ? 1.0f : mask, dot_no == 1 ? 1.0f : mask, (dot_no != 0 && dot_no != 1) ? 1.0f : mask};
  if (tsy >= cutoff) scan_weight = 1.0f;
  float out[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float c0 = col[c] * scan_weight;
    const float r = (c0 + scan_bright * (col[c] - c0)) * mw[c];
    out[c] = pow_(r, 1.0f / gout) * boost;
  }
  if (GENERIC) store_rt(L, z, x, y, make_float4(out[0], out[1], out[2], 1.0f), &lds);
  else store<FMT_RGBA8>(L, z, x, y, make_float4(out[0], out[1], out[2], 1.0f), &lds);
  RC_TILE_LOOP_END
}

}  // namespace

namespace rck {

hipError_t launch_stock(const PassLaunch& L, hipStream_t s) {
  if (L.in.fmt == FMT_RGBA8 && L.in.wrap == WRAP_EDGE && L.out_fmt == FMT_RGBA8 && !(L.flags & RC_FLAG_STOCK_NO_BLIT)) {
    if (L.in.linear) hipLaunchKernelGGL(k_stock_blit<true>, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
    else hipLaunchKernelGGL(k_stock_blit<false>, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_stock, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
hipError_t launch_mix_frames(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL(k_mix_frames, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
#define RC_SIMPLE_LAUNCH(fn, kernel)                                                        \
  hipError_t fn(const PassLaunch& L, hipStream_t s) {                                       \
    hipLaunchKernelGGL(kernel, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);       \
    return hipGetLastError();                                                               \
  }
RC_SIMPLE_LAUNCH(launch_history_size, k_history_size)
RC_SIMPLE_LAUNCH(launch_motionblur_simple, k_motionblur_simple)
RC_SIMPLE_LAUNCH(launch_braid_rewind, k_braid_rewind)
RC_SIMPLE_LAUNCH(launch_response_time, k_response_time)
RC_SIMPLE_LAUNCH(launch_mix_frames_smart, k_mix_frames_smart)
RC_SIMPLE_LAUNCH(launch_color_matrix, k_color_matrix)
RC_SIMPLE_LAUNCH(launch_retro_v2, k_retro_v2)
RC_SIMPLE_LAUNCH(launch_agb001, k_agb001)
RC_SIMPLE_LAUNCH(launch_gb_pass_5, k_gb_pass_5)
RC_SIMPLE_LAUNCH(launch_imgborder, k_imgborder)
RC_SIMPLE_LAUNCH(launch_lut, k_lut)
RC_SIMPLE_LAUNCH(launch_gb_palette, k_gb_palette)
RC_SIMPLE_LAUNCH(launch_crt_potato, k_crt_potato)
RC_SIMPLE_LAUNCH(launch_ntsc_gauss, k_ntsc_gauss)
RC_SIMPLE_LAUNCH(launch_interlacing, k_interlacing)
RC_SIMPLE_LAUNCH(launch_gbc_gambatte_color, k_gbc_gambatte_color)
RC_SIMPLE_LAUNCH(launch_shutter_3d, k_shutter_3d)
RC_SIMPLE_LAUNCH(launch_anti_flicker, k_anti_flicker)
#undef RC_SIMPLE_LAUNCH
hipError_t launch_feedback_persist(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL(k_feedback_persist, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
hipError_t launch_scanline(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL(k_scanline, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
hipError_t launch_lcd1x(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL(k_lcd1x, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
hipError_t launch_bayer(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL(k_bayer, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
hipError_t launch_lcd3x(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL(k_lcd3x, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
hipError_t launch_epx(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL(k_epx, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
hipError_t launch_quilez(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL(k_interp<0>, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
hipError_t launch_sharp_bilinear(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL(k_interp<1>, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
hipError_t launch_smootheststep(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL(k_interp<2>, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
hipError_t launch_crt_nes_mini(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL(k_crt_nes_mini, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
hipError_t launch_crt_easymode(const PassLaunch& L, hipStream_t s) {
  const bool fast = L.in.fmt == FMT_RGBX8 && !L.in.linear && L.in.wrap == WRAP_EDGE && L.in.n_levels <= 1 && L.out_fmt == FMT_RGBA8 &&
                    !(L.flags & RC_FLAG_GENERAL_ONLY);
  if (fast) hipLaunchKernelGGL(k_crt_easymode<false>, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  else hipLaunchKernelGGL(k_crt_easymode<true>, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
hipError_t launch_zfast_crt(const PassLaunch& L, hipStream_t s) {
  const bool fast = L.in.fmt == FMT_RGBX8 && L.in.linear && L.in.wrap == WRAP_EDGE && L.in.n_levels <= 1 && L.out_fmt == FMT_RGBA8 &&
                    !(L.flags & RC_FLAG_GENERAL_ONLY);
  if (fast) hipLaunchKernelGGL(k_zfast_crt<false>, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  else hipLaunchKernelGGL(k_zfast_crt<true>, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
}  // namespace rck
