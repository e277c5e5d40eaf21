// Host-visible entry points of the pass kernels: one launch function per registered
// shader.  A launch renders L.n_frames frames (grid z) of one pass on `stream`.
#pragma once
#include <hip/hip_runtime.h>

#include "rc_device.h"

namespace rck {
using rcd::PassLaunch;
typedef hipError_t (*LaunchFn)(const PassLaunch& L, hipStream_t stream);

hipError_t launch_stock(const PassLaunch& L, hipStream_t s);
hipError_t launch_scanline(const PassLaunch& L, hipStream_t s);
hipError_t launch_crt_pi(const PassLaunch& L, hipStream_t s);
hipError_t launch_zfast_crt(const PassLaunch& L, hipStream_t s);
hipError_t launch_crt_easymode(const PassLaunch& L, hipStream_t s);
hipError_t launch_crt_geom(const PassLaunch& L, hipStream_t s);   // pass_geom.hip
hipError_t launch_crt_nes_mini(const PassLaunch& L, hipStream_t s);
hipError_t launch_quilez(const PassLaunch& L, hipStream_t s);
hipError_t launch_epx(const PassLaunch& L, hipStream_t s);
hipError_t launch_lcd3x(const PassLaunch& L, hipStream_t s);
hipError_t launch_lcd1x(const PassLaunch& L, hipStream_t s);
hipError_t launch_bayer(const PassLaunch& L, hipStream_t s);
hipError_t launch_smootheststep(const PassLaunch& L, hipStream_t s);
hipError_t launch_sharp_bilinear(const PassLaunch& L, hipStream_t s);
hipError_t launch_feedback_persist(const PassLaunch& L, hipStream_t s);
hipError_t launch_history_size(const PassLaunch& L, hipStream_t s);
hipError_t launch_mix_frames(const PassLaunch& L, hipStream_t s);
hipError_t launch_motionblur_simple(const PassLaunch& L, hipStream_t s);
hipError_t launch_braid_rewind(const PassLaunch& L, hipStream_t s);
hipError_t launch_response_time(const PassLaunch& L, hipStream_t s);
hipError_t launch_mix_frames_smart(const PassLaunch& L, hipStream_t s);
hipError_t launch_shutter_3d(const PassLaunch& L, hipStream_t s);
hipError_t launch_color_matrix(const PassLaunch& L, hipStream_t s);
hipError_t launch_retro_v2(const PassLaunch& L, hipStream_t s);
hipError_t launch_agb001(const PassLaunch& L, hipStream_t s);
hipError_t launch_gb_pass_5(const PassLaunch& L, hipStream_t s);
hipError_t launch_imgborder(const PassLaunch& L, hipStream_t s);
hipError_t launch_lut(const PassLaunch& L, hipStream_t s);
hipError_t launch_gb_palette(const PassLaunch& L, hipStream_t s);
hipError_t launch_crt_potato(const PassLaunch& L, hipStream_t s);
hipError_t launch_ntsc_gauss(const PassLaunch& L, hipStream_t s);
hipError_t launch_interlacing(const PassLaunch& L, hipStream_t s);
hipError_t launch_tvout_tweaks(const PassLaunch& L, hipStream_t s);       // pass_lists.hip
hipError_t launch_image_adjustment(const PassLaunch& L, hipStream_t s);
hipError_t launch_jinc2_sharper(const PassLaunch& L, hipStream_t s);
hipError_t launch_crt_lottes(const PassLaunch& L, hipStream_t s);
hipError_t launch_fakelottes(const PassLaunch& L, hipStream_t s);
hipError_t launch_side_by_side(const PassLaunch& L, hipStream_t s);
hipError_t launch_sameboy_lcd(const PassLaunch& L, hipStream_t s);
hipError_t launch_crt_consumer(const PassLaunch& L, hipStream_t s);
hipError_t launch_reverse_aa(const PassLaunch& L, hipStream_t s);
hipError_t launch_advanced_aa(const PassLaunch& L, hipStream_t s);
hipError_t launch_lcd_grid_v2(const PassLaunch& L, hipStream_t s);
hipError_t launch_lcd_grid(const PassLaunch& L, hipStream_t s);     // pass_lcd_grid.hip
hipError_t launch_gbc_gambatte_color(const PassLaunch& L, hipStream_t s);
hipError_t launch_anti_flicker(const PassLaunch& L, hipStream_t s);
hipError_t launch_ntsc_pass1(const PassLaunch& L, hipStream_t s);
hipError_t launch_ntsc_pass2(const PassLaunch& L, hipStream_t s);
hipError_t launch_ntsc_pass1_composite_3phase(const PassLaunch& L, hipStream_t s);
hipError_t launch_ntsc_pass1_svideo_2phase(const PassLaunch& L, hipStream_t s);
hipError_t launch_ntsc_pass1_composite_2phase(const PassLaunch& L, hipStream_t s);
hipError_t launch_ntsc_pass2_3phase_linear(const PassLaunch& L, hipStream_t s);
hipError_t launch_ntsc_pass2_3phase_plain(const PassLaunch& L, hipStream_t s);
hipError_t launch_ntsc_pass2_2phase_gamma(const PassLaunch& L, hipStream_t s);
hipError_t launch_ntsc_pass2_2phase_linear(const PassLaunch& L, hipStream_t s);
hipError_t launch_ntsc_pass2_2phase_plain(const PassLaunch& L, hipStream_t s);
hipError_t launch_xbr_lv3(const PassLaunch& L, hipStream_t s);
hipError_t launch_xbr_lv2(const PassLaunch& L, hipStream_t s);
hipError_t launch_scalefx0(const PassLaunch& L, hipStream_t s);
hipError_t launch_scalefx1(const PassLaunch& L, hipStream_t s);
hipError_t launch_scalefx2(const PassLaunch& L, hipStream_t s);
hipError_t launch_scalefx3(const PassLaunch& L, hipStream_t s);
hipError_t launch_scalefx4(const PassLaunch& L, hipStream_t s);
hipError_t launch_glow_linearize(const PassLaunch& L, hipStream_t s);
hipError_t launch_glow_threshold(const PassLaunch& L, hipStream_t s);
hipError_t launch_glow_blur_h(const PassLaunch& L, hipStream_t s);
hipError_t launch_glow_blur_v(const PassLaunch& L, hipStream_t s);
hipError_t launch_crt_hyllian_glow(const PassLaunch& L, hipStream_t s);
hipError_t launch_hyllian_resolve2(const PassLaunch& L, hipStream_t s);
hipError_t launch_royale_first(const PassLaunch& L, hipStream_t s);
bool royale_first_byte_map(const PassLaunch& L, hipStream_t s, float* d_dec256);   // KernelEntry::byte_map
hipError_t launch_royale_scan_v(const PassLaunch& L, hipStream_t s);
hipError_t launch_royale_bloom_approx(const PassLaunch& L, hipStream_t s);
hipError_t launch_blur9(const PassLaunch& L, hipStream_t s);
bool launch_blur9_tile(const PassLaunch& L, hipStream_t s, hipError_t* err);   // pass_royale_blur.hip; false: not this geometry
hipError_t launch_royale_mask_v(const PassLaunch& L, hipStream_t s);
hipError_t launch_royale_mask_h(const PassLaunch& L, hipStream_t s);
hipError_t launch_royale_scan_h(const PassLaunch& L, hipStream_t s);
hipError_t launch_royale_scan_h_fake(const PassLaunch& L, hipStream_t s);
hipError_t launch_royale_brightpass(const PassLaunch& L, hipStream_t s);
hipError_t launch_royale_bloom_v(const PassLaunch& L, hipStream_t s);
hipError_t launch_royale_bloom_h(const PassLaunch& L, hipStream_t s);
hipError_t launch_royale_last(const PassLaunch& L, hipStream_t s);
hipError_t launch_royale_last_general(const PassLaunch& L, hipStream_t s);   // pass_royale_last_general.hip

// frame_io.hip: pixel-format conversion either side of the chain (fmt: 0 RGB24, 1 BGRA, 2 RGBA, 3 YUYV422)
hipError_t launch_ingest(const void* src, int fmt, uint32_t w, uint32_t h, uint32_t n, void* dst_rgba8, hipStream_t s);
hipError_t launch_egress_rgb24(const void* src_rgba8, uint32_t w, uint32_t h, uint32_t n, int flip_y, void* dst, hipStream_t s);
hipError_t launch_selftest(unsigned long long* d_counts, hipStream_t s);
hipError_t launch_selftest_copy(const void* d_src, void* d_dst, size_t bytes, hipStream_t s);   // a 16-byte-per-lane streaming copy
// pass_royale_scan.hip: host-built expansion tables of the crt-royale scanline pass (tests)
void royale_scan_tables_host(float off, float* A, uint32_t* B);
int royale_scan_table_nodes();
hipError_t royale_scan_tables_device(float off, const float* dists, int n_dists, float* A, float* bound, hipStream_t s);
// pass_royale.hip: the last pass's output-gamma table for 1 / lcd_gamma (royale_common.h), cached per device; nullptr if it cannot be built
const float4* royale_last_gamma_table(float inv_gamma, hipStream_t s);
hipError_t launch_selftest_srgb8(const float* d_src, uint8_t* d_dst, size_t n, const uint32_t* table, hipStream_t s, int form = 1);

// 64x4 pixel tiles: one wave per row segment, so each wave stores 256 contiguous bytes of an
// RGBA8 row.  A workgroup walks tiles grid-stride (tile index = frame, tile row, tile column),
// which amortises its prologue (sRGB tables into LDS) over many tiles; the grid is capped at
// 8 workgroups per CU on a 256-CU part.
inline dim3 px_block() { return dim3(64, 4, 1); }
inline dim3 px_grid(const PassLaunch& L) {
  const long tiles = (long)((L.out_w + 63) / 64) * ((L.out_h + 3) / 4) * L.n_frames;
  return dim3((unsigned)(tiles < 2048 ? (tiles > 0 ? tiles : 1) : 2048), 1, 1);
}
}  // namespace rck
