/* TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT.
 *
 * CPU restatement (plain C) of the reference's per-frame shader chain: what
 * ShaderEngine::applyShader (reference src/shader/ShaderEngine.cpp:1531-1879) makes the GL
 * driver compute when it draws each pass of a .glslp preset, with the arithmetic of the
 * GLSL shader assets (reference shaders/shaders_glsl/...) written out by hand per pass.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  The product (retrocapture_amd/) never links it.
 *
 * Pinning: the reference ShaderEngine itself cannot be built here without a stand-in for
 * GLFW, so engine-level parity is pinned by (a) the reference's own preset parser compiled
 * unmodified (oracle/_ref/dump_preset) and (b) golden vectors produced by executing the
 * reference's GLSL files on Mesa llvmpipe 23.2.1 through oracle/glrun (tests/golden/).
 * The float primitives below (pow/exp2/log2/sin/cos, texel decode, filtering, varying
 * interpolation, UNORM8 store) reproduce llvmpipe's results bit-for-bit as measured by
 * oracle/probes/; the sRGB8 encode is a monotone threshold table (llvmpipe's own encode goes
 * through the x86 RSQRTPS approximation and is not monotone).
 */
#ifndef RC_ORACLE_H
#define RC_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z, w; } o_vec4;

/* ---- float primitives (rc_math.c) ---------------------------------------------------- */
float o_exp2(float x);
float o_log2(float x);
float o_pow(float x, float y);
float o_exp(float x);
float o_log(float x);
float o_sin(float x);
float o_cos(float x);
unsigned o_fp_enter(void);      /* set FTZ|DAZ, returns the previous MXCSR */
void o_fp_leave(unsigned old);

/* ---- textures / sampling (rc_sampler.c) ---------------------------------------------- */
enum { O_FMT_RGBA8 = 0, O_FMT_SRGB8 = 1, O_FMT_RGBX8 = 2, O_FMT_F32 = 3 };
enum { O_WRAP_EDGE = 0, O_WRAP_BORDER = 1, O_WRAP_REPEAT = 2, O_WRAP_MIRROR = 3 };

typedef struct {
  const void* data; /* row 0 = t 0; RGBA8/SRGB8/RGBX8: 4 bytes per texel; F32: 4 floats */
  int w, h;
  int fmt;
  int linear; /* filter: 1 = GL_LINEAR, 0 = GL_NEAREST */
  int wrap;
  /* mipmap_input (ShaderEngine.cpp:1022-1033: GL_LINEAR_MIPMAP_LINEAR + glGenerateMipmap): levels 1.. of the
   * chain, level k is max(1, w >> k) x max(1, h >> k); n_levels = 0 or 1: not mip-mapped */
  int n_levels;
  const void* mip[15]; /* mip[k] = level k (mip[0] = data) */
} o_tex;
#define O_MAX_LEVELS 15
/* number of levels of a full chain, and llvmpipe's glGenerateMipmap: every level is a LINEAR blit of the
 * one above (sRGB8: decoded, averaged and re-encoded; F32: plain); dst[k] must hold level k (k >= 1) */
int o_mip_levels(int w, int h);
void o_gen_mipmaps(const void* level0, int w, int h, int fmt, void* const* dst, int n_levels);
/* texture() on a mip-mapped texture: the coordinate at this pixel and at its horizontal / vertical
 * neighbour inside the 2x2 quad (llvmpipe takes per-pixel differences, rc_sampler.c) */
o_vec4 o_sample_quad(const o_tex* t, float s, float v, float s_dx0, float s_dx1, float v_dx0, float v_dx1,
                     float s_dy0, float s_dy1, float v_dy0, float v_dy1);
float o_lod_from_quad(const o_tex* t, float s_dx0, float s_dx1, float v_dx0, float v_dx1,
                      float s_dy0, float s_dy1, float v_dy0, float v_dy1);

o_vec4 o_texel(const o_tex* t, int x, int y); /* decoded texel, no wrap */
o_vec4 o_sample(const o_tex* t, float s, float v);
extern const float o_srgb_decode_table[256];
uint8_t o_store_unorm8(float x);
uint8_t o_store_srgb8(float x);
/* bulk form for the exhaustive probe and the tests: dst[i] = o_store_srgb8(src[i]) */
void o_store_srgb8_array(const float* src, uint8_t* dst, size_t n);
/* crt-royale pass 1: one scanline's contribution to one channel, on arrays (rc_passes_royale.c) */
void o_royale_beam_array(const float* dist, const float* color, float ph, float* out, size_t n);

/* ---- varyings (rc_varying.c) --------------------------------------------------------- */
/* A varying written by the vertex shader, as the rasteriser hands it to pixel (x, y) of a
 * W x H target: plane-equation setup per triangle of the quad (BL,BR,TR)+(TR,TL,BL). */
typedef struct { float a0_lo, dx_lo, dy_lo, a0_up, dx_up, dy_up; } o_varying;
o_varying o_varying_setup(float a_bl, float a_br, float a_tr, float a_tl, int W, int H, int out_fmt);
o_varying o_varying_setup_fan(float a_bl, float a_br, float a_tr, float a_tl, int W, int H, int out_fmt);   /* the GL's own blit quad */
static inline int o_lower_tri(int x, int y, int W, int H) {
  return ((double)y + 0.5) * (double)W <= ((double)x + 0.5) * (double)H;
}
float o_varying_at(const o_varying* v, int x, int y, int lower);

/* ---- passes ---------------------------------------------------------------------------- */
/* Every pass renders a full out_w x out_h target from `in` (the previous pass's output with
 * the sampler state the pass's preset entry sets on it).  dst receives stored texels
 * (4 bytes, or 4 floats for O_FMT_F32).  `frame_count` is the value of the FrameCount
 * uniform (first applied frame = 1).  params: the pass's #pragma parameter values in
 * declaration order. */
typedef struct {
  const o_tex* in;       /* "Texture" sampler */
  const o_tex* extra[8]; /* PassPrev / alias / LUT samplers, pass specific */
  int src_w, src_h;      /* OriginalSize: size of the frame handed to applyShader */
  int out_w, out_h;
  int out_fmt;           /* O_FMT_RGBA8 / SRGB8 / F32 */
  int frame_count;
  const float* params;
  void* dst;
  int y0, y1;            /* row range to render (for threading); 0,out_h for all */
  /* chain geometry, for shaders that read PassPrev<N>TextureSize / InputSize uniforms
   * (reference ShaderEngine.cpp:1191-1227): output size of every pass, and this pass's index */
  int pass_index;
  int n_passes;
  int chain_w[16], chain_h[16];
  int vp_w, vp_h;
  int flags;             /* pass specific, see O_FLAG_* */
  /* The size uniforms as the shader reads them, where they are NOT the real sizes of `in` and the target: the reference's
   * history push re-draws the final output through pass 0's program with the uniforms pass 0's own draw left behind
   * (ShaderEngine.cpp:1805-1834).  TextureSize == InputSize = uni_tex, OutputSize = uni_out; 0 = the real sizes.
   * Read by the passes that declare it (chain_specs "stale_size_uniforms"). */
  int uni_tex_w, uni_tex_h, uni_out_w, uni_out_h;
} o_pass_args;

/* crt-royale pass 6 reads a varying its vertex shader never writes.  On Mesa llvmpipe the
 * undefined value makes the fragment shader discard every pixel (default, flags = 0); GL
 * drivers that hand back 0 for such a varying render the resized mask instead. */
#define O_FLAG_ROYALE_UNDEF_VARYING_ZERO 1
/* o_pass_stock: take the ordinary sampler even where llvmpipe's blit fast path would apply to a plain draw
 * (glGenerateMipmap's per-level blits of an RGBA8 texture do, measured: rc_sampler.c o_gen_mipmaps) */
#define O_FLAG_STOCK_NO_BLIT 2

/* scalefx/scalefx.glslp (rc_passes_scalefx.c): pass 1 params SFX_CLR, SFX_SAA; pass 2 extra[0] = PassPrev2Texture;
 * pass 3 param SFX_SCN; pass 4 extra[0] = PassPrev5Texture (the original frame) */
void o_pass_scalefx0(const o_pass_args* a);
void o_pass_scalefx1(const o_pass_args* a);
void o_pass_scalefx2(const o_pass_args* a);
void o_pass_scalefx3(const o_pass_args* a);
void o_pass_scalefx4(const o_pass_args* a);
void o_pass_stock(const o_pass_args* a);
void o_pass_scanline(const o_pass_args* a);
void o_pass_crt_pi(const o_pass_args* a);
void o_pass_crt_easymode(const o_pass_args* a);        /* 17 params */
void o_pass_crt_geom(const o_pass_args* a);            /* 17 params; reads FrameCount (rc_passes_geom.c) */
float o_acos(float x);
void o_crt_geom_vertex(const float* params, float* out); /* sinangle.xy, cosangle.xy, stretch.xyz */
void o_pass_crt_nes_mini(const o_pass_args* a);        /* 3 params */
void o_pass_bayer(const o_pass_args* a);               /* 2 params; reads FrameCount */
void o_pass_epx(const o_pass_args* a);
void o_pass_lcd1x(const o_pass_args* a);               /* 2 params */
void o_pass_lcd3x(const o_pass_args* a);               /* 2 params */
void o_pass_quilez(const o_pass_args* a);
void o_pass_smootheststep(const o_pass_args* a);
void o_pass_sharp_bilinear(const o_pass_args* a);      /* 2 params */
void o_pass_zfast_crt(const o_pass_args* a);           /* 6 params (always the reference's fixed values) */
void o_pass_feedback_persist(const o_pass_args* a);   /* fixture; extra = PassFeedback0, PassFeedback1; 1 param */
void o_pass_history_size(const o_pass_args* a);      /* fixture; extra = PrevTexture, Prev1Texture; 1 param; reads uni_* */
void o_pass_mix_frames(const o_pass_args* a);         /* extra[0] = PrevTexture (frame history) */
void o_pass_motionblur_simple(const o_pass_args* a);  /* extra[0..6] = Prev6 .. Prev1, PrevTexture */
void o_pass_braid_rewind(const o_pass_args* a);       /* history declared, not used (FrameDirection = 1) */
void o_pass_response_time(const o_pass_args* a);      /* 1 param; extra[0..6] = PrevTexture, Prev1 .. Prev6 */
void o_pass_crt_lottes(const o_pass_args* a);         /* 13 params (rc_passes_lists.c) */
void o_pass_fakelottes(const o_pass_args* a);         /* 10 params (rc_passes_lists.c) */
void o_pass_advanced_aa(const o_pass_args* a);        /* 2 params (rc_passes_lists.c) */
void o_pass_reverse_aa(const o_pass_args* a);         /* 1 param (rc_passes_lists.c) */
void o_pass_crt_consumer(const o_pass_args* a);       /* 33 params (rc_passes_lists.c) */
void o_pass_sameboy_lcd(const o_pass_args* a);        /* 3 params (rc_passes_lists.c) */
void o_pass_side_by_side(const o_pass_args* a);       /* 9 params (rc_passes_lists.c) */
void o_pass_jinc2_sharper(const o_pass_args* a);      /* no params (rc_passes_lists.c) */
void o_pass_tvout_tweaks(const o_pass_args* a);       /* 6 params (rc_passes_lists.c) */
void o_pass_image_adjustment(const o_pass_args* a);   /* 23 params (rc_passes_lists.c) */
void o_pass_ntsc_gauss(const o_pass_args* a);         /* 2 params (rc_passes_ntsc_xbr.c) */
void o_pass_interlacing(const o_pass_args* a);        /* 3 params */
void o_pass_crt_potato(const o_pass_args* a);         /* no params; extra[0] = MASK */
void o_pass_gb_palette(const o_pass_args* a);         /* no params; extra[0] = COLOR_PALETTE */
void o_pass_lut(const o_pass_args* a);                /* 1 param; extra[0] = SamplerLUT */
void o_pass_console_border(const o_pass_args* a);     /* 8 params; extra[0] = BORDER */
void o_pass_imgborder(const o_pass_args* a);          /* 12 params; extra[0] = BORDER */
void o_pass_gb_pass_5(const o_pass_args* a);          /* 3 params; extra[0] = BORDER */
void o_pass_agb001(const o_pass_args* a);             /* no params */
void o_pass_retro_v2(const o_pass_args* a);           /* 1 param */
void o_pass_lcd_grid(const o_pass_args* a);           /* 2 params (rc_passes_lcd.c) */
void o_pass_lcd_grid_v2(const o_pass_args* a);        /* 15 params (rc_passes_lcd.c) */
void o_pass_gba_color(const o_pass_args* a);          /* handheld/shaders/color/: 1 param (gba, gbc, vba) or none */
void o_pass_gbc_color(const o_pass_args* a);
void o_pass_vba_color(const o_pass_args* a);
void o_pass_nds_color(const o_pass_args* a);
void o_pass_palm_color(const o_pass_args* a);
void o_pass_psp_color(const o_pass_args* a);
void o_pass_gbc_gambatte_color(const o_pass_args* a);
void o_pass_shutter_3d(const o_pass_args* a);         /* 7 params; extra[0] = PrevTexture */
void o_pass_anti_flicker(const o_pass_args* a);       /* 1 param; extra[0] = PrevTexture, extra[1] = Prev1Texture */
void o_pass_mix_frames_smart(const o_pass_args* a);   /* 1 param; extra[0..4] = PrevTexture, Prev1 .. Prev4 */
/* crt-royale (shaders/shaders_glsl/crt/shaders/crt-royale/src/ and blurs/blur9fast-{vertical,horizontal}.glsl) */
void o_pass_royale_first(const o_pass_args* a);       /* P0  first-pass-linearize-crt-gamma-bob-fields */
void o_pass_royale_scan_v(const o_pass_args* a);      /* P1  scanlines-vertical-interlacing */
void o_pass_royale_bloom_approx(const o_pass_args* a);/* P2  bloom-approx; extra[0] = PassPrev2Texture */
void o_pass_blur9_v(const o_pass_args* a);            /* P3  blurs/blur9fast-vertical */
void o_pass_blur9_h(const o_pass_args* a);            /* P4  blurs/blur9fast-horizontal */
void o_pass_royale_mask_v(const o_pass_args* a);      /* P5  mask-resize-vertical; extra[0] = mask_slot_texture_small */
void o_pass_royale_mask_h(const o_pass_args* a);      /* P6  mask-resize-horizontal */
void o_pass_royale_scan_h(const o_pass_args* a);      /* P7  scanlines-horizontal-apply-mask; extra = PassPrev6, PassPrev3 */
void o_pass_royale_scan_h_fake(const o_pass_args* a); /* P7 of crt-royale-fake-bloom; extra = PassPrev6, PassPrev5, PassPrev3 */
void o_pass_royale_brightpass(const o_pass_args* a);  /* P8  brightpass; extra[0] = PassPrev4 */
void o_pass_royale_bloom_v(const o_pass_args* a);     /* P9  bloom-vertical */
void o_pass_royale_bloom_h(const o_pass_args* a);     /* P10 bloom-horizontal-reconstitute; extra = PassPrev3, PassPrev2, PassPrev6 */
int o_royale_last_is_general(const float* params);       /* tex2Daa / curved geometry selected (rc_passes_royale_last.c) */
void o_pass_royale_last_general(const o_pass_args* a);
void o_pass_royale_last(const o_pass_args* a);        /* P11 geometry-aa-last-pass; 44 params */
/* ntsc/ntsc-256px-svideo.glslp (2 passes) and xbr/xbr-lv3.glslp (1 pass) */
/* the ntsc family: ntsc-pass1-{svideo,composite}-{2,3}phase.glsl, ntsc-pass2-{2,3}phase{,-gamma,-linear}.glsl */
void o_pass_ntsc_pass1_svideo_3phase(const o_pass_args* a);
void o_pass_ntsc_pass1_composite_3phase(const o_pass_args* a);
void o_pass_ntsc_pass1_svideo_2phase(const o_pass_args* a);
void o_pass_ntsc_pass1_composite_2phase(const o_pass_args* a);
void o_pass_ntsc_pass2_3phase_gamma(const o_pass_args* a);
void o_pass_ntsc_pass2_3phase_linear(const o_pass_args* a);
void o_pass_ntsc_pass2_3phase(const o_pass_args* a);
void o_pass_ntsc_pass2_2phase_gamma(const o_pass_args* a);
void o_pass_ntsc_pass2_2phase_linear(const o_pass_args* a);
void o_pass_ntsc_pass2_2phase(const o_pass_args* a);
void o_pass_xbr_lv3(const o_pass_args* a);            /* 5 params */
void o_pass_xbr_lv2(const o_pass_args* a);            /* 5 params (small_details < 0.5 only) */
/* crt/crt-hyllian-glow.glslp (rc_passes_glow.c) */
void o_pass_glow_linearize(const o_pass_args* a);     /* 1 param */
void o_pass_crt_hyllian_glow(const o_pass_args* a);   /* 10 params */
void o_pass_glow_threshold(const o_pass_args* a);     /* 2 params */
void o_pass_glow_blur_h(const o_pass_args* a);        /* mip-mapped input */
void o_pass_glow_blur_v(const o_pass_args* a);
void o_pass_hyllian_resolve2(const o_pass_args* a);   /* 4 params; extra[0] = PassPrev4Texture */
void o_store_pixel(const o_pass_args* a, int x, int y, o_vec4 c);

/* ---- OpenGLRenderer::renderTexture off-screen (rc_present.c) ------------------------------ */
typedef struct {
  const o_tex* src;      /* RGBX8 (captured frame, GL_RGB) or RGBA8 (shader output); wrap = edge */
  int dst_w, dst_h;
  int dst_fmt;           /* O_FMT_RGBX8 (GL_RGB target), O_FMT_RGBA8, O_FMT_F32 */
  int vp_x, vp_y, vp_w, vp_h;
  int flip_y;
  float brightness, contrast;
  o_vec4 clear;          /* what target pixels outside the viewport hold */
  void* dst;
} o_present_args;
void o_present(const o_present_args* a);
/* FrameCapturePipeline.cpp:205-216: the pre-pass viewport for an overscan crop, percent per side */
void o_overscan_viewport(int fbo_w, int fbo_h, float pct_x, float pct_y, int vp[4]);

#ifdef __cplusplus
}
#endif
#endif
