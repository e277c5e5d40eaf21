/* rc_shaderchain.h - C ABI of the MI355X shader-chain engine (librcshaderchain.so).
 *
 * Drop-in boundary for RetroCapture's per-frame RetroArch-GLSL shader chain: every entry
 * point replaces one member of the reference's C++ class `ShaderEngine`
 * (reference src/shader/ShaderEngine.h:42-98; implementation src/shader/ShaderEngine.cpp).
 * Plain pointers and sizes only; frames are DEVICE pointers to 8-bit RGBA texels, row 0
 * first (= texture coordinate t 0, the reference's upload order, FrameProcessor.cpp:172-205),
 * alpha ignored on input (the reference's source texture is GL_RGB) and written by the last
 * pass on output.  Output frames are owned by the engine and stay valid until the next
 * apply / load / destroy on the same engine (reference ShaderEngine.cpp:1873).
 *
 * Error convention, as in the reference: calls return a status / handle and describe the
 * failure in the log (stderr, level from RETROCAPTURE_LOG_LEVEL); rc_last_error() returns the
 * last error text of the calling thread.  No C++ exception crosses this boundary.
 *
 * Threading, as in the reference: one thread drives an engine (create, load, viewport,
 * apply); rc_engine_param_* and rc_engine_preset_path may additionally be called from
 * other threads (the reference's HTTP worker does, APIController.cpp:1046,1340,1759) and
 * are internally locked here.
 */
#ifndef RC_SHADERCHAIN_H
#define RC_SHADERCHAIN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rc_engine rc_engine;

enum {
  RC_OK = 0,
  RC_ERR_INVALID = -1,     /* bad handle / argument                                        */
  RC_ERR_LOAD = -2,        /* preset / shader could not be parsed or opened                */
  RC_ERR_DEVICE = -3,      /* HIP failure (allocation, launch)                             */
  RC_WARN_PASSES = 1       /* preset loaded, but >= 1 pass has no kernel and is skipped,
                              exactly like a pass whose GLSL failed to compile in the
                              reference (ShaderEngine.cpp:968-975)                          */
};

/* ShaderEngine::ShaderEngine + init() (ShaderEngine.cpp:22-60).  device < 0 keeps the
 * calling thread's current HIP device; hip_stream is a hipStream_t (NULL = default
 * stream) on which every kernel of this engine is enqueued.  NULL on failure (no HIP
 * device: there is no CPU fallback). */
rc_engine* rc_engine_create(int device, void* hip_stream);
/* shutdown() + destructor (ShaderEngine.cpp:62-86). */
void rc_engine_destroy(rc_engine* e);

/* loadPreset(path) (ShaderEngine.cpp:228-319): parses the .glslp, loads LUT PNGs, reads
 * each pass's #pragma parameters and resolves its HIP kernel.  Clears custom parameters. */
int rc_engine_load_preset(rc_engine* e, const char* glslp_path);
/* loadShader(path) (ShaderEngine.cpp:149-226): one .glsl file as a one-pass chain. */
int rc_engine_load_shader(rc_engine* e, const char* glsl_path);
/* getPresetPath() (ShaderEngine.h:56).  Copies up to cap-1 bytes; returns full length. */
size_t rc_engine_preset_path(rc_engine* e, char* buf, size_t cap);
/* disableShader() / isShaderActive() (ShaderEngine.h:65-67). */
void rc_engine_disable(rc_engine* e);
int rc_engine_is_active(rc_engine* e);

/* setViewport(w,h) (ShaderEngine.cpp:3154-3206): must precede apply for presets. */
void rc_engine_set_viewport(rc_engine* e, uint32_t width, uint32_t height);
/* setMaxShaderResolution(w,h) (ShaderEngine.h:74-76); 0,0 = off (default). */
void rc_engine_set_max_resolution(rc_engine* e, uint32_t max_width, uint32_t max_height);

/* applyShader(texture, w, h) (ShaderEngine.cpp:1531-1879).  d_in: device pointer to
 * width*height RGBA8 texels.  *d_out receives the device pointer of the output frame:
 * the engine's buffer, or d_in itself when no shader is active / no pass is usable (the
 * reference returns the input texture in those cases, :1533, :1606-1619).  Advances the
 * engine's FrameCount by one (:1688).  Work is enqueued on the engine's stream and NOT
 * synchronised. */
int rc_engine_apply(rc_engine* e, const void* d_in, uint32_t width, uint32_t height,
                    const void** d_out, uint32_t* out_width, uint32_t* out_height);
/* N independent frames in one call: frame k (0-based) is processed with the FrameCount
 * that the k-th of N successive rc_engine_apply calls would see.  frame_stride = bytes
 * between input frames (0 = width*height*4).  *d_out: N output frames, tightly packed. */
int rc_engine_apply_batch(rc_engine* e, const void* d_in, uint32_t n_frames, uint32_t width,
                          uint32_t height, uint64_t frame_stride, const void** d_out,
                          uint32_t* out_width, uint32_t* out_height);
/* getOutputWidth()/getOutputHeight() (ShaderEngine.h:70-71): size of the last output. */
uint32_t rc_engine_output_width(rc_engine* e);
uint32_t rc_engine_output_height(rc_engine* e);
/* hipStreamSynchronize on the engine's stream. */
int rc_engine_sync(rc_engine* e);

/* getShaderParameters() (ShaderEngine.cpp:3264-3351): union of all passes' #pragma
 * parameters, sorted by name; value = custom > preset > default. */
typedef struct {
  char name[64];
  char description[128];
  float value, default_value, min, max, step;
} rc_param;
int rc_engine_param_count(rc_engine* e);
int rc_engine_param_get(rc_engine* e, int index, rc_param* out);
/* setShaderParameter(name, v) (ShaderEngine.cpp:3353-3387): clamps to [min,max];
 * returns 1 if the parameter exists, 0 otherwise. */
int rc_engine_param_set(rc_engine* e, const char* name, float value);
/* setUniform(name, ...) x3 (ShaderEngine.cpp:3006-3041): recorded only; as in the
 * reference they do not reach preset passes. */
void rc_engine_set_uniform1(rc_engine* e, const char* name, float x);
void rc_engine_set_uniform2(rc_engine* e, const char* name, float x, float y);
void rc_engine_set_uniform4(rc_engine* e, const char* name, float x, float y, float z, float w);
/* ShaderPreset::saveAs via getPreset() (ShaderPreset.cpp:557-661) with the engine's custom
 * parameters, as the reference's UI does (UICallbackWiring.cpp:172-175). */
int rc_engine_save_preset(rc_engine* e, const char* path);

/* ---- inspection (no reference counterpart; used by tests and tools) -------------------- */
typedef struct {
  uint32_t width, height;   /* render-target size at the last apply                       */
  int format;               /* 0 RGBA8, 1 sRGB8_ALPHA8, 3 RGBA32F, 4 four binary16 (fp16 option) */
  int has_kernel;           /* 0: pass is skipped                                          */
  int filter_linear;        /* the pass's filter_linear / wrap (state set on its INPUT)    */
  int wrap;                 /* 0 edge, 1 border, 2 repeat, 3 mirrored_repeat               */
  char kernel[48];
  char alias[48];
} rc_pass_info;
int rc_engine_pass_count(rc_engine* e);
int rc_engine_pass_info(rc_engine* e, int pass, rc_pass_info* out);
/* Copies pass `pass`'s render target of frame `frame` (index within the last batch; an
 * intermediate pass only holds the last chunk) to host memory.  Synchronises. */
int rc_engine_read_pass(rc_engine* e, int pass, uint32_t frame, void* host, size_t bytes);
/* Per-pass device timing (HIP events on the engine's stream around every pass launch) and
 * the pass's algorithmic bytes per frame (each distinct sampled texture once at its stored
 * size + the target once).  rc_engine_pass_profile synchronises the stream. */
typedef struct {
  double total_ms;
  uint64_t launches;
  uint64_t frames;
  uint64_t read_bytes_per_frame;
  uint64_t write_bytes_per_frame;
  uint32_t folded;     /* 1: the pass was folded into its consumers for the last chunk (rc_engine_set_fold_passes): no launch, no bytes moved */
  uint32_t reserved;
} rc_pass_profile;
void rc_engine_set_profiling(rc_engine* e, int on);
int rc_engine_pass_profile(rc_engine* e, int pass, rc_pass_profile* out);
/* Frames processed per kernel launch through the whole chain (default: 128 where 128 frames of the chain's largest pass
 * target stay below 2 GiB, else 64). */
void rc_engine_set_chunk_frames(rc_engine* e, uint32_t n);
/* Not in the reference (its one GL context draws one frame at a time): with n = 2 - the default since round 4 - the second half
 * of every batch is rendered on a second HIP stream of the same device (a helper engine instance inside `e`: same preset,
 * parameters and flags) straight into the batch output; the helper's stream is ordered after the engine's stream at the start
 * of the call and before it at the end, so callers still see one stream-ordered result and the same bytes, and the kernels of
 * the two halves overlap (a pass that waits on memory under one that waits on arithmetic: +4 ... 10 % on the 1080p chains).
 * Presets that sample frame history or PassFeedback, single-shader mode, single-frame calls and profiled runs
 * (rc_engine_set_profiling: a kernel's own duration is a one-lane figure) use one lane.  rc_engine_read_pass finds an
 * intermediate pass of the second half in the helper; device memory for intermediates doubles.  n = 1: one lane. */
void rc_engine_set_lanes(rc_engine* e, uint32_t n);
/* 1: a pass whose .glsl file is unreadable still runs if its shader identity is registered
 * (built-in parameter table).  Default 0 = the reference's behaviour (pass fails). */
void rc_engine_set_allow_missing_sources(rc_engine* e, int allow);

/* crt-royale's pass 6 (mask-resize-horizontal.glsl FS ~3377) tests a varying that its vertex
 * shader never writes (VS 3288-3328 shadows it with a local).  0 (default): every fragment of
 * that pass is discarded, which is what Mesa llvmpipe - the GL the parity vectors come from -
 * does; 1: the varying reads 0, as GL drivers that zero undefined varyings behave, and the
 * resized phosphor mask is rendered. */
void rc_engine_set_undefined_varying_zero(rc_engine* e, int zero);

/* Frame history (reference ShaderEngine.h:140-143, .cpp:1735-1865): presets whose first pass samples
 * PrevTexture / Prev<N>Texture keep a ring of at most 7 RGBA8 frames, newest first; such presets
 * process the frames of a batch one after the other.  rc_engine_history_count returns the ring's
 * length; rc_engine_read_history copies entry k to host memory (host may be NULL to query the size). */
int rc_engine_history_count(rc_engine* e);
int rc_engine_read_history(rc_engine* e, int k, uint32_t* width, uint32_t* height, void* host, size_t bytes);

/* Some passes have a specialised form next to their general one (e.g. xbr-lv3 evaluates its edge
 * rules once per source pixel when the sampling pattern allows it).  Both forms give identical
 * results; 1 forces the general form (diagnostics / tests).  Default 0. */
void rc_engine_set_general_kernels_only(rc_engine* e, int general_only);
/* A pass whose target is, byte for byte, a per-channel map of its input's bytes (crt-royale's first pass at 1:1: gamma 2.5
 * through an sRGB8 target) is folded into its consumers: it is not rendered, the passes that sample its target read its input
 * through the composed decode table - the same values, so every other pass's bytes are unchanged (tests).  Default 1.
 * rc_engine_read_pass of a folded pass renders it on demand from the input frames of the last apply call, which the
 * caller must still hold; rc_pass_profile::folded tells.  0 renders every pass. */
void rc_engine_set_fold_passes(rc_engine* e, int on);
/* Some specialised forms need per-(geometry, parameter) tables that take a while to build - crt-royale's scanline pass proves
 * its tables' error bounds by exhaustion, 140 ms at 1080p; crt-pi's two pow tables likewise per gamma setting.  By default (1)
 * such a build runs on a worker thread with its own HIP stream and the pass renders with its general form - the same bytes -
 * until the tables are ready: no apply call waits for it (the reference's callers move parameters from a UI slider and from
 * HTTP threads).  0: the first frame of a new configuration waits for its tables (deterministic form selection: tests, benchmarks). */
void rc_engine_set_async_table_builds(rc_engine* e, int on);
/* float_framebuffer render targets (the reference creates GL_RGBA32F, ShaderEngine.cpp:2872-2923) stored as
 * four binary16 values per texel instead: 8 bytes instead of 16 (ntsc-256px-svideo at 1080p: 45.9 -> 28.2 MB of
 * algorithmic bytes per frame).  Every pass still computes in float; a store to such a target rounds to nearest
 * even, a fetch widens exactly - the chain equals the fp32 chain with those targets rounded to binary16, which
 * the tests check bit for bit, and differs from the fp32 (bit-exact, default) path by at most 1 step of the final
 * 8-bit output on the ntsc presets (asserted in tests/test_fp16_targets.py).  rc_pass_info.format reports 4
 * (RC_FMT_F16) for such a target and rc_engine_read_pass returns its 8-byte texels.  Default 0. */
void rc_engine_set_float_target_fp16(rc_engine* e, int on);

/* ---- frame ingest / egress (device buffers; stream may be NULL) ------------------------------
 * rc_ingest replaces the pixel-format handling of FrameProcessor::processFrame (reference
 * src/processing/FrameProcessor.cpp:43-222: RGB24 upload, BGRA/RGBA swizzle, YUYV422 through
 * libswscale) and produces the tightly packed RGBA8 frames rc_engine_apply takes (row 0 first,
 * alpha 255).  rc_egress_rgb24 replaces the readback's alpha strip (reference
 * src/core/FrameCapturePipeline.cpp:1060-1080): RGBA8 -> tightly packed RGB24, rows optionally
 * reversed.  n frames are contiguous on both sides.  YUYV422 needs an even width. */
typedef enum { RC_PIX_RGB24 = 0, RC_PIX_BGRA = 1, RC_PIX_RGBA = 2, RC_PIX_YUYV422 = 3 } rc_pixfmt;
int rc_ingest(const void* d_src, int pixfmt, uint32_t width, uint32_t height, uint32_t n_frames, void* d_rgba8, void* stream);
int rc_egress_rgb24(const void* d_rgba8, uint32_t width, uint32_t height, uint32_t n_frames, int flip_y, void* d_rgb24,
                    void* stream);
/* bytes per frame of a pixel format (0 for an unknown one) */
size_t rc_pixfmt_frame_bytes(int pixfmt, uint32_t width, uint32_t height);

/* ---- OpenGLRenderer::renderTexture, off-screen ------------------------------------------------
 * The reference draws one textured quad with a fixed program (reference
 * src/renderer/OpenGLRenderer.cpp:378-470; program :141-158: rgb = (t.rgb * brightness - 0.5) * contrast
 * + 0.5, alpha = t.a) in three places of the frame path; rc_present is that draw on device buffers:
 *   pre-pass  NEAREST downscale to the logical capture size and overscan crop of the captured frame
 *             into a GL_RGB target, through the viewport rc_overscan_viewport computes
 *             (src/core/FrameCapturePipeline.cpp:160-250)          src_rgb 1, src_linear 0, RC_PRESENT_RGBX8
 *   resize    shader output -> configured output resolution (:413-505)   src_linear 1, RC_PRESENT_RGBA8
 *   bake      brightness / contrast into the frame for stream / recording (:739-804), same size
 * `bake` != 0 runs a second draw (bake_brightness, bake_contrast) on the first one's RGBA8 result at the
 * same size inside the same kernel - resize followed by bake as the reference chains them - and
 * RC_PRESENT_RGB24 with out_flip_rows fuses the readback's alpha strip and row flip (:1060-1080).
 * Source frames are RGBA8-sized (4 bytes per pixel; rc_ingest output or an engine output), n frames
 * contiguous on both sides.  Target pixels outside the viewport receive `clear` (the reference clears
 * to 0,0,0,0 before the resize and to 0,0,0,1 before the bake; its viewports always cover the target). */
typedef enum { RC_PRESENT_RGBA8 = 0, RC_PRESENT_RGBX8 = 1, RC_PRESENT_RGB24 = 2 } rc_present_kind;
typedef struct {
  uint32_t src_w, src_h;
  int src_rgb;               /* 1: GL_RGB source texture (alpha samples as 1.0) */
  int src_linear;            /* the source texture's filter: 1 GL_LINEAR, 0 GL_NEAREST */
  uint32_t dst_w, dst_h;
  int dst_kind;              /* rc_present_kind */
  int32_t vp_x, vp_y, vp_w, vp_h; /* glViewport; vp_w == 0: the whole target */
  int flip_y;                /* the program's flipY uniform */
  float brightness, contrast;
  float clear[4];
  int bake;
  float bake_brightness, bake_contrast;
  int out_flip_rows;         /* store rows bottom-up */
} rc_present_desc;
int rc_present(const void* d_src, void* d_dst, const rc_present_desc* desc, uint32_t n_frames, void* stream);
/* bytes per frame of an rc_present target */
size_t rc_present_frame_bytes(int dst_kind, uint32_t width, uint32_t height);
/* glViewport of the pre-pass for an overscan crop of pct percent per side (clamped to 45):
 * FrameCapturePipeline.cpp:205-216.  vp = x, y, w, h. */
void rc_overscan_viewport(uint32_t fbo_w, uint32_t fbo_h, float pct_x, float pct_y, int32_t vp[4]);

/* ---- host-to-host frame pipeline -----------------------------------------------------------
 * The reference's per-frame path between a capture buffer and the encoder's RGB24 buffer:
 * FrameProcessor upload (FrameProcessor.cpp:43-222) -> applyShader -> readback through double-
 * buffered PBOs (PBOManager.cpp:86-170, FrameCapturePipeline.cpp:974-1084).  Here: a ring of `slots`
 * (2..8) on three HIP streams (copy in / engine stream / copy out) linked by events, so the PCIe
 * copies of neighbouring frames overlap the kernels.  submit() copies the caller's (unpinned) frame
 * into pinned staging and queues H2D + rc_ingest + chain + rc_egress_rgb24 + D2H; receive() hands out
 * the oldest finished frame (RGB24, row 0 first) in pipeline-owned pinned memory.  Ownership rule: a
 * received frame is the caller's until the NEXT rc_pipeline_receive (or rc_pipeline_destroy); its slot
 * is not reused before that, so N slots carry at most N - 1 frames in flight besides the one the caller
 * holds.  Frames return in submission order.  Lifetime: a pipeline may outlive its engine - when the
 * engine is destroyed (or shut down) the pipeline drains its streams and every later submit / receive /
 * input_buffer call fails; destroy the pipeline afterwards as usual. */
typedef struct rc_pipeline rc_pipeline;
rc_pipeline* rc_pipeline_create(rc_engine* e, int slots);
void rc_pipeline_destroy(rc_pipeline* p);
/* RC_OK, RC_ERR_INVALID (bad arguments or no free slot: receive first) or RC_ERR_DEVICE */
int rc_pipeline_submit(rc_pipeline* p, const void* host_frame, int pixfmt, uint32_t width, uint32_t height);
/* RC_OK and *host_rgb24 / *width / *height; 1 if wait == 0 and the frame is not finished; < 0 on error
 * or when nothing is in flight */
int rc_pipeline_receive(rc_pipeline* p, const void** host_rgb24, uint32_t* width, uint32_t* height, int wait);
/* Pinned staging memory of the slot the next submit will use (NULL if no slot is free): capture into
 * it and pass the same pointer to rc_pipeline_submit to skip the host-side copy. */
void* rc_pipeline_input_buffer(rc_pipeline* p, int pixfmt, uint32_t width, uint32_t height);
int rc_pipeline_in_flight(rc_pipeline* p);
void rc_pipeline_set_flip_y(rc_pipeline* p, int flip_y);
/* The frame path's optional stages either side of the chain, as FrameCapturePipeline applies them:
 * source pre-pass (logical capture size smaller than the captured frame in both dimensions => NEAREST
 * downscale; overscan percent per side > 0.001 => crop; FrameCapturePipeline.cpp:160-250; 0 = off),
 * output resolution (:413-505; 0 = off) and image adjustments (:739-804; 1.0 = off).  They apply to
 * frames submitted afterwards. */
void rc_pipeline_set_source_prepass(rc_pipeline* p, uint32_t logical_w, uint32_t logical_h, float overscan_pct_x, float overscan_pct_y);
void rc_pipeline_set_output_resolution(rc_pipeline* p, uint32_t width, uint32_t height);
void rc_pipeline_set_image_adjust(rc_pipeline* p, float brightness, float contrast);

/* Device self-test: the division shortcuts the kernels use (log2's mantissa division, the
 * safe-range division, constant divisors) against IEEE division on the device's own reciprocal
 * instruction: all 2^23 mantissas / 2^26 operand pairs / 6 x 2^24 quotients.  mismatches[0..2]
 * receive the counts (all 0 on a conforming device).  Returns RC_OK or RC_ERR_DEVICE. */
int rc_selftest_fastmath(int device, uint64_t mismatches[3]);
/* The device's streaming rate as a kernel reaches it: `reps` grid-stride copies of `bytes` bytes (a multiple of 16), 16 bytes
 * per lane in and out, timed with HIP events on the null stream; *gb_per_s = read + written bytes per second / 1e9.  The
 * ceiling the byte-moving passes are measured against beside the 8 TB/s vendor figure (bench.py). */
int rc_selftest_copy_rate(int device, size_t bytes, int reps, double* gb_per_s);
/* The sRGB8 encode of an sRGB render target (GL_FRAMEBUFFER_SRGB, reference ShaderEngine.cpp:944-952; the
 * conversion is the GL's: Mesa llvmpipe's RSQRTPS-based lp_build_linear_to_srgb, restated exactly in
 * csrc/srgb_encode.cpp) applied to a flat array of floats: with the host-side per-run table the kernels
 * use (host pointers), and by the device function every pass kernel stores through (device pointers).
 * For tests: both must equal the oracle's byte for every float. */
int rc_selftest_srgb8_host(const float* src, uint8_t* dst, size_t n);
int rc_selftest_srgb8_device(int device, const float* d_src, uint8_t* d_dst, size_t n, void* stream);
/* The same through the second form of the table (form = 2: one entry per run from the first float that stores a non-zero
 * byte, the linear segment included; the strip kernels' encode - clamp, one LDS read, one add); form = 1 is the above (on the
 * device: the table read where it lies in device memory, as small targets do); form = 3, device only, is form 1 with the
 * table copied into every workgroup's LDS first, as every large sRGB8 target does. */
int rc_selftest_srgb8_host_form(const float* src, uint8_t* dst, size_t n, int form);
int rc_selftest_srgb8_device_form(int device, const float* d_src, uint8_t* d_dst, size_t n, void* stream, int form);
/* crt-royale's scanline pass (crt-royale-scanlines-vertical-interlacing.glsl) runs, at 1:1 geometry, from an
 * expansion table of the beam function around its possible colours with a certified remainder bound
 * (csrc/kernels/pass_royale_scan.hip).  This returns the host-built part of that table for sub-pixel offset
 * `off` so that tests can check the bound against exact evaluations: A = [9][nodes][4] floats (0, dK/dc,
 * d2K/dc2 / 2, dK/ddist), B = [9][nodes][2] words (0, node colour as float bits: the bounds are measured on the device, below), index
 * 9 = scanline * 3 + channel.  Returns the node count (also with null pointers), negative on error. */
int rc_selftest_royale_scan_tables(float off, float* A, uint32_t* B, size_t a_floats, size_t b_words);
/* The complete tables as the device builds them for sub-pixel offset `off` and the row distances dists[0 .. n_dists) of a
 * geometry: A = [9][nodes][4] floats WITH the node values (T, dK/dc, d2K/dc2 / 2, dK/ddist), bound = [9][nodes] floats - for
 * every node the largest difference between the exact beam function and the expansion over EVERY float colour the node is
 * selected for and every given distance, measured exhaustively on the device (see pass_royale_scan.hip).  Needs a GPU.
 * Returns the node count, negative on error. */
int rc_selftest_royale_scan_bounds(int device, float off, const float* dists, int n_dists, float* A, float* bound, size_t a_floats, size_t bound_floats);

/* crt/crt-geom.glslp: what the shader's VERTEX stage hands every pixel (it depends on uniforms only), evaluated on the
 * host exactly as the engine does per launch (csrc/kernels/geom_math.h): params = the 17 #pragma parameters in
 * declaration order, out = sinangle.xy, cosangle.xy, stretch.xyz.  For tests against the oracle; no GPU needed. */
int rc_selftest_crt_geom_vertex(const float* params, float* out);

const char* rc_last_error(void);
const char* rc_version(void);
/* Names of the registered kernels ("identity\n" list) for diagnostics. */
size_t rc_kernel_list(char* buf, size_t cap);

/* ShaderPreset::saveAs without an engine (reference ShaderPreset.cpp:557-661): loads `preset_path` and writes it
 * to `out_path` line by line, replacing the value of every line whose key is one of the preset's own global
 * parameters or one of the n custom (name, value) pairs - a parameter without a line in the file gets none, as
 * in the reference.  RC_OK, RC_ERR_LOAD or RC_ERR_INVALID. */
int rc_preset_save_as(const char* preset_path, const char* out_path, const char* const* names, const float* values, int n);

/* Standalone preset parser check (ShaderPreset::load, ShaderPreset.cpp:18-333): writes a JSON
 * description of the parsed preset; returns the length needed.  No GPU needed. */
size_t rc_preset_dump_json(const char* glslp_path, char* buf, size_t cap);
/* LUT decode as loadTextureReference does it (ShaderEngine.cpp:2535-2706): PNG -> RGBA8, row 0
 * first.  Returns 0 and the size, or <0 (buffer too small / not a PNG).  No GPU needed. */
int rc_png_decode_rgba8(const char* png_path, void* rgba, size_t cap, int* width, int* height);
/* #pragma parameter scan of one shader file (ShaderPreprocessor.cpp:30-79) as JSON. */
size_t rc_shader_params_json(const char* glsl_path, char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* RC_SHADERCHAIN_H */
