// extern "C" shim over rc::ShaderEngine - see include/rc_shaderchain.h for the contract.
#include "../../include/rc_shaderchain.h"

#include <cstdio>
#include <cstring>
#include <sstream>

#include "png_lut.h"
#include "rc_log.h"
#include "frame_pipeline.h"
#include "kernels/geom_math.h"
#include "present_setup.h"
#include "shader_engine.h"
#include "srgb_encode.h"

struct rc_engine {
  rc::ShaderEngine impl;
};

namespace {

size_t copy_out(const std::string& s, char* buf, size_t cap) {
  if (buf && cap) {
    size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    std::memcpy(buf, s.data(), n);
    buf[n] = 0;
  }
  return s.size();
}

std::string jstr(const std::string& s) {
  std::string o = "\"";
  for (char c : s) {
    if (c == '"' || c == '\\') o += '\\';
    if ((unsigned char)c < 0x20) {
      char b[8];
      std::snprintf(b, sizeof b, "\\u%04x", c);
      o += b;
      continue;
    }
    o += c;
  }
  return o + "\"";
}
std::string jnum(float v) {
  char b[32];
  std::snprintf(b, sizeof b, "%.9g", v);
  return b;
}

template <typename F>
int guarded(F&& f) {
  try {
    return f();
  } catch (const std::exception& ex) {
    RC_LOG_ERROR(std::string("exception: ") + ex.what());
    return RC_ERR_INVALID;
  } catch (...) {
    RC_LOG_ERROR("unknown exception");
    return RC_ERR_INVALID;
  }
}

}  // namespace

extern "C" {

rc_engine* rc_engine_create(int device, void* hip_stream) {
  rc_engine* e = nullptr;
  try {
    e = new rc_engine();
    if (!e->impl.init(device, static_cast<hipStream_t>(hip_stream))) {
      delete e;
      return nullptr;
    }
  } catch (...) {
    delete e;
    return nullptr;
  }
  return e;
}

void rc_engine_destroy(rc_engine* e) {
  try {
    delete e;
  } catch (...) {
  }
}

static int load_status(rc_engine* e, bool ok) {
  if (!ok) return RC_ERR_LOAD;
  for (size_t i = 0; i < e->impl.passCount(); ++i)
    if (!e->impl.pass(i)->kernel) return RC_WARN_PASSES;
  return RC_OK;
}

int rc_engine_load_preset(rc_engine* e, const char* path) {
  if (!e || !path) return RC_ERR_INVALID;
  return guarded([&] { return load_status(e, e->impl.loadPreset(path)); });
}

int rc_engine_load_shader(rc_engine* e, const char* path) {
  if (!e || !path) return RC_ERR_INVALID;
  return guarded([&] { return load_status(e, e->impl.loadShader(path)); });
}

size_t rc_engine_preset_path(rc_engine* e, char* buf, size_t cap) {
  if (!e) return 0;
  return copy_out(e->impl.getPresetPath(), buf, cap);
}

void rc_engine_disable(rc_engine* e) {
  if (e) e->impl.disableShader();
}
int rc_engine_is_active(rc_engine* e) { return e && e->impl.isShaderActive() ? 1 : 0; }

void rc_engine_set_viewport(rc_engine* e, uint32_t w, uint32_t h) {
  if (e) e->impl.setViewport(w, h);
}
void rc_engine_set_max_resolution(rc_engine* e, uint32_t w, uint32_t h) {
  if (e) e->impl.setMaxShaderResolution(w, h);
}

int rc_engine_apply_batch(rc_engine* e, const void* d_in, uint32_t n, uint32_t w, uint32_t h, uint64_t stride,
                          const void** d_out, uint32_t* ow, uint32_t* oh) {
  if (!e || !d_out) return RC_ERR_INVALID;
  return guarded([&] {
    rc::clear_last_error();  // the status below must reflect THIS call only
    const void* out = e->impl.applyShaderBatch(d_in, n, w, h, stride);
    *d_out = out;
    if (!out) return (int)RC_ERR_INVALID;
    const bool passthrough = (out == d_in);
    if (ow) *ow = passthrough ? w : e->impl.getOutputWidth();
    if (oh) *oh = passthrough ? h : e->impl.getOutputHeight();
    if (passthrough && e->impl.isShaderActive() && !rc::last_error().empty()) {
      // active engine that fell back to the input: a device failure or no usable pass
      bool any = false;
      for (size_t i = 0; i < e->impl.passCount(); ++i) any |= e->impl.pass(i)->kernel != nullptr;
      if (any) return (int)RC_ERR_DEVICE;
    }
    return (int)RC_OK;
  });
}

int rc_engine_apply(rc_engine* e, const void* d_in, uint32_t w, uint32_t h, const void** d_out, uint32_t* ow,
                    uint32_t* oh) {
  return rc_engine_apply_batch(e, d_in, 1, w, h, 0, d_out, ow, oh);
}

uint32_t rc_engine_output_width(rc_engine* e) { return e ? e->impl.getOutputWidth() : 0; }
uint32_t rc_engine_output_height(rc_engine* e) { return e ? e->impl.getOutputHeight() : 0; }

int rc_engine_sync(rc_engine* e) {
  if (!e) return RC_ERR_INVALID;
  return hipStreamSynchronize(e->impl.stream()) == hipSuccess ? RC_OK : RC_ERR_DEVICE;
}

int rc_engine_param_count(rc_engine* e) {
  if (!e) return 0;
  return guarded([&] { return (int)e->impl.getShaderParameters().size(); });
}

int rc_engine_param_get(rc_engine* e, int index, rc_param* out) {
  if (!e || !out || index < 0) return RC_ERR_INVALID;
  return guarded([&] {
    auto ps = e->impl.getShaderParameters();
    if ((size_t)index >= ps.size()) return (int)RC_ERR_INVALID;
    const auto& p = ps[(size_t)index];
    std::memset(out, 0, sizeof(*out));
    std::snprintf(out->name, sizeof(out->name), "%s", p.name.c_str());
    std::snprintf(out->description, sizeof(out->description), "%s", p.description.c_str());
    out->value = p.value;
    out->default_value = p.defaultValue;
    out->min = p.min;
    out->max = p.max;
    out->step = p.step;
    return (int)RC_OK;
  });
}

int rc_engine_param_set(rc_engine* e, const char* name, float value) {
  if (!e || !name) return 0;
  return guarded([&] { return e->impl.setShaderParameter(name, value) ? 1 : 0; });
}

void rc_engine_set_uniform1(rc_engine* e, const char* n, float x) {
  if (e && n) e->impl.setUniform(n, x);
}
void rc_engine_set_uniform2(rc_engine* e, const char* n, float x, float y) {
  if (e && n) e->impl.setUniform(n, x, y);
}
void rc_engine_set_uniform4(rc_engine* e, const char* n, float x, float y, float z, float w) {
  if (e && n) e->impl.setUniform(n, x, y, z, w);
}

int rc_engine_save_preset(rc_engine* e, const char* path) {
  if (!e || !path) return RC_ERR_INVALID;
  return guarded([&] {
    std::unordered_map<std::string, float> custom;
    for (const auto& p : e->impl.getShaderParameters())
      if (p.value != p.defaultValue) custom[p.name] = p.value;
    return e->impl.getPreset().saveAs(path, custom) ? (int)RC_OK : (int)RC_ERR_LOAD;
  });
}

int rc_engine_pass_count(rc_engine* e) { return e ? (int)e->impl.passCount() : 0; }

int rc_engine_pass_info(rc_engine* e, int pass, rc_pass_info* out) {
  if (!e || !out || pass < 0) return RC_ERR_INVALID;
  const rc::ShaderPassData* pd = e->impl.pass((size_t)pass);
  if (!pd) return RC_ERR_INVALID;
  std::memset(out, 0, sizeof(*out));
  out->width = pd->width;
  out->height = pd->height;
  out->format = pd->format;
  out->has_kernel = pd->kernel ? 1 : 0;
  out->filter_linear = pd->passInfo.filterLinear ? 1 : 0;
  const std::string& w = pd->passInfo.wrapMode;
  out->wrap = w == "repeat" ? 2 : w == "mirrored_repeat" ? 3 : w == "clamp_to_border" ? 1 : 0;
  std::snprintf(out->kernel, sizeof(out->kernel), "%s", pd->kernel ? pd->kernel->name : "");
  std::snprintf(out->alias, sizeof(out->alias), "%s", pd->passInfo.alias.c_str());
  return RC_OK;
}

int rc_engine_read_pass(rc_engine* e, int pass, uint32_t frame, void* host, size_t bytes) {
  if (!e || pass < 0) return RC_ERR_INVALID;
  return guarded([&] { return e->impl.readPass((size_t)pass, frame, host, bytes) ? (int)RC_OK : (int)RC_ERR_INVALID; });
}

void rc_engine_set_profiling(rc_engine* e, int on) {
  if (e) e->impl.setProfiling(on != 0);
}

int rc_engine_pass_profile(rc_engine* e, int pass, rc_pass_profile* out) {
  if (!e || !out || pass < 0) return RC_ERR_INVALID;
  return guarded([&] {
    std::vector<rc::ShaderEngine::PassProfile> prof;
    if (!e->impl.collectProfile(&prof)) return (int)RC_ERR_DEVICE;
    if ((size_t)pass >= prof.size()) return (int)RC_ERR_INVALID;
    std::memset(out, 0, sizeof(*out));
    out->total_ms = prof[(size_t)pass].totalMs;
    out->launches = prof[(size_t)pass].launches;
    out->frames = prof[(size_t)pass].frames;
    e->impl.passBytes((size_t)pass, &out->read_bytes_per_frame, &out->write_bytes_per_frame);
    out->folded = e->impl.passFolded((size_t)pass) ? 1u : 0u;
    return (int)RC_OK;
  });
}

void rc_engine_set_chunk_frames(rc_engine* e, uint32_t n) {
  if (e) e->impl.setChunkFrames(n);
}
void rc_engine_set_lanes(rc_engine* e, uint32_t n) {
  if (e) e->impl.setLanes(n);
}
void rc_engine_set_undefined_varying_zero(rc_engine* e, int zero) {
  if (e) e->impl.setUndefinedVaryingZero(zero != 0);
}
int rc_ingest(const void* d_src, int pixfmt, uint32_t width, uint32_t height, uint32_t n_frames, void* d_rgba8, void* stream) {
  if (!d_src || !d_rgba8 || pixfmt < 0 || pixfmt > 3 || (pixfmt == RC_PIX_YUYV422 && (width & 1u))) return RC_ERR_INVALID;
  return rck::launch_ingest(d_src, pixfmt, width, height, n_frames, d_rgba8, static_cast<hipStream_t>(stream)) == hipSuccess
             ? RC_OK : RC_ERR_DEVICE;
}
int rc_egress_rgb24(const void* d_rgba8, uint32_t width, uint32_t height, uint32_t n_frames, int flip_y, void* d_rgb24,
                    void* stream) {
  if (!d_rgba8 || !d_rgb24) return RC_ERR_INVALID;
  return rck::launch_egress_rgb24(d_rgba8, width, height, n_frames, flip_y, d_rgb24, static_cast<hipStream_t>(stream)) == hipSuccess
             ? RC_OK : RC_ERR_DEVICE;
}
int rc_present(const void* d_src, void* d_dst, const rc_present_desc* desc, uint32_t n_frames, void* stream) {
  if (!d_src || !d_dst || !desc) return RC_ERR_INVALID;
  rc::PresentDesc d;
  d.srcW = desc->src_w;
  d.srcH = desc->src_h;
  d.srcRgb = desc->src_rgb != 0;
  d.srcLinear = desc->src_linear != 0;
  d.dstW = desc->dst_w;
  d.dstH = desc->dst_h;
  d.dstKind = desc->dst_kind;
  d.vpX = desc->vp_x;
  d.vpY = desc->vp_y;
  d.vpW = desc->vp_w;
  d.vpH = desc->vp_h;
  d.flipY = desc->flip_y != 0;
  d.brightness = desc->brightness;
  d.contrast = desc->contrast;
  for (int k = 0; k < 4; ++k) d.clear[k] = desc->clear[k];
  d.bake = desc->bake != 0;
  d.bakeBrightness = desc->bake_brightness;
  d.bakeContrast = desc->bake_contrast;
  d.outFlipRows = desc->out_flip_rows != 0;
  rck::PresentLaunch L;
  if (!rc::makePresentLaunch(d, d_src, d_dst, n_frames, &L)) return RC_ERR_INVALID;
  return rck::launch_present(L, static_cast<hipStream_t>(stream)) == hipSuccess ? RC_OK : RC_ERR_DEVICE;
}
size_t rc_present_frame_bytes(int dst_kind, uint32_t width, uint32_t height) {
  if (dst_kind < 0 || dst_kind > 2) return 0;
  return (size_t)width * height * (dst_kind == RC_PRESENT_RGB24 ? 3 : 4);
}
void rc_overscan_viewport(uint32_t fbo_w, uint32_t fbo_h, float pct_x, float pct_y, int32_t vp[4]) {
  int v[4];
  rc::overscanViewport(fbo_w, fbo_h, pct_x, pct_y, v);
  for (int k = 0; k < 4; ++k) vp[k] = v[k];
}
size_t rc_pixfmt_frame_bytes(int pixfmt, uint32_t width, uint32_t height) {
  const size_t px = (size_t)width * height;
  switch (pixfmt) {
    case RC_PIX_RGB24: return px * 3;
    case RC_PIX_BGRA:
    case RC_PIX_RGBA: return px * 4;
    case RC_PIX_YUYV422: return px * 2;
    default: return 0;
  }
}
struct rc_pipeline {
  rc::FramePipeline impl;
  rc_pipeline(rc::ShaderEngine* e, int slots) : impl(e, slots) {}
};
rc_pipeline* rc_pipeline_create(rc_engine* e, int slots) {
  if (!e) return nullptr;
  rc_pipeline* p = new rc_pipeline(&e->impl, slots);
  if (!p->impl.ok()) {
    delete p;
    return nullptr;
  }
  return p;
}
void rc_pipeline_destroy(rc_pipeline* p) { delete p; }
int rc_pipeline_submit(rc_pipeline* p, const void* host_frame, int pixfmt, uint32_t width, uint32_t height) {
  if (!p || !host_frame) return RC_ERR_INVALID;
  return p->impl.submit(host_frame, pixfmt, width, height) ? RC_OK : RC_ERR_INVALID;
}
int rc_pipeline_receive(rc_pipeline* p, const void** host_rgb24, uint32_t* width, uint32_t* height, int wait) {
  if (!p || p->impl.inFlight() == 0) return RC_ERR_INVALID;
  if (p->impl.receive(host_rgb24, width, height, wait != 0)) return RC_OK;
  return wait ? RC_ERR_DEVICE : 1;
}
void* rc_pipeline_input_buffer(rc_pipeline* p, int pixfmt, uint32_t width, uint32_t height) {
  return p ? p->impl.inputBuffer(pixfmt, width, height) : nullptr;
}
int rc_pipeline_in_flight(rc_pipeline* p) { return p ? p->impl.inFlight() : 0; }
void rc_pipeline_set_flip_y(rc_pipeline* p, int flip_y) {
  if (p) p->impl.setFlipY(flip_y != 0);
}
void rc_pipeline_set_source_prepass(rc_pipeline* p, uint32_t logical_w, uint32_t logical_h, float overscan_pct_x, float overscan_pct_y) {
  if (p) p->impl.setSourcePrepass(logical_w, logical_h, overscan_pct_x, overscan_pct_y);
}
void rc_pipeline_set_output_resolution(rc_pipeline* p, uint32_t width, uint32_t height) {
  if (p) p->impl.setOutputResolution(width, height);
}
void rc_pipeline_set_image_adjust(rc_pipeline* p, float brightness, float contrast) {
  if (p) p->impl.setImageAdjust(brightness, contrast);
}
int rc_selftest_copy_rate(int device, size_t bytes, int reps, double* gb_per_s) {
  if (!gb_per_s || bytes < 16 || (bytes & 15) || reps < 1) return RC_ERR_INVALID;
  return guarded([&] {
    if (hipSetDevice(device) != hipSuccess) return (int)RC_ERR_DEVICE;
    void *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = hipMalloc(&a, bytes) == hipSuccess && hipMalloc(&b, bytes) == hipSuccess && hipMemset(a, 7, bytes) == hipSuccess &&
              hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    for (int i = 0; ok && i < 3; ++i) ok = rck::launch_selftest_copy(a, b, bytes, nullptr) == hipSuccess;
    if (ok) ok = hipEventRecord(e0, nullptr) == hipSuccess;
    for (int i = 0; ok && i < reps; ++i) ok = rck::launch_selftest_copy(a, b, bytes, nullptr) == hipSuccess;
    float ms = 0.f;
    if (ok) ok = hipEventRecord(e1, nullptr) == hipSuccess && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f;
    if (ok) *gb_per_s = 2.0 * (double)bytes * reps / ((double)ms * 1e-3) / 1e9;
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    return ok ? (int)RC_OK : (int)RC_ERR_DEVICE;
  });
}

int rc_selftest_fastmath(int device, uint64_t mismatches[3]) {
  if (!mismatches) return RC_ERR_INVALID;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return RC_ERR_DEVICE;
  unsigned long long* d = nullptr;
  if (hipMalloc(&d, 3 * sizeof(unsigned long long)) != hipSuccess) return RC_ERR_DEVICE;
  int rc = RC_OK;
  unsigned long long h[3] = {0, 0, 0};
  if (hipMemset(d, 0, sizeof(h)) != hipSuccess || rck::launch_selftest(d, nullptr) != hipSuccess ||
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess)
    rc = RC_ERR_DEVICE;
  (void)hipFree(d);
  for (int i = 0; i < 3; ++i) mismatches[i] = h[i];
  return rc;
}
int rc_selftest_srgb8_host(const float* src, uint8_t* dst, size_t n) {
  if (!src || !dst) return RC_ERR_INVALID;
  return guarded([&] {
    std::vector<uint32_t> table;
    if (!rc::buildSrgbRunTable(&table)) return (int)RC_ERR_INVALID;
    for (size_t i = 0; i < n; ++i) dst[i] = rc::srgb8EncodeByRunTable(src[i], table.data());
    return (int)RC_OK;
  });
}
int rc_selftest_srgb8_host_form(const float* src, uint8_t* dst, size_t n, int form) {
  if (form != 2) return rc_selftest_srgb8_host(src, dst, n);
  if (!src || !dst) return RC_ERR_INVALID;
  return guarded([&] {
    std::vector<uint32_t> table;
    if (!rc::buildSrgbRunTable2(&table)) return (int)RC_ERR_INVALID;
    for (size_t i = 0; i < n; ++i) dst[i] = rc::srgb8EncodeByRunTable2(src[i], table.data());
    return (int)RC_OK;
  });
}
int rc_selftest_srgb8_device_form(int device, const float* d_src, uint8_t* d_dst, size_t n, void* stream, int form) {
  if (form != 2 && form != 3) return rc_selftest_srgb8_device(device, d_src, d_dst, n, stream);
  if (!d_src || !d_dst) return RC_ERR_INVALID;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return RC_ERR_DEVICE;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return RC_ERR_DEVICE;
  const uint32_t* table = rc::deviceSrgbRunTable(dev);
  if (!table) return RC_ERR_DEVICE;
  return rck::launch_selftest_srgb8(d_src, d_dst, n, table, static_cast<hipStream_t>(stream), form) == hipSuccess ? RC_OK : RC_ERR_DEVICE;
}
int rc_selftest_srgb8_device(int device, const float* d_src, uint8_t* d_dst, size_t n, void* stream) {
  if (!d_src || !d_dst) return RC_ERR_INVALID;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return RC_ERR_DEVICE;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return RC_ERR_DEVICE;
  const uint32_t* table = rc::deviceSrgbRunTable(dev);
  if (!table) return RC_ERR_DEVICE;
  return rck::launch_selftest_srgb8(d_src, d_dst, n, table, static_cast<hipStream_t>(stream)) == hipSuccess ? RC_OK : RC_ERR_DEVICE;
}
int rc_selftest_royale_scan_tables(float off, float* A, uint32_t* B, size_t a_floats, size_t b_words) {
  const size_t n = (size_t)rck::royale_scan_table_nodes();
  if (!A || !B) return (int)n;
  if (a_floats < 9 * n * 4 || b_words < 9 * n * 2) return RC_ERR_INVALID;
  return guarded([&] {
    rck::royale_scan_tables_host(off, A, B);
    return (int)n;
  });
}
int rc_selftest_royale_scan_bounds(int device, float off, const float* dists, int n_dists, float* A, float* bound, size_t a_floats, size_t bound_floats) {
  const size_t n = (size_t)rck::royale_scan_table_nodes();
  if (!dists || !A || !bound || a_floats < 9 * n * 4 || bound_floats < 9 * n) return RC_ERR_INVALID;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return RC_ERR_DEVICE;
  return guarded([&] { return rck::royale_scan_tables_device(off, dists, n_dists, A, bound, nullptr) == hipSuccess ? (int)n : (int)RC_ERR_DEVICE; });
}
int rc_selftest_crt_geom_vertex(const float* params, float* out) {
  if (!params || !out) return RC_ERR_INVALID;
  float P[rcd::kMaxParams] = {};
  for (int k = 0; k < 17; ++k) P[k] = params[k];
  rcgeom::vertex_constants(P);
  for (int k = 0; k < 7; ++k) out[k] = P[rcgeom::GP_SIN_X + k];
  return RC_OK;
}
int rc_engine_history_count(rc_engine* e) { return e ? (int)e->impl.historyCount() : 0; }
int rc_engine_read_history(rc_engine* e, int k, uint32_t* width, uint32_t* height, void* host, size_t bytes) {
  if (!e || k < 0) return RC_ERR_INVALID;
  return e->impl.readHistory((size_t)k, width, height, host, bytes) ? RC_OK : RC_ERR_INVALID;
}
void rc_engine_set_general_kernels_only(rc_engine* e, int general_only) {
  if (e) e->impl.setGeneralKernelsOnly(general_only != 0);
}
void rc_engine_set_async_table_builds(rc_engine* e, int on) {
  if (e) e->impl.setAsyncTableBuilds(on != 0);
}
void rc_engine_set_fold_passes(rc_engine* e, int on) {
  if (e) e->impl.setFoldPasses(on != 0);
}
void rc_engine_set_float_target_fp16(rc_engine* e, int on) {
  if (e) e->impl.setFloatTargetFp16(on != 0);
}
void rc_engine_set_allow_missing_sources(rc_engine* e, int allow) {
  if (e) e->impl.setAllowMissingSources(allow != 0);
}

const char* rc_last_error(void) { return rc::last_error().c_str(); }
const char* rc_version(void) { return "retrocapture_amd shaderchain 0.1 (gfx950)"; }

size_t rc_kernel_list(char* buf, size_t cap) {
  std::string s;
  for (const auto& k : rc::allKernels()) s += std::string(k.identity) + "\n";
  return copy_out(s, buf, cap);
}

size_t rc_preset_dump_json(const char* path, char* buf, size_t cap) {
  if (!path) return 0;
  std::string out;
  try {
    rc::ShaderPreset p;
    rc::MissingSourceScope missing(true);   // a dump is about the preset text: absent shader files are counted, not logged
    const bool ok = p.load(path);
    std::ostringstream o;
    o << "{\"preset\":" << jstr(path) << ",\"ok\":" << (ok ? "true" : "false") << ",\"missing_files\":" << missing.count() << ",\"passes\":[";
    bool first = true;
    for (const auto& s : p.getPasses()) {
      o << (first ? "" : ",") << "{\"shader\":" << jstr(s.shaderPath) << ",\"filter_linear\":" << (s.filterLinear ? "true" : "false")
        << ",\"wrap\":" << jstr(s.wrapMode) << ",\"mipmap\":" << (s.mipmapInput ? "true" : "false") << ",\"alias\":" << jstr(s.alias)
        << ",\"float_fb\":" << (s.floatFramebuffer ? "true" : "false") << ",\"srgb_fb\":" << (s.srgbFramebuffer ? "true" : "false")
        << ",\"fcm\":" << s.frameCountMod << ",\"stx\":" << jstr(s.scaleTypeX) << ",\"sx\":" << jnum(s.scaleX)
        << ",\"sty\":" << jstr(s.scaleTypeY) << ",\"sy\":" << jnum(s.scaleY) << "}";
      first = false;
    }
    o << "],\"textures\":{";
    first = true;
    for (const auto& t : p.getTextures()) {
      o << (first ? "" : ",") << jstr(t.first) << ":{\"path\":" << jstr(t.second.path) << ",\"wrap\":" << jstr(t.second.wrapMode)
        << ",\"mipmap\":" << (t.second.mipmap ? "true" : "false") << ",\"linear\":" << (t.second.linear ? "true" : "false") << "}";
      first = false;
    }
    o << "},\"params\":{";
    first = true;
    for (const auto& q : p.getParameters()) {
      o << (first ? "" : ",") << jstr(q.first) << ":" << jnum(q.second);
      first = false;
    }
    o << "}}";
    out = o.str();
  } catch (const std::exception& ex) {
    out = std::string("{\"preset\":") + jstr(path) + ",\"ok\":false,\"exception\":" + jstr(ex.what()) + "}";
  }
  return copy_out(out, buf, cap);
}

int rc_preset_save_as(const char* preset_path, const char* out_path, const char* const* names, const float* values, int n) {
  if (!preset_path || !out_path || n < 0 || (n > 0 && (!names || !values))) return RC_ERR_INVALID;
  return guarded([&] {
    rc::ShaderPreset p;
    rc::MissingSourceScope missing(true);
    if (!p.load(preset_path)) return (int)RC_ERR_LOAD;
    std::unordered_map<std::string, float> custom;
    for (int i = 0; i < n; ++i)
      if (names[i]) custom[names[i]] = values[i];
    return p.saveAs(out_path, custom) ? (int)RC_OK : (int)RC_ERR_LOAD;
  });
}

int rc_png_decode_rgba8(const char* path, void* rgba, size_t cap, int* width, int* height) {
  if (!path || !rgba) return RC_ERR_INVALID;
  return guarded([&] {
    std::vector<uint8_t> px;
    int w = 0, h = 0;
    std::string err;
    if (!rc::loadPngRgba8(path, &px, &w, &h, &err)) {
      RC_LOG_ERROR(err);
      return (int)RC_ERR_LOAD;
    }
    if (width) *width = w;
    if (height) *height = h;
    if (px.size() > cap) return (int)RC_ERR_INVALID;
    std::memcpy(rgba, px.data(), px.size());
    return (int)RC_OK;
  });
}

size_t rc_shader_params_json(const char* path, char* buf, size_t cap) {
  if (!path) return 0;
  std::string out;
  try {
    rc::ShaderSourceInfo info = rc::scanShaderSource(path);
    std::ostringstream o;
    o << "{\"readable\":" << (info.readable ? "true" : "false") << ",\"parameter_uniform\":" << (info.parameterUniform ? "true" : "false")
      << ",\"params\":[";
    bool first = true;
    for (const auto& name : info.declarationOrder) {
      const auto& p = info.parameterInfo.at(name);
      o << (first ? "" : ",") << "{\"name\":" << jstr(name) << ",\"description\":" << jstr(p.description) << ",\"default\":" << jnum(p.defaultValue)
        << ",\"min\":" << jnum(p.min) << ",\"max\":" << jnum(p.max) << ",\"step\":" << jnum(p.step) << "}";
      first = false;
    }
    o << "]}";
    out = o.str();
  } catch (const std::exception& ex) {
    out = std::string("{\"readable\":false,\"exception\":") + jstr(ex.what()) + "}";
  }
  return copy_out(out, buf, cap);
}

}  // extern "C"
