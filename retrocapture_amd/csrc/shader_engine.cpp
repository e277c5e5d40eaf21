// ShaderEngine on HIP.  Section references are to the reference implementation
// (src/shader/ShaderEngine.cpp) whose observable behaviour each block keeps.
#include "shader_engine.h"

#include <atomic>
#include "varying.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <filesystem>

#include "png_lut.h"
#include "rc_log.h"
#include "srgb_encode.h"

namespace fs = std::filesystem;

namespace rc {
namespace {

int wrapFromString(const std::string& w) {  // ShaderEngine.cpp:3208-3228
  if (w == "repeat") return rcd::WRAP_REPEAT;
  if (w == "mirrored_repeat") return rcd::WRAP_MIRROR;
  if (w == "clamp_to_border") return rcd::WRAP_BORDER;
  return rcd::WRAP_EDGE;
}

size_t texelBytes(int fmt) { return fmt == rcd::FMT_F32 ? 16 : (fmt == rcd::FMT_F16 ? 8 : 4); }

bool hipOk(hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  RC_LOG_ERROR(std::string(what) + ": " + hipGetErrorString(e));
  return false;
}

std::string lowerExt(const std::string& path) {
  std::string e = fs::path(path).extension().string();
  std::transform(e.begin(), e.end(), e.begin(), [](unsigned char c) { return (char)std::tolower(c); });
  return e;
}

// Uniforms the reference overwrites with fixed values after the #pragma parameters
// (ShaderEngine.cpp:2260-2374 and :2382-2392).
const std::pair<const char*, float> kHardCoded[] = {
    {"BLURSCALEX", 0.30f}, {"LOWLUMSCAN", 6.0f}, {"HILUMSCAN", 8.0f}, {"BRIGHTBOOST", 1.25f},
    {"MASK_DARK", 0.25f}, {"MASK_FADE", 0.8f}, {"RESSWITCH_ENABLE", 1.0f},
    {"RESSWITCH_GLITCH_TRESHOLD", 0.1f}, {"RESSWITCH_GLITCH_BAR_STR", 0.6f},
    {"RESSWITCH_GLITCH_BAR_SIZE", 0.5f}, {"RESSWITCH_GLITCH_BAR_SMOOTH", 1.0f},
    {"RESSWITCH_GLITCH_SHAKE_MAX", 0.25f}, {"RESSWITCH_GLITCH_ROT_MAX", 0.2f},
    {"RESSWITCH_GLITCH_WOB_MAX", 0.1f}, {"AS", 0.20f}, {"asat", 0.33f}, {"PR", 0.32f},
    {"PG", 0.32f}, {"PB", 0.32f}, {"internal_res", 1.0f}, {"auto_res", 0.0f}};

}  // namespace

ShaderEngine::ShaderEngine() {}
ShaderEngine::~ShaderEngine() { shutdown(); }

bool ShaderEngine::init(int device, hipStream_t stream) {  // :22-60
  if (m_initialized) return true;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
    RC_LOG_ERROR("ShaderEngine: no HIP device available (this engine has no CPU fallback)");
    return false;
  }
  if (device >= 0) {
    if (!hipOk(hipSetDevice(device), "hipSetDevice")) return false;
    m_device = device;
  } else if (!hipOk(hipGetDevice(&m_device), "hipGetDevice")) {
    return false;
  }
  hipDeviceProp_t prop;
  if (hipOk(hipGetDeviceProperties(&prop, m_device), "hipGetDeviceProperties")) {
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
      RC_LOG_WARN(std::string("ShaderEngine: kernels are built for gfx950, device is ") + prop.gcnArchName);
  }
  m_stream = stream;
  m_srgbEnc = deviceSrgbRunTable(m_device);
  if (!m_srgbEnc) {
    RC_LOG_ERROR("ShaderEngine: could not create the sRGB8 encode table on the device");
    return false;
  }
  m_initialized = true;
  return true;
}

void ShaderEngine::attachClient(EngineClient* c) {
  if (c) m_clients.push_back(c);
}
void ShaderEngine::detachClient(EngineClient* c) {
  m_clients.erase(std::remove(m_clients.begin(), m_clients.end(), c), m_clients.end());
}

void ShaderEngine::shutdown() {  // :62-86
  // pipelines built on this engine drain their streams and let go of it first
  std::vector<EngineClient*> clients;
  clients.swap(m_clients);
  for (EngineClient* c : clients) c->engineGone();
  if (!m_initialized) return;
  destroyHelper();
  disableShader();
  cleanupPresetPasses();
  cleanupTextureReferences();
  if (m_batchOutput.ptr) (void)hipFree(m_batchOutput.ptr);
  m_batchOutput = DeviceBuffer();
  m_initialized = false;
}

void ShaderEngine::cleanupPresetPasses() {
  for (auto& p : m_passes) {
    if (p.target.ptr) (void)hipFree(p.target.ptr);
    if (p.scratch.ptr) (void)hipFree(p.scratch.ptr);
    if (p.feedback.ptr) (void)hipFree(p.feedback.ptr);
    if (p.lastTarget.ptr) (void)hipFree(p.lastTarget.ptr);
    if (p.mips.ptr) (void)hipFree(p.mips.ptr);
  }
  m_passes.clear();
  if (m_sourceMips.ptr) (void)hipFree(m_sourceMips.ptr);
  m_sourceMips = DeviceBuffer();
  if (m_historyCleared.ptr) (void)hipFree(m_historyCleared.ptr);
  m_historyCleared = DeviceBuffer();
  m_historyClearedBytes = 0;
  for (auto& h : m_frameHistory)
    if (h.buf.ptr) (void)hipFree(h.buf.ptr);
  m_frameHistory.clear();
  m_units.clear();
  m_pass0Units.clear();
}

void ShaderEngine::cleanupTextureReferences() {
  for (auto& t : m_textureReferences)
    if (t.second.data.ptr) (void)hipFree(t.second.data.ptr);
  m_textureReferences.clear();
}

std::string ShaderEngine::getPresetPath() const {
  std::lock_guard<std::mutex> lock(m_paramMutex);
  return m_presetPath;
}

void ShaderEngine::disableShader() {
  m_shaderActive = false;
}

// A single .glsl file runs as a one-pass preset (the reference has a separate "simple mode",
// :149-226, whose draw is the same full-target quad; its output size is the input size).
bool ShaderEngine::loadShader(const std::string& shaderPath) {
  if (!m_initialized) {
    RC_LOG_ERROR("ShaderEngine not initialized");
    return false;
  }
  ++m_configEpoch;
  disableShader();
  cleanupPresetPasses();
  cleanupTextureReferences();
  m_preset.clear();
  {
    std::lock_guard<std::mutex> lock(m_paramMutex);
    m_customParameters.clear();
    m_presetPath.clear();
  }
  if (lowerExt(shaderPath) == ".slang") {
    RC_LOG_ERROR("Slang shaders (.slang) are not supported; use GLSL shaders (.glsl)");
    return false;
  }
  m_passes.resize(1);
  m_passes[0].passInfo.shaderPath = shaderPath;
  m_passes[0].passInfo.scaleTypeX = m_passes[0].passInfo.scaleTypeY = "source";
  if (!compilePass(0)) {
    cleanupPresetPasses();
    return false;
  }
  m_singleShader = true;
  m_shaderActive = true;
  return true;
}

bool ShaderEngine::loadPreset(const std::string& presetPath) {  // :228-319
  if (!m_initialized) {
    RC_LOG_ERROR("ShaderEngine not initialized");
    return false;
  }
  ++m_configEpoch;
  disableShader();
  cleanupPresetPasses();
  cleanupTextureReferences();
  m_singleShader = false;

  const std::string ext = lowerExt(presetPath);
  if (ext == ".slangp") {
    RC_LOG_ERROR("Slang presets (.slangp) are not supported. Use GLSLP presets (.glslp)");
    return false;
  }
  if (ext != ".glslp" && !ext.empty()) RC_LOG_WARN("Unrecognized preset extension: " + ext + ". Expected .glslp");
  {
    std::lock_guard<std::mutex> lock(m_paramMutex);
    m_customParameters.clear();
  }
  MissingSourceScope missing(m_allowMissingSources);
  if (!m_preset.load(presetPath)) return false;
  {
    std::lock_guard<std::mutex> lock(m_paramMutex);
    m_presetPath = presetPath;
  }
  for (const auto& tex : m_preset.getTextures())
    if (!loadTextureReference(tex.first, tex.second.path)) RC_LOG_ERROR("Failed to load reference texture: " + tex.first);

  const bool passesLoaded = loadPresetPasses();
  size_t totalParams = 0;
  for (const auto& p : m_passes) totalParams += p.parameterInfo.size();
  if (passesLoaded) {
    RC_LOG_INFO("Preset loaded with " + std::to_string(m_passes.size()) + " pass(es) and " +
                std::to_string(totalParams) + " parameter(s)");
  } else if (totalParams > 0) {
    RC_LOG_WARN("Preset loaded but some passes have no kernel; " + std::to_string(totalParams) +
                " parameter(s) extracted");
  } else {
    RC_LOG_WARN("Preset loaded but no pass is usable");
  }
  if (missing.count() > 0)
    RC_LOG_WARN("Preset " + presetPath + ": " + std::to_string(missing.count()) +
                " shader file lookup(s) found no file on this machine; those passes run from the kernel registry's parameter tables");
  // As in the reference, a preset that parsed is "active" even if passes failed (:316-318).
  m_shaderActive = true;
  return true;
}

// Render-target format of a pass (:2882-2890): RGBA32F for float_framebuffer, else SRGB8_ALPHA8 for srgb_framebuffer,
// else RGBA8.  setFloatTargetFp16(true) stores the float targets as four binary16 values instead (not the reference's
// format: an opt-in that halves their traffic, within the tolerance stated in DESIGN.md).
int ShaderEngine::targetFormat(const ShaderPass& pi) const {
  if (pi.floatFramebuffer) return m_floatTargetFp16 ? rcd::FMT_F16 : rcd::FMT_F32;
  return pi.srgbFramebuffer ? rcd::FMT_SRGB8 : rcd::FMT_RGBA8;
}

bool ShaderEngine::loadPresetPasses() {  // :750-848
  const auto& passes = m_preset.getPasses();
  cleanupPresetPasses();
  m_passes.resize(passes.size());
  bool all = true;
  for (size_t i = 0; i < passes.size(); ++i) {
    m_passes[i].passInfo = passes[i];
    if (!compilePass(i)) all = false;
  }
  return all;
}

// "Compiling" a pass = reading its #pragma parameters from the shader file and finding the
// hand-written kernel registered for that shader (:321-748).
bool ShaderEngine::compilePass(size_t i) {
  ShaderPassData& pd = m_passes[i];
  const ShaderPass& pi = pd.passInfo;
  pd.kernel = nullptr;
  pd.format = targetFormat(pi);

  if (lowerExt(pi.shaderPath) == ".slang") {
    RC_LOG_ERROR("Slang shaders (.slang) are not supported in pass " + std::to_string(i));
    return false;
  }
  const KernelEntry* entry = findKernel(pi.shaderPath);
  ShaderSourceInfo src = scanShaderSource(pi.shaderPath);
  if (!src.readable) {
    if (!m_allowMissingSources || !entry) {
      RC_LOG_ERROR("Failed to open shader for pass " + std::to_string(i) + ": " + pi.shaderPath);
      return false;
    }
    // No shader text on this machine: take the parameter table the registry carries.
    if (!note_missing_source()) RC_LOG_WARN("Shader file not readable, using the built-in parameter table: " + pi.shaderPath);
    for (const KernelParam& kp : entry->params) {
      ShaderParameterInfo info;
      info.defaultValue = kp.def;
      info.min = kp.min;
      info.max = kp.max;
      info.step = kp.step;
      info.description = kp.description;
      src.parameterInfo[kp.name] = info;
    }
  }
  pd.parameterInfo = src.parameterInfo;
  pd.extractedParameters.clear();
  for (const auto& kv : src.parameterInfo) pd.extractedParameters[kv.first] = kv.second.defaultValue;

  if (!entry) {
    RC_LOG_ERROR("No HIP kernel is registered for shader of pass " + std::to_string(i) + " (" +
                 shaderIdentity(pi.shaderPath) + "); the pass will be skipped");
    return false;
  }
  // mipmap_input (:1022-1033, GL_LINEAR_MIPMAP_LINEAR + glGenerateMipmap on the input): no mip chain is
  // built here.  Accepted only where level 0 is all that is sampled - a pass whose target has the size of
  // its input (checked per frame in runChunk; crt-royale-fake-bloom's last pass) - see DESIGN.md section 3
  // for the 1e-5-level residual this leaves against llvmpipe.
  pd.kernel = entry;
  return true;
}

bool ShaderEngine::loadTextureReference(const std::string& name, const std::string& path) {  // :2535-2706
  if (m_textureReferences.count(name)) return true;
  std::vector<uint8_t> rgba;
  int w = 0, h = 0;
  std::string err;
  if (!loadPngRgba8(path, &rgba, &w, &h, &err)) {
    RC_LOG_ERROR("Texture " + name + ": " + err);
    return false;
  }
  LutTexture t;
  t.width = w;
  t.height = h;
  if (!ensureBuffer(t.data, rgba.size())) return false;
  if (!hipOk(hipMemcpy(t.data.ptr, rgba.data(), rgba.size(), hipMemcpyHostToDevice), "LUT upload")) return false;
  m_textureReferences[name] = t;
  return true;
}

void ShaderEngine::setMaxShaderResolution(uint32_t maxWidth, uint32_t maxHeight) {
  m_maxShaderWidth = maxWidth;
  m_maxShaderHeight = maxHeight;
}

void ShaderEngine::setViewport(uint32_t width, uint32_t height) {  // :3154-3206
  if (m_maxShaderWidth > 0 && m_maxShaderHeight > 0 && (width > m_maxShaderWidth || height > m_maxShaderHeight)) {
    const float aspect = (float)width / (float)height;
    uint32_t w = width, h = height;
    if (width > m_maxShaderWidth) {
      w = m_maxShaderWidth;
      h = (uint32_t)std::round(m_maxShaderWidth / aspect);
    }
    if (h > m_maxShaderHeight) {
      h = m_maxShaderHeight;
      w = (uint32_t)std::round(m_maxShaderHeight * aspect);
    }
    m_viewportWidth = (w / 2) * 2;
    m_viewportHeight = (h / 2) * 2;
    return;
  }
  m_viewportWidth = width;
  m_viewportHeight = height;
}

// In the reference these only feed the single-shader mode's uniform map (:3006-3041); the
// kernels take their inputs from the preset, so the values are recorded and otherwise unused.
void ShaderEngine::setUniform(const std::string& name, float value) { m_uniforms[name] = {value, 0, 0, 0}; }
void ShaderEngine::setUniform(const std::string& name, float x, float y) { m_uniforms[name] = {x, y, 0, 0}; }
void ShaderEngine::setUniform(const std::string& name, float x, float y, float z, float w) { m_uniforms[name] = {x, y, z, w}; }

std::vector<ShaderEngine::ShaderParameter> ShaderEngine::getShaderParameters() const {  // :3264-3351
  std::vector<ShaderParameter> out;
  if (!m_shaderActive || m_passes.empty()) return out;
  std::lock_guard<std::mutex> lock(m_paramMutex);
  std::map<std::string, ShaderParameter> byName;
  for (const auto& pd : m_passes)
    for (const auto& kv : pd.parameterInfo) {
      if (byName.count(kv.first)) continue;  // first pass that declares it wins
      ShaderParameter p;
      p.name = kv.first;
      p.defaultValue = kv.second.defaultValue;
      p.min = kv.second.min;
      p.max = kv.second.max;
      p.step = kv.second.step;
      p.description = kv.second.description;
      auto c = m_customParameters.find(kv.first);
      if (c != m_customParameters.end()) {
        p.value = c->second;
      } else {
        auto g = m_preset.getParameters().find(kv.first);
        p.value = g != m_preset.getParameters().end() ? g->second : kv.second.defaultValue;
      }
      byName[kv.first] = p;
    }
  for (const auto& kv : byName) out.push_back(kv.second);
  return out;
}

bool ShaderEngine::setShaderParameter(const std::string& name, float value) {  // :3353-3387
  if (!m_shaderActive) return false;
  for (const auto& pd : m_passes) {
    auto it = pd.parameterInfo.find(name);
    if (it == pd.parameterInfo.end()) continue;
    const float clamped = std::max(it->second.min, std::min(it->second.max, value));
    std::lock_guard<std::mutex> lock(m_paramMutex);
    m_customParameters[name] = clamped;
    return true;
  }
  return false;
}

// Value a pass's kernel sees for one of its parameters.  Order of the reference's uniform
// writes (later write wins): #pragma value as custom > preset > default (:2227-2249), then
// the hard-coded names (:2260-2374, :2382-2392), then every global preset parameter by name
// (:2512-2520) - so a preset-file value beats a custom one at draw time.
float ShaderEngine::effectiveParameter(const ShaderPassData& pass, const KernelParam& kp,
                                       const std::map<std::string, float>& custom) const {
  float v = kp.def;
  auto declared = pass.extractedParameters.find(kp.name);
  const bool isDeclared = declared != pass.extractedParameters.end();
  if (isDeclared) {
    v = declared->second;
    auto c = custom.find(kp.name);
    if (c != custom.end()) {
      v = c->second;
    } else {
      auto g = m_preset.getParameters().find(kp.name);
      if (g != m_preset.getParameters().end()) v = g->second;
    }
  }
  for (const auto& hc : kHardCoded)
    if (hc.first == std::string(kp.name)) v = hc.second;
  auto g = m_preset.getParameters().find(kp.name);
  if (g != m_preset.getParameters().end()) v = g->second;
  return v;
}

uint32_t ShaderEngine::calculateScale(uint32_t sourceSize, const std::string& scaleType, float scale,
                                      uint32_t viewportSize) const {  // :1881-1910
  if (scaleType.empty() || scaleType == "source") {
    if (scale == 0.0f) scale = 1.0f;
    return (uint32_t)std::round(sourceSize * scale);
  }
  if (scaleType == "viewport") {
    if (scale == 0.0f) scale = 1.0f;
    return (uint32_t)std::round(viewportSize * scale);
  }
  if (scaleType == "absolute") return (uint32_t)std::round(scale);
  return sourceSize;
}

// Output size of every pass for this input size / viewport (:856-910).
void ShaderEngine::resolvePassSizes(uint32_t width, uint32_t height) {
  uint32_t cw = width, ch = height;
  for (size_t i = 0; i < m_passes.size(); ++i) {
    ShaderPassData& pd = m_passes[i];
    const ShaderPass& pi = pd.passInfo;
    std::string tx = pi.scaleTypeX, ty = pi.scaleTypeY;
    float sx = pi.scaleX, sy = pi.scaleY;
    const bool last = (i == m_passes.size() - 1);
    if (!m_singleShader) {
      // last pass: "source x1.0" (or unspecified) means the viewport, per axis (:871-889)
      if (last && tx != "viewport" && (tx.empty() || (tx == "source" && sx == 1.0f))) {
        tx = "viewport";
        sx = 1.0f;
      }
      if (last && ty != "viewport" && (ty.empty() || (ty == "source" && sy == 1.0f))) {
        ty = "viewport";
        sy = 1.0f;
      }
    }
    uint32_t ow = calculateScale(cw, tx, sx, m_viewportWidth);
    uint32_t oh = calculateScale(ch, ty, sy, m_viewportHeight);
    if (m_maxShaderWidth > 0 && ow > m_maxShaderWidth) {  // :897-903
      const float aspect = (float)ow / (float)oh;
      ow = m_maxShaderWidth;
      oh = ((uint32_t)std::round(m_maxShaderWidth / aspect) / 2) * 2;
    }
    if (m_maxShaderHeight > 0 && oh > m_maxShaderHeight) {  // :904-910
      const float aspect = (float)ow / (float)oh;
      oh = m_maxShaderHeight;
      ow = ((uint32_t)std::round(m_maxShaderHeight * aspect) / 2) * 2;
    }
    pd.width = ow;
    pd.height = oh;
    pd.format = targetFormat(pi);   // (the float storage choice can change between frames)
    pd.frameBytes = (size_t)ow * oh * texelBytes(pd.format);
    cw = ow;
    ch = oh;
  }
}

bool ShaderEngine::ensureBuffer(DeviceBuffer& b, size_t bytes) {
  if (b.ptr && b.bytes >= bytes) return true;
  if (b.ptr) (void)hipFree(b.ptr);
  b = DeviceBuffer();
  if (bytes == 0) return true;
  if (!hipOk(hipMalloc(&b.ptr, bytes), "hipMalloc")) return false;
  b.bytes = bytes;
  static std::atomic<uint64_t> generation{0};
  b.gen = ++generation;   // (a frame-invariant pass kept in a buffer that was re-allocated must be rendered again)
  return true;
}

// Texture view of pass p's target with the sampler state that persists on it: a pass's
// output is first consumed as the next pass's input, which sets filter/wrap on the texture
// object (:1008-1036); later PassPrev / alias reads see that state.  The last pass's target
// keeps its creation state, LINEAR + CLAMP_TO_EDGE (:2903-2906).
rcd::Tex ShaderEngine::passTexture(size_t p) const {
  const ShaderPassData& pd = m_passes[p];
  if (pd.folded) return pd.foldedView;   // (never written this chunk: the pass's input, seen through its decode table)
  rcd::Tex t;
  t.base = pd.target.ptr;
  t.frame_stride = pd.invariant ? 0 : pd.frameBytes;
  t.w = (int)pd.width;
  t.h = (int)pd.height;
  t.fmt = pd.format;
  if (p + 1 < m_passes.size()) {
    t.linear = m_passes[p + 1].passInfo.filterLinear ? 1 : 0;
    t.wrap = wrapFromString(m_passes[p + 1].passInfo.wrapMode);
  } else {
    t.linear = 1;
    t.wrap = rcd::WRAP_EDGE;
  }
  if (pd.mipLevels > 1 && pd.mips.ptr) {
    t.n_levels = pd.mipLevels;
    t.mip_base = pd.mips.ptr;
    t.mip_frame_stride = pd.mipFrameBytes;
  }
  return t;
}

// glGenerateMipmap as llvmpipe does it (measured, oracle/rc_sampler.c): every level is a LINEAR,
// clamp-to-edge draw of the level above in the texture's own format - the stock kernel with the ordinary sampler
// (sRGB8: decoded, filtered in float, re-encoded; RGBA8 / GL_RGB: the 8-bit fixed-point filter - NOT the blit
// fast path a plain RGBA8 -> RGBA8 copy takes).
bool ShaderEngine::buildMipLevels(const rcd::Tex& level0, uint32_t nFrames, DeviceBuffer* mips, int* outLevels, size_t* outFrameBytes) {
  const int storeFmt = level0.fmt == rcd::FMT_RGBX8 ? rcd::FMT_RGBA8 : level0.fmt;   // GL_RGB: alpha reads 1 at every level
  const uint32_t bpp = texelBytes(storeFmt);
  int levels = 1;
  size_t bytes = 0;
  for (uint32_t w = (uint32_t)level0.w, h = (uint32_t)level0.h; w > 1 || h > 1;) {
    w = std::max(1u, w >> 1);
    h = std::max(1u, h >> 1);
    bytes += (size_t)w * h * bpp;
    if (++levels == 15) break;
  }
  *outLevels = levels;
  *outFrameBytes = bytes;
  if (levels <= 1) return true;
  if (!ensureBuffer(*mips, bytes * nFrames)) return false;
  rcd::Tex src = level0;
  src.linear = 1;
  src.wrap = rcd::WRAP_EDGE;
  src.n_levels = 0;
  src.mip_base = nullptr;
  size_t off = 0;
  for (int k = 1; k < levels; ++k) {
    const int dw = std::max(1, level0.w >> k), dh = std::max(1, level0.h >> k);
    rcd::PassLaunch L;
    std::memset(static_cast<void*>(&L), 0, sizeof(L));
    L.srgb_enc = m_srgbEnc;
    L.in = src;
    L.out = static_cast<uint8_t*>(mips->ptr) + off;
    L.out_frame_stride = bytes;
    L.out_w = dw;
    L.out_h = dh;
    L.out_fmt = storeFmt;
    L.src_w = src.w;
    L.src_h = src.h;
    L.vp_w = dw;
    L.vp_h = dh;
    L.n_frames = (int)nFrames;
    L.flags = rcd::RC_FLAG_STOCK_NO_BLIT;
    L.plane[0] = makePlaneFan(0.f, 1.f, 1.f, 0.f, dw, dh, storeFmt);   // drawn by the GL's blitter: a triangle fan
    L.plane[1] = makePlaneFan(0.f, 0.f, 1.f, 1.f, dw, dh, storeFmt);
    if (!hipOk(rck::launch_stock(L, m_stream), "mip level")) return false;
    src.base = L.out;
    src.frame_stride = bytes;
    src.w = dw;
    src.h = dh;
    src.fmt = storeFmt;
    off += (size_t)dw * dh * bpp;
  }
  return true;
}
bool ShaderEngine::buildMipChain(size_t p, const void* level0, uint32_t nFrames) {
  ShaderPassData& pd = m_passes[p];
  rcd::Tex t;
  t.base = level0;
  t.frame_stride = pd.frameBytes;
  t.w = (int)pd.width;
  t.h = (int)pd.height;
  t.fmt = pd.format;
  return buildMipLevels(t, nFrames, &pd.mips, &pd.mipLevels, &pd.mipFrameBytes);
}

rcd::Tex ShaderEngine::lutTexture(const std::string& name) const {  // :1361-1415
  rcd::Tex t;
  std::memset(&t, 0, sizeof(t));
  auto lut = m_textureReferences.find(name);
  if (lut == m_textureReferences.end()) return t;
  t.base = lut->second.data.ptr;
  t.frame_stride = 0;
  t.w = lut->second.width;
  t.h = lut->second.height;
  t.fmt = rcd::FMT_RGBA8;
  t.linear = 1;
  t.wrap = rcd::WRAP_EDGE;
  auto st = m_preset.getTextures().find(name);
  if (st != m_preset.getTextures().end()) {
    t.linear = st->second.linear ? 1 : 0;
    t.wrap = wrapFromString(st->second.wrapMode);
  }
  return t;
}

namespace {
std::string prevName(int k) { return k == 0 ? std::string("PrevTexture") : "Prev" + std::to_string(k) + "Texture"; }
bool declares(const KernelEntry& k, const std::string& name) {
  for (const char* s : k.samplers)
    if (name == s) return true;
  return false;
}
}  // namespace

bool ShaderEngine::presetSamplesHistory() const {
  if (m_passes.empty() || !m_passes[0].kernel) return false;
  for (int k = 0; k < 7; ++k)
    if (declares(*m_passes[0].kernel, prevName(k)) || declares(*m_passes[0].kernel, "PassPrev" + std::to_string(k) + "Texture"))
      return true;
  return false;
}

// The reference binds, in this order and on consecutive texture units starting at 1: frame history
// (pass 0 only, and only as far as history exists, :1095-1159) or the outputs of earlier passes under
// their PassPrev / Prev names (:1163-1228), the original input for PassPrev<N> with N beyond pass 0
// (:1234-1245), earlier passes by alias (:1251-1277), OrigTexture (:1351-1358) and every preset LUT
// (:1361-1415; a LUT takes a unit whether or not the program declares it).  A sampler uniform that
// is not set keeps its value: 0 (= the pass input on unit 0) on a fresh program.
bool ShaderEngine::presetSamplesFeedback() const {
  for (const auto& p : m_passes)
    if (p.kernel)
      for (const char* s : p.kernel->samplers)
        if (std::strncmp(s, "PassFeedback", 12) == 0) return true;
  return false;
}

// Whether every later pass that samples pass i's target does so through a texture its kernel can read with a decode table
// (KernelEntry::decode_table_inputs) - the static side of bindSamplers: the next pass's `Texture`, PassPrev<n> / Prev names, the
// alias; PassFeedback, frame history through pass 0, a mip chain on the target and a program-less consumer all rule folding out.
bool ShaderEngine::consumersTakeDecodeTable(size_t i) const {
  if (i + 1 >= m_passes.size() || m_passes[i].feedbackEnabled || presetSamplesFeedback()) return false;
  if (i == 0 && presetSamplesHistory()) return false;
  if (m_passes[i + 1].passInfo.mipmapInput) return false;
  for (size_t j = i + 1; j < m_passes.size(); ++j) {
    const KernelEntry* kj = m_passes[j].kernel;
    if (!kj) {
      if (j == i + 1) return false;   // (its cleared target replaces the chain's texture; keep the ordinary path)
      continue;
    }
    if (j == i + 1 && kj->reads_input && !(kj->decode_table_inputs & 1u)) return false;
    const std::string& al = m_passes[i].passInfo.alias;
    for (size_t s = 0; s < kj->samplers.size() && s < (size_t)rcd::kMaxExtra; ++s) {
      const std::string n = kj->samplers[s];
      const bool names_i = n == "PassPrev" + std::to_string(j - i) + "Texture" || n == prevName((int)i) || (!al.empty() && n == al);
      // a sampler left unbound reads unit 0, the pass's own input (bindSamplers): for the next pass that is this target too
      bool unbound = j == i + 1;
      if (unbound) {
        for (size_t pp = 0; pp < j && unbound; ++pp)
          unbound = !(n == "PassPrev" + std::to_string(j - pp) + "Texture" || n == prevName((int)pp) ||
                      (!m_passes[pp].passInfo.alias.empty() && n == m_passes[pp].passInfo.alias));
        for (size_t q = j + 1; q <= j + 12 && unbound; ++q) unbound = n != "PassPrev" + std::to_string(q) + "Texture";
        if (n == "OrigTexture" || m_textureReferences.count(n) || n.rfind("PassFeedback", 0) == 0) unbound = false;
      }
      if ((names_i || unbound) && !(kj->decode_table_inputs & (1u << (1 + s)))) return false;
    }
  }
  return true;
}

bool ShaderEngine::bindSamplers(size_t i, const KernelEntry& k, const rcd::Tex& inputTex, const rcd::Tex& sourceTex,
                                rcd::PassLaunch* L, bool* lostDraw) {
  *lostDraw = false;
  if (m_units.size() < 64) m_units.resize(64);
  std::map<std::string, int> bound;
  int unit = 1;
  auto bind = [&](const std::string& name, const rcd::Tex& t) {
    if (unit < (int)m_units.size()) m_units[(size_t)unit] = t;
    bound[name] = unit++;
  };
  if (i == 0) {
    for (int h = 0; h < 7; ++h) {
      const std::string names[2] = {prevName(h), "PassPrev" + std::to_string(h) + "Texture"};
      for (const std::string& n : names)
        if (declares(k, n)) {
          if ((size_t)h < m_frameHistory.size() && m_frameHistory[(size_t)h].buf.ptr) {
            rcd::Tex t;
            std::memset(&t, 0, sizeof(t));
            t.base = m_frameHistory[(size_t)h].buf.ptr;
            t.frame_stride = 0;
            t.w = (int)m_frameHistory[(size_t)h].width;
            t.h = (int)m_frameHistory[(size_t)h].height;
            t.fmt = rcd::FMT_RGBA8;
            t.linear = 1;  // creation state of a history texture, never changed (:1757-1761)
            t.wrap = rcd::WRAP_EDGE;
            bind(n, t);
          }
          break;
        }
    }
  } else {
    for (size_t pp = 0; pp < i; ++pp) {
      const std::string names[2] = {"PassPrev" + std::to_string(i - pp) + "Texture", prevName((int)pp)};
      for (const std::string& n : names)
        if (declares(k, n)) {
          bind(n, passTexture(pp));
          break;
        }
    }
    for (size_t n = i + 1; n <= i + 12; ++n) {
      const std::string name = "PassPrev" + std::to_string(n) + "Texture";
      if (declares(k, name)) bind(name, sourceTex);
    }
    for (size_t pp = 0; pp < i; ++pp) {
      const std::string& al = m_passes[pp].passInfo.alias;
      if (!al.empty() && declares(k, al)) bind(al, passTexture(pp));
    }
  }
  // PassFeedback<fp> for fp <= i (:1285-1347): the partner texture of pass fp, created (zero-filled,
  // LINEAR / clamp to edge) the first time any program asks for it.  createFramebuffer ends with
  // framebuffer 0 bound (:2931), so the draw of the pass that triggered the creation does not reach
  // its target, which keeps its clear colour - reproduced through *lostDraw.
  for (size_t fp = 0; fp <= i && fp < m_passes.size(); ++fp) {
    const std::string names[2] = {"PassFeedback" + std::to_string(fp), "PassFeedback" + std::to_string(fp) + "Texture"};
    const std::string* hit = nullptr;
    for (const std::string& n : names)
      if (declares(k, n)) {
        hit = &n;
        break;
      }
    if (!hit) continue;
    ShaderPassData& t = m_passes[fp];
    t.feedbackEnabled = true;
    if (!t.feedback.ptr && t.frameBytes > 0) {
      if (!ensureBuffer(t.feedback, t.frameBytes)) return false;
      if (!hipOk(hipMemsetAsync(t.feedback.ptr, 0, t.frameBytes, m_stream), "feedback clear")) return false;
      t.feedbackLinear = 1;
      t.feedbackWrap = rcd::WRAP_EDGE;
      t.feedbackWidth = t.width;
      t.feedbackHeight = t.height;
      t.feedbackFormat = t.format;
      *lostDraw = true;
    }
    rcd::Tex ft = passTexture(fp);
    ft.base = t.feedback.ptr;
    ft.frame_stride = 0;
    ft.linear = t.feedbackLinear;
    ft.wrap = t.feedbackWrap;
    bind(*hit, ft);
  }
  if (declares(k, "OrigTexture")) bind("OrigTexture", sourceTex);
  for (const auto& lt : m_preset.getTextures())  // std::map: by name
    if (m_textureReferences.count(lt.first)) bind(lt.first, lutTexture(lt.first));
  if (i == 0)
    for (const auto& b : bound) m_pass0Units[b.first] = b.second;
  const std::map<std::string, int>& units = (i == 0) ? m_pass0Units : bound;
  for (size_t s = 0; s < k.samplers.size() && s < (size_t)rcd::kMaxExtra; ++s) {
    auto it = units.find(k.samplers[s]);
    const int u = it == units.end() ? 0 : it->second;
    L->extra[s] = (u > 0 && u < (int)m_units.size()) ? m_units[(size_t)u] : inputTex;
  }
  return true;
}

const void* ShaderEngine::applyShader(const void* input, uint32_t width, uint32_t height) {
  return applyShaderBatch(input, 1, width, height, 0);
}

void ShaderEngine::destroyHelper() {
  if (m_helper) {
    if (m_helperStream) (void)hipStreamSynchronize(m_helperStream);
    m_helper->m_externalOut = nullptr;
    m_helper->shutdown();
    m_helper.reset();
  }
  if (m_laneFork) (void)hipEventDestroy(m_laneFork);
  if (m_laneJoin) (void)hipEventDestroy(m_laneJoin);
  if (m_helperStream) (void)hipStreamDestroy(m_helperStream);
  m_laneFork = m_laneJoin = nullptr;
  m_helperStream = nullptr;
  m_helperEpoch = 0;
}

// The second lane: created on first use, reloaded when this engine's preset changed, and given this engine's settings
// before every batch.  false: stay on one lane.
bool ShaderEngine::syncHelper() {
  if (m_singleShader || m_presetPath.empty()) return false;
  if (!m_helper) {
    if (!hipOk(hipStreamCreateWithFlags(&m_helperStream, hipStreamNonBlocking), "hipStreamCreate") ||
        !hipOk(hipEventCreateWithFlags(&m_laneFork, hipEventDisableTiming), "hipEventCreate") ||
        !hipOk(hipEventCreateWithFlags(&m_laneJoin, hipEventDisableTiming), "hipEventCreate")) {
      destroyHelper();
      return false;
    }
    m_helper.reset(new ShaderEngine());
    if (!m_helper->init(m_device, m_helperStream)) {
      destroyHelper();
      return false;
    }
  }
  ShaderEngine& o = *m_helper;
  o.m_allowMissingSources = m_allowMissingSources;
  if (m_helperEpoch != m_configEpoch) {
    if (hipStreamSynchronize(m_helperStream) != hipSuccess) return false;
    if (!o.loadPreset(m_presetPath) || !o.m_shaderActive) return false;
    m_helperEpoch = m_configEpoch;
  }
  if (!o.m_shaderActive || o.m_passes.size() != m_passes.size()) return false;
  o.m_inputLinear = m_inputLinear;
  o.m_undefVaryingZero = m_undefVaryingZero;
  o.m_generalOnly = m_generalOnly;
  o.m_foldPasses = m_foldPasses;
  o.m_asyncTables = m_asyncTables;
  o.m_floatTargetFp16 = m_floatTargetFp16;
  o.m_chunk = m_chunk;
  o.m_chunkAuto = m_chunkAuto;
  o.m_maxShaderWidth = m_maxShaderWidth;
  o.m_maxShaderHeight = m_maxShaderHeight;
  o.m_viewportWidth = m_viewportWidth;
  o.m_viewportHeight = m_viewportHeight;
  o.m_uniforms = m_uniforms;
  o.m_time = m_time;
  std::map<std::string, float> custom;
  {
    std::lock_guard<std::mutex> lock(m_paramMutex);
    custom = m_customParameters;
  }
  {
    std::lock_guard<std::mutex> lock(o.m_paramMutex);
    o.m_customParameters = custom;
  }
  return true;
}

const void* ShaderEngine::applyShaderBatch(const void* inputs, uint32_t nFrames, uint32_t width, uint32_t height,
                                           uint64_t frameStride) {  // :1531-1879
  if (!m_shaderActive) return inputs;
  if (m_passes.empty()) return inputs;
  bool hasValidPass = false;
  for (const auto& p : m_passes) hasValidPass |= (p.kernel != nullptr);
  if (!hasValidPass) {
    RC_LOG_ERROR("No valid pass found in preset. Returning original frame.");
    return inputs;
  }
  if (!inputs) {
    RC_LOG_ERROR("applyShader: invalid input frame (null)");
    return nullptr;
  }
  if (nFrames == 0) return inputs;
  if (frameStride == 0) frameStride = (uint64_t)width * height * 4;

  // optional processing-resolution clamp (:1623-1660)
  uint32_t pw = width, ph = height;
  if (m_maxShaderWidth > 0 && m_maxShaderHeight > 0 && (width > m_maxShaderWidth || height > m_maxShaderHeight)) {
    const float aspect = (float)width / (float)height;
    if (width > m_maxShaderWidth) {
      pw = m_maxShaderWidth;
      ph = (uint32_t)std::round(m_maxShaderWidth / aspect);
    }
    if (ph > m_maxShaderHeight) {
      ph = m_maxShaderHeight;
      pw = (uint32_t)std::round(m_maxShaderHeight * aspect);
    }
    pw = (pw / 2) * 2;
    ph = (ph / 2) * 2;
  }
  m_sourceWidth = pw;
  m_sourceHeight = ph;
  if (m_singleShader || m_viewportWidth == 0 || m_viewportHeight == 0) {
    // single-shader mode renders at the input size; a preset needs setViewport() first
    if (m_viewportWidth == 0 || m_viewportHeight == 0) {
      m_viewportWidth = pw;
      m_viewportHeight = ph;
    }
  }
  resolvePassSizes(pw, ph);
  // the kernels address texels inside one frame with 32-bit byte offsets (rc_device.h texel_off)
  {
    const uint64_t limit = 1ull << 32;
    bool ok = (uint64_t)width * height * 4 < limit;
    for (const auto& p : m_passes) ok = ok && (uint64_t)p.frameBytes < limit;
    if (!ok) {
      RC_LOG_ERROR("applyShader: a frame or pass target of 4 GiB or more is not supported");
      return inputs;
    }
  }

  // buffers: every pass holds `chunk` frames, except the last which holds the whole batch
  // a preset whose first pass samples frame history is sequential: one frame per chunk, history
  // pushed after each (frames of a batch are successive frames)
  const bool history = presetSamplesHistory();
  const bool feedback = presetSamplesFeedback();
  if (history && feedback) {
    RC_LOG_ERROR("applyShader: a preset that samples both frame history and PassFeedback is not supported");
    return inputs;
  }
  uint32_t chunkFrames = m_chunk;
  if (m_chunkAuto) {
    uint64_t largest = (uint64_t)width * height * 4;
    for (const auto& p : m_passes) largest = std::max<uint64_t>(largest, p.frameBytes);
    if (largest * 128ull <= (1ull << 31)) chunkFrames = 128u;
  }
  // two lanes (setLanes): this engine renders the first half, the helper - on its own stream - the second half
  // (a single-pass preset whose kernel is marked so stays on one lane: KernelEntry::one_lane)
  const bool oneLaneKernel = m_passes.size() == 1 && m_passes[0].kernel && m_passes[0].kernel->one_lane;
  const bool twoLanes = m_lanes == 2 && !m_externalOut && !history && !feedback && !m_profiling && !oneLaneKernel && nFrames >= 2 && syncHelper();
  const uint32_t nOwn = twoLanes ? (nFrames + 1) / 2 : nFrames;
  m_lastTwoLanes = twoLanes;
  m_lastOwnFrames = nOwn;
  uint32_t chunk = (history || feedback) ? 1u : std::min(chunkFrames, nOwn);
  for (;;) {
    bool ok = true;
    for (size_t i = 0; ok && i + 1 < m_passes.size(); ++i) {
      // (a pass that may be folded into its consumers gets its target when it is first rendered: runChunk, readPass)
      const bool mayFold = m_foldPasses && !m_generalOnly && m_passes[i].kernel && m_passes[i].kernel->byte_map && consumersTakeDecodeTable(i);
      if (!mayFold) ok = ensureBuffer(m_passes[i].target, m_passes[i].frameBytes * chunk);
    }
    if (ok) break;
    // out of device memory at the automatic launch size (128 frames of every intermediate target, twice with two lanes): fall
    // back towards the old default of 8 frames per launch before giving up
    if (!m_chunkAuto || chunk <= 8u) return inputs;
    chunk = std::max(8u, chunk / 2u);
    RC_LOG_WARN("applyShader: not enough device memory for the intermediate targets; " + std::to_string(chunk) + " frames per launch");
  }
  m_chunkCapacity = chunk;
  ShaderPassData& lastPass = m_passes.back();
  if (!m_externalOut && !ensureBuffer(lastPass.target, lastPass.frameBytes * nFrames)) return inputs;
  uint8_t* const outBase = m_externalOut ? m_externalOut : static_cast<uint8_t*>(lastPass.target.ptr);
  if (twoLanes) {
    // the helper starts once everything enqueued on this engine's stream so far (the caller's frames) is done
    if (!hipOk(hipEventRecord(m_laneFork, m_stream), "hipEventRecord") ||
        !hipOk(hipStreamWaitEvent(m_helperStream, m_laneFork, 0), "hipStreamWaitEvent"))
      return inputs;
  }

  for (uint32_t f0 = 0; f0 < nOwn; f0 += chunk) {
    const uint32_t n = std::min(chunk, nOwn - f0);
    // FrameCount is a float that is incremented once per frame (:1688) and handed to int
    // uniforms by truncation (:2138)
    const int firstCount = (int)(m_frameCount + 1.0f);
    const uint8_t* in = static_cast<const uint8_t*>(inputs) + frameStride * f0;
    uint8_t* out = outBase + lastPass.frameBytes * f0;
    if (!runChunk(in, frameStride, width, height, n, firstCount, out)) return inputs;
    if (feedback) {
      // ping-pong swap (:1710-1718): what this frame wrote becomes next frame's "previous"; the texture
      // object that held it keeps the sampler state its consumer set on it
      for (size_t pi = 0; pi < m_passes.size(); ++pi) {
        ShaderPassData& fp = m_passes[pi];
        if (!fp.feedbackEnabled || !fp.feedback.ptr) continue;
        const rcd::Tex written = passTexture(pi);
        if (pi + 1 == m_passes.size()) std::swap(fp.lastTarget, fp.feedback);
        else std::swap(fp.target, fp.feedback);
        fp.feedbackLinear = written.linear;
        fp.feedbackWrap = written.wrap;
      }
    }
    if (history) {
      std::map<std::string, float> custom;
      {
        std::lock_guard<std::mutex> lock(m_paramMutex);
        custom = m_customParameters;
      }
      rcd::Tex none;
      std::memset(&none, 0, sizeof(none));
      if (!pushHistory(out, firstCount, none, custom)) return inputs;
    }
    for (uint32_t k = 0; k < n; ++k) {
      m_frameCount += 1.0f;
      m_time += 0.016f;
    }
    m_lastChunkFrames = n;
    m_lastChunkFirst = f0;
  }
  if (twoLanes) {
    ShaderEngine& o = *m_helper;
    const uint32_t nOther = nFrames - nOwn;
    o.m_frameCount = m_frameCount;   // the helper's frames follow this lane's: FrameCount continues
    o.m_time = m_time;
    o.m_externalOut = outBase + lastPass.frameBytes * nOwn;
    const uint8_t* in = static_cast<const uint8_t*>(inputs) + frameStride * nOwn;
    const void* r = o.applyShaderBatch(in, nOther, width, height, frameStride);
    const bool ok = r == static_cast<const void*>(o.m_externalOut);
    o.m_externalOut = nullptr;
    // whatever happened, this engine's stream continues only after the helper's stream has drained
    const bool joined = hipOk(hipEventRecord(m_laneJoin, m_helperStream), "hipEventRecord") &&
                        hipOk(hipStreamWaitEvent(m_stream, m_laneJoin, 0), "hipStreamWaitEvent");
    if (!ok || !joined) {
      RC_LOG_ERROR("applyShader: the second lane failed");
      return inputs;
    }
    m_frameCount = o.m_frameCount;
    m_time = o.m_time;
  }
  m_outputWidth = lastPass.width;
  m_outputHeight = lastPass.height;
  return m_externalOut ? static_cast<const void*>(m_externalOut) : lastPass.target.ptr;
}

void ShaderEngine::fillGeometry(size_t i, const rcd::Tex& inputTex, const rcd::PassLaunch& L, PassGeometry* geo) const {
  geo->pass_index = (int)i;
  geo->in_w = inputTex.w;
  geo->in_h = inputTex.h;
  geo->out_w = L.out_w;
  geo->out_h = L.out_h;
  geo->out_fmt = L.out_fmt;
  geo->src_w = L.src_w;
  geo->src_h = L.src_h;
  geo->vp_w = L.vp_w;
  geo->vp_h = L.vp_h;
  geo->n_passes = (int)std::min<size_t>(m_passes.size(), 32);
  for (int q = 0; q < geo->n_passes; ++q) {
    geo->chain_w[q] = (int)m_passes[(size_t)q].width;
    geo->chain_h[q] = (int)m_passes[(size_t)q].height;
  }
}

// History push (:1735-1865).  The reference re-draws the final output through pass 0's PROGRAM -
// sampler uniforms as pass 0's last draw left them, unit 0 = the final output (LINEAR, clamp to
// edge: its creation state), every other unit as the frame's last draws left it - into an RGBA8
// texture of the output size, and puts that at the front of the ring.  What Prev<N>Texture later
// samples is therefore not the plain previous output but pass 0 applied to it again.  The size
// uniforms of that draw are pass 0's stale ones, so only kernels that read none are accepted unless
// all sizes coincide.
bool ShaderEngine::pushHistory(const void* finalFrame, int frameCount, const rcd::Tex& sourceTex,
                               const std::map<std::string, float>& custom) {
  ShaderPassData& p0 = m_passes[0];
  const ShaderPassData& lastPass = m_passes.back();
  if (!p0.kernel) return true;
  const KernelEntry& k = *p0.kernel;
  const bool sameSizes = m_passes.size() == 1 && lastPass.width == m_sourceWidth && lastPass.height == m_sourceHeight;
  if (!k.size_independent && !k.stale_size_uniforms && !sameSizes) {
    RC_LOG_ERROR(std::string("frame history: pass 0 kernel '") + k.name +
                 "' reads size uniforms and does not take them from PassLaunch::uni_*; its history re-draw is only supported at "
                 "1:1 single-pass geometry");
    return false;
  }
  HistoryFrame hf;
  const void* recycled = nullptr;
  if (m_frameHistory.size() >= kMaxFrameHistory) {  // reuse the oldest (:1764-1778)
    hf = m_frameHistory.back();
    m_frameHistory.pop_back();
    recycled = hf.buf.ptr;
  }
  const size_t bytes = (size_t)lastPass.width * lastPass.height * 4;
  if (!ensureBuffer(hf.buf, bytes)) return false;
  if (recycled) {
    // The reference clears the recycled texture to (0, 0, 0, 1) (:1797-1798) and draws into it while this frame's
    // binding still has it on a sampler unit (the oldest entry: Prev6Texture) - every fragment of the re-draw reads
    // its own, just cleared, texel (pinned on llvmpipe, tests/golden/motionblur_simple_*_f9).  Here: a constant
    // cleared image of the same size stands in for it on those units.
    if (m_historyCleared.bytes < bytes || m_historyClearedBytes != bytes) {
      if (!ensureBuffer(m_historyCleared, bytes)) return false;
      if (!hipOk(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(m_historyCleared.ptr), (int)0xff000000u, bytes / 4, m_stream), "history clear")) return false;
      m_historyClearedBytes = bytes;
    }
  }
  hf.width = lastPass.width;
  hf.height = lastPass.height;

  rcd::PassLaunch L;
  std::memset(&L, 0, sizeof(L));
  L.srgb_enc = m_srgbEnc;
  L.in = passTexture(m_passes.size() - 1);
  L.in.base = finalFrame;
  L.in.frame_stride = 0;
  L.out = hf.buf.ptr;
  L.out_frame_stride = bytes;
  L.out_w = (int)lastPass.width;
  L.out_h = (int)lastPass.height;
  L.out_fmt = rcd::FMT_RGBA8;
  L.src_w = (int)m_sourceWidth;
  L.src_h = (int)m_sourceHeight;
  L.vp_w = (int)m_viewportWidth;
  L.vp_h = (int)m_viewportHeight;
  L.frame_count0 = frameCount;
  L.n_frames = 1;
  L.flags = (m_undefVaryingZero ? 1 : 0) | (m_generalOnly ? rcd::RC_FLAG_GENERAL_ONLY : 0) | (m_asyncTables ? rcd::RC_FLAG_ASYNC_TABLES : 0);
  if (k.stale_size_uniforms && !sameSizes) {
    // the program's size uniforms are what pass 0's own draw of this frame set (:2401-2437): its input - the source frame -
    // and its output
    L.uni_tex_w = (int)m_sourceWidth;
    L.uni_tex_h = (int)m_sourceHeight;
    L.uni_out_w = (int)p0.width;
    L.uni_out_h = (int)p0.height;
  }
  for (size_t s = 0; s < k.samplers.size() && s < (size_t)rcd::kMaxExtra; ++s) {
    auto it = m_pass0Units.find(k.samplers[s]);
    const int u = it == m_pass0Units.end() ? 0 : it->second;
    L.extra[s] = (u > 0 && u < (int)m_units.size()) ? m_units[(size_t)u] : L.in;
    if (recycled && L.extra[s].base == recycled) {
      // the reference re-specifies the recycled texture at the new size before it clears it (:1768-1798)
      L.extra[s].base = m_historyCleared.ptr;
      L.extra[s].w = (int)lastPass.width;
      L.extra[s].h = (int)lastPass.height;
    }
  }
  for (size_t q = 0; q < k.params.size() && q < (size_t)rcd::kMaxParams; ++q) L.params[q] = effectiveParameter(p0, k.params[q], custom);
  PassGeometry geo;
  fillGeometry(0, L.in, L, &geo);
  if (k.scratch_bytes) {
    if (!ensureBuffer(p0.scratch, k.scratch_bytes(geo))) return false;
    L.scratch = p0.scratch.ptr;
    L.scratch_frame_stride = 0;
  }
  if (k.setup) k.setup(geo, L);
  (void)sourceTex;
  if (!hipOk(k.launch(L, m_stream), "history re-draw")) return false;
  m_frameHistory.insert(m_frameHistory.begin(), hf);
  return true;
}

bool ShaderEngine::runChunk(const void* inputs, uint64_t inStride, uint32_t width, uint32_t height,
                            uint32_t nFrames, int firstFrameCount, void* finalOut) {
  rcd::Tex sourceTex;
  sourceTex.base = inputs;
  sourceTex.frame_stride = inStride;
  sourceTex.w = (int)width;
  sourceTex.h = (int)height;
  sourceTex.fmt = rcd::FMT_RGBX8;  // GL_RGB texture: alpha reads 1.0
  // pass 0 sets the source texture's sampler state when it binds it (:1008-1036)
  sourceTex.linear = m_passes[0].passInfo.filterLinear ? 1 : 0;
  sourceTex.wrap = wrapFromString(m_passes[0].passInfo.wrapMode);

  // mipmap_input0: the reference generates the chain on the source texture as well (ShaderEngine.cpp:1019-1031)
  if (m_passes[0].kernel && m_passes[0].kernel->mip_aware && m_passes[0].passInfo.mipmapInput) {
    int levels = 0;
    size_t frameBytes = 0;
    if (!buildMipLevels(sourceTex, nFrames, &m_sourceMips, &levels, &frameBytes)) return false;
    if (levels > 1) {
      sourceTex.n_levels = levels;
      sourceTex.mip_base = m_sourceMips.ptr;
      sourceTex.mip_frame_stride = frameBytes;
    }
  }
  rcd::Tex current = sourceTex;
  std::map<std::string, float> custom;
  {
    std::lock_guard<std::mutex> lock(m_paramMutex);
    custom = m_customParameters;
  }
  for (size_t i = 0; i < m_passes.size(); ++i) {
    ShaderPassData& pd = m_passes[i];
    const bool last = (i + 1 == m_passes.size());
    // A pass whose size changed (viewport, input size, resolution clamp) gets a new render target in the reference,
    // which also deletes its feedback partner and clears feedbackEnabled (:918-933); the partner is created again,
    // empty, when a program next asks for it - including the lost draw of that frame (bindSamplers).
    if (pd.feedback.ptr && (pd.feedbackWidth != pd.width || pd.feedbackHeight != pd.height || pd.feedbackFormat != pd.format)) {
      if (!hipOk(hipStreamSynchronize(m_stream), "sync")) return false;   // the old partner may still be read by queued kernels
      (void)hipFree(pd.feedback.ptr);
      pd.feedback = DeviceBuffer();
      pd.feedbackEnabled = false;
    }
    void* target = last ? finalOut : pd.target.ptr;
    if (last && pd.feedbackEnabled) {
      // the last pass ping-pongs too: it renders into its own buffer, copied to the output below
      if (!ensureBuffer(pd.lastTarget, pd.frameBytes * nFrames)) return false;
      target = pd.lastTarget.ptr;
    }
    pd.lastWritten = target;
    if (!pd.kernel) {
      // A pass without a program is skipped after its target was cleared to (0,0,0,0); the
      // cleared target is the next pass's input (:959-975).
      if (!hipOk(hipMemsetAsync(target, 0, pd.frameBytes * nFrames, m_stream), "clear")) return false;
    } else {
      rcd::PassLaunch L;
      std::memset(&L, 0, sizeof(L));
      L.srgb_enc = m_srgbEnc;
      L.in = current;
      L.out = target;
      L.out_frame_stride = pd.frameBytes;
      L.out_w = (int)pd.width;
      L.out_h = (int)pd.height;
      L.out_fmt = pd.format;
      L.src_w = (int)m_sourceWidth;
      L.src_h = (int)m_sourceHeight;
      L.vp_w = (int)m_viewportWidth;
      L.vp_h = (int)m_viewportHeight;
      L.frame_count0 = firstFrameCount;
      L.n_frames = (int)nFrames;
      const KernelEntry& k = *pd.kernel;
      bool lostDraw = false;
      if (!bindSamplers(i, k, current, sourceTex, &L, &lostDraw)) return false;
      if (last && pd.feedbackEnabled && target == finalOut) {  // became enabled during this very binding
        if (!ensureBuffer(pd.lastTarget, pd.frameBytes * nFrames)) return false;
        target = pd.lastTarget.ptr;
        pd.lastWritten = target;
        L.out = target;
      }
      if (lostDraw) {
        // the target was cleared (:953-957) and the draw went to framebuffer 0
        if (!hipOk(hipMemsetAsync(target, 0, pd.frameBytes * nFrames, m_stream), "clear")) return false;
        if (last && target != finalOut &&
            !hipOk(hipMemcpyAsync(finalOut, target, pd.frameBytes * nFrames, hipMemcpyDeviceToDevice, m_stream), "copy"))
          return false;
        rcd::Tex nextLost = passTexture(i);
        nextLost.base = target;
        current = nextLost;
        continue;
      }
      for (size_t q = 0; q < k.params.size() && q < (size_t)rcd::kMaxParams; ++q) {
        L.params[q] = effectiveParameter(pd, k.params[q], custom);
      }
      if (pd.passInfo.mipmapInput && k.mip_aware) {
        // (the chain was built when this texture became `current`: GL_LINEAR_MIPMAP_LINEAR with filter_linear,
        // GL_NEAREST_MIPMAP_NEAREST without - the samplers in rc_device.h tell them apart by Tex::linear)
      } else if (pd.passInfo.mipmapInput && (current.w != L.out_w || current.h != L.out_h)) {
        RC_LOG_ERROR("pass " + std::to_string(i) + ": mipmap_input with a " + std::to_string(current.w) + "x" + std::to_string(current.h) +
                     " input and a " + std::to_string(L.out_w) + "x" + std::to_string(L.out_h) +
                     " target would sample mip levels above 0, which the HIP shader chain does not build");
        return false;
      }
      if (i == 3 && current.h != L.out_h && !k.size_independent && !k.ignores_texture_height && !k.texture_height_override) {
        // The reference hands pass index 3 TextureSize.y = the TARGET's height whenever that differs from the
        // input's (ShaderEngine.cpp:2418-2421, written for interlacing.glsl) while InputSize and the texture keep
        // the real height.  No registered kernel restates its shader under that mismatch, so refuse rather
        // than render something the reference would not (e.g. crt-royale-ntsc-*.glslp, whose pass 3 is
        // scanlines-vertical-interlacing).
        RC_LOG_ERROR("pass 3 scales its height (" + std::to_string(current.h) + " -> " + std::to_string(L.out_h) +
                     "): the reference's TextureSize.y override for pass index 3 is not restated for this shader");
        return false;
      }
      PassGeometry geo;
      fillGeometry(i, current, L, &geo);
      L.flags = (m_undefVaryingZero ? 1 : 0) | (m_generalOnly ? rcd::RC_FLAG_GENERAL_ONLY : 0) | (m_asyncTables ? rcd::RC_FLAG_ASYNC_TABLES : 0);
      if (k.scratch_bytes) {
        const uint64_t per_frame = k.scratch_bytes(geo);
        if (!ensureBuffer(pd.scratch, per_frame * nFrames)) return false;
        L.scratch = pd.scratch.ptr;
        L.scratch_frame_stride = per_frame;
      }
      if (k.setup) k.setup(geo, L);
      if (k.validate) {
        if (const char* why = k.validate(L.params)) {
          RC_LOG_ERROR(why);
          return false;
        }
      }
      if (k.validate_launch) {
        if (const char* why = k.validate_launch(L)) {
          RC_LOG_ERROR(why);
          return false;
        }
      }
      // safety net of consumersTakeDecodeTable: a folded pass's view must only reach a slot whose kernel reads it through the table
      {
        bool ok = !(k.reads_input && L.in.dec && !(k.decode_table_inputs & 1u));
        for (size_t s2 = 0; s2 < k.samplers.size() && s2 < (size_t)rcd::kMaxExtra; ++s2)
          ok = ok && !(L.extra[s2].dec && !(k.decode_table_inputs & (1u << (1 + s2))));
        if (!ok) {
          RC_LOG_ERROR("pass " + std::to_string(i) + ": a folded pass's target reached a sampler that cannot read it");
          return false;
        }
      }
      // A pass that is a byte map of its input (KernelEntry::byte_map) is folded into its consumers: not rendered, they read its
      // input through its decode table
      pd.folded = false;
      pd.deferredPending = false;
      if (m_foldPasses && k.byte_map && !last && !m_generalOnly && !L.in.dec && consumersTakeDecodeTable(i) && ensureBuffer(pd.foldedDec, rcd::kFoldedTableWords * sizeof(float)) &&
          k.byte_map(L, m_stream, static_cast<float*>(pd.foldedDec.ptr))) {
        rcd::Tex view = passTexture(i);   // this pass's format, size and the sampler state its consumer sets
        view.base = L.in.base;
        view.frame_stride = L.in.frame_stride;
        view.n_levels = 0;
        view.mip_base = nullptr;
        view.mip_frame_stride = 0;
        view.dec = static_cast<const float*>(pd.foldedDec.ptr);
        view.alpha_one = 1;
        pd.foldedView = view;
        pd.folded = true;
        pd.deferred = L;
        pd.deferredPending = true;
        pd.invariant = false;
        pd.invariantKey.clear();
        pd.mipLevels = 0;
        if (m_passReadBytes.size() != m_passes.size()) m_passReadBytes.assign(m_passes.size(), 0);
        m_passReadBytes[i] = (uint64_t)L.in.w * L.in.h * texelBytes(L.in.fmt);   // (algorithmic: what the pass would read)
        current = view;
        continue;
      }
      if (!last && !pd.target.ptr) {   // a candidate for folding that is rendered after all: its target, left out by applyShaderBatch
        if (!ensureBuffer(pd.target, pd.frameBytes * std::max(m_chunkCapacity, nFrames))) return false;
        target = pd.target.ptr;
        pd.lastWritten = target;
        L.out = target;
      }
      // Frame-invariant pass (see ShaderPassData::invariant): every texture it samples is shared by all frames
      bool servedFromCache = false;
      std::vector<uint8_t> invariantCandidate;
      pd.invariant = false;
      if (k.frame_invariant && !last && !pd.feedbackEnabled && !m_generalOnly) {
        bool inv = !k.reads_input || L.in.frame_stride == 0;
        for (size_t s2 = 0; s2 < k.samplers.size() && s2 < (size_t)rcd::kMaxExtra; ++s2) inv = inv && L.extra[s2].frame_stride == 0;
        if (inv) {
          pd.invariant = true;
          L.n_frames = 1;
          L.frame_count0 = 0;
          // the key: the launch descriptor, which allocation the target is (not just its address), and how often the
          // frame-invariant passes before this one have been rendered (their targets are what this pass samples)
          invariantCandidate.assign(sizeof(L), 0);
          std::memcpy(invariantCandidate.data(), &L, sizeof(L));
          auto append = [&](uint64_t v) {
            const uint8_t* b = reinterpret_cast<const uint8_t*>(&v);
            invariantCandidate.insert(invariantCandidate.end(), b, b + sizeof(v));
          };
          append(pd.target.gen);
          for (size_t j = 0; j < i; ++j) append(m_passes[j].invariant ? m_passes[j].renderCount : 0);
          servedFromCache = invariantCandidate == pd.invariantKey;
        }
      }
      if (!pd.invariant) pd.invariantKey.clear();
      // algorithmic read bytes per frame: distinct sampled textures, once each
      {
        const void* seen[1 + rcd::kMaxExtra];
        int ns = 0;
        uint64_t rb = 0;
        auto add = [&](const rcd::Tex& t) {
          for (int q = 0; q < ns; ++q)
            if (seen[q] == t.base) return;
          seen[ns++] = t.base;
          rb += (uint64_t)t.w * t.h * texelBytes(t.fmt);
        };
        if (k.reads_input) add(L.in);
        for (size_t s2 = 0; s2 < k.samplers.size() && s2 < (size_t)rcd::kMaxExtra; ++s2) add(L.extra[s2]);
        if (m_passReadBytes.size() != m_passes.size()) m_passReadBytes.assign(m_passes.size(), 0);
        m_passReadBytes[i] = pd.invariant ? 0 : rb;   // a pass rendered once per configuration moves no bytes per frame
      }
      if (servedFromCache) {
        rcd::Tex nextCached = passTexture(i);
        nextCached.base = target;
        current = nextCached;
        continue;
      }
      TimedLaunch tl{i, nFrames, nullptr, nullptr};
      if (m_profiling) {
        if (!hipOk(hipEventCreate(&tl.start), "hipEventCreate") || !hipOk(hipEventCreate(&tl.stop), "hipEventCreate")) return false;
        (void)hipEventRecord(tl.start, m_stream);
      }
      if (log_enabled(LogLevel::Debug))
        RC_LOG_DEBUG("pass " + std::to_string(i) + " " + k.name + ": " + std::to_string(L.n_frames) + " frame(s) " + std::to_string(L.out_w) + "x" + std::to_string(L.out_h));
      if (!hipOk(k.launch(L, m_stream), k.name)) {
        pd.invariantKey.clear();
        return false;
      }
      ++pd.renderCount;
      if (pd.invariant) pd.invariantKey.swap(invariantCandidate);   // only a pass that was launched counts as rendered
      if (m_profiling) {
        (void)hipEventRecord(tl.stop, m_stream);
        m_timed.push_back(tl);
      }
    }
    if (last && target != finalOut &&
        !hipOk(hipMemcpyAsync(finalOut, target, pd.frameBytes * nFrames, hipMemcpyDeviceToDevice, m_stream), "copy"))
      return false;
    // mipmap_input of the next pass: the reference generates the chain when that pass binds this texture
    pd.mipLevels = 0;
    if (!last && m_passes[i + 1].kernel && m_passes[i + 1].kernel->mip_aware && m_passes[i + 1].passInfo.mipmapInput) {
      if (!buildMipChain(i, target, nFrames)) return false;
    }
    // this pass's output becomes the next pass's input
    rcd::Tex next = passTexture(i);
    next.base = target;
    current = next;
  }
  return true;
}

bool ShaderEngine::readHistory(size_t k, uint32_t* width, uint32_t* height, void* host, size_t bytes) {
  if (k >= m_frameHistory.size()) return false;
  const HistoryFrame& h = m_frameHistory[k];
  if (width) *width = h.width;
  if (height) *height = h.height;
  if (!host) return true;
  const size_t need = (size_t)h.width * h.height * 4;
  if (bytes < need) return false;
  if (!hipOk(hipStreamSynchronize(m_stream), "sync")) return false;
  return hipOk(hipMemcpy(host, h.buf.ptr, need, hipMemcpyDeviceToHost), "history readback");
}

void ShaderEngine::setProfiling(bool on) {
  m_profiling = on;
  if (on) m_profile.assign(m_passes.size(), PassProfile());
}

bool ShaderEngine::collectProfile(std::vector<PassProfile>* out) {
  if (!hipOk(hipStreamSynchronize(m_stream), "sync")) return false;
  if (m_profile.size() != m_passes.size()) m_profile.assign(m_passes.size(), PassProfile());
  for (auto& t : m_timed) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess && t.pass < m_profile.size()) {
      m_profile[t.pass].totalMs += ms;
      m_profile[t.pass].launches += 1;
      m_profile[t.pass].frames += t.frames;
    }
    (void)hipEventDestroy(t.start);
    (void)hipEventDestroy(t.stop);
  }
  m_timed.clear();
  if (out) *out = m_profile;
  return true;
}

void ShaderEngine::passBytes(size_t i, uint64_t* readBytes, uint64_t* writeBytes) const {
  if (readBytes) *readBytes = i < m_passReadBytes.size() ? m_passReadBytes[i] : 0;
  if (writeBytes) *writeBytes = i < m_passes.size() && m_passes[i].kernel && !m_passes[i].invariant ? m_passes[i].frameBytes : 0;
}

bool ShaderEngine::readPass(size_t i, uint32_t frame, void* host, size_t bytes) {
  if (i >= m_passes.size() || !host) return false;
  ShaderPassData& pd = m_passes[i];
  if (bytes < pd.frameBytes) return false;
  if (pd.folded && pd.deferredPending) {
    // the pass was folded into its consumers: render it now, from the input frames of the last call (still the caller's to keep)
    if (!ensureBuffer(pd.target, pd.frameBytes * std::max<uint32_t>(m_chunkCapacity, (uint32_t)pd.deferred.n_frames))) return false;
    pd.deferred.out = pd.target.ptr;
    if (!hipOk(pd.kernel->launch(pd.deferred, m_stream), pd.kernel->name)) return false;
    pd.lastWritten = pd.target.ptr;
    pd.deferredPending = false;
  }
  if (!pd.target.ptr) return false;
  const bool last = (i + 1 == m_passes.size());
  // two lanes: the intermediates of the batch's second half are the helper instance's
  if (!last && m_lastTwoLanes && m_helper && frame >= m_lastOwnFrames && !pd.invariant) return m_helper->readPass(i, frame - m_lastOwnFrames, host, bytes);
  // intermediates hold the last chunk only; the last pass holds the whole batch
  uint64_t index = frame;
  if (pd.invariant) {
    index = 0;
  } else if (!last) {
    if (frame < m_lastChunkFirst || frame >= m_lastChunkFirst + m_lastChunkFrames) return false;
    index = frame - m_lastChunkFirst;
  }
  if (!hipOk(hipStreamSynchronize(m_stream), "sync")) return false;
  const void* base = (!last && pd.lastWritten) ? pd.lastWritten : pd.target.ptr;
  return hipOk(hipMemcpy(host, static_cast<const uint8_t*>(base) + pd.frameBytes * index, pd.frameBytes, hipMemcpyDeviceToHost),
               "readPass");
}

}  // namespace rc
