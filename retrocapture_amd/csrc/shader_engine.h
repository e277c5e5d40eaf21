// ShaderEngine: the reference's shader-chain API (reference src/shader/ShaderEngine.h:42-98)
// executed by HIP pass kernels on an MI355X instead of OpenGL FBO ping-pong.
//
// Same method names, argument meaning and error behaviour as the reference class; the
// only change of type is that a frame is a device pointer to RGBA8 texels (row 0 first =
// texture t 0, alpha ignored and read as 1.0, like the reference's GL_RGB source texture,
// FrameProcessor.cpp:172-205) instead of a GLuint texture name.  Output frames stay owned by
// the engine and are valid until the next apply, as in the reference (ShaderEngine.cpp:1873).
//
// Additions for throughput: applyShaderBatch() runs N independent frames per call (frame k
// sees FrameCount = count+1+k, exactly what N successive applyShader calls would see);
// frames are the grid's z dimension, processed `chunk` at a time through the whole chain so
// intermediates stay cache resident.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "kernel_registry.h"
#include "pragma_params.h"
#include "shader_preset.h"

namespace rc {

struct DeviceBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
  uint64_t gen = 0;   // allocation generation: changes whenever the buffer is (re)allocated, even at the same address
};

struct ShaderPassData {  // reference ShaderEngine.h:19-40
  ShaderPass passInfo;
  const KernelEntry* kernel = nullptr;  // "program": null = pass failed to "compile"
  uint32_t width = 0, height = 0;
  int format = rcd::FMT_RGBA8;
  DeviceBuffer target;      // [chunk or batch][height][width] texels
  DeviceBuffer scratch;     // kernel-private scratch, [chunk] frames (KernelEntry::scratch_bytes)
  // mip levels 1.. of the target, [chunk] frames, built after the pass when the next pass declares
  // mipmap_input and its kernel samples mip-mapped (glGenerateMipmap at bind time, .cpp:1022-1033)
  DeviceBuffer mips;
  int mipLevels = 0;        // levels of the chain including level 0; 0: none
  size_t mipFrameBytes = 0;
  // PassFeedback ping-pong (reference ShaderEngine.h feedbackTexture / feedbackFramebuffer, .cpp:1285-1347,
  // :1710-1718): the previous frame's output of this pass, allocated when a program first asks for it
  DeviceBuffer feedback;
  bool feedbackEnabled = false;
  int feedbackLinear = 1, feedbackWrap = rcd::WRAP_EDGE;  // sampler state of the partner texture object
  uint32_t feedbackWidth = 0, feedbackHeight = 0;          // the size it was created with (it is dropped when the pass's size changes)
  int feedbackFormat = -1;
  DeviceBuffer lastTarget;  // last pass only, when it takes part in feedback: its own render target
  const void* lastWritten = nullptr;  // where the most recent chunk of this pass was rendered
  // A pass whose shader and inputs do not depend on the frame (crt-royale's two mask-resize passes: LUT -> 64x68 ->
  // 120x68) is rendered once into frame 0 of its target and sampled with frame stride 0 until anything its result
  // depends on changes (sizes, parameters, flags, the textures bound): `invariantKey` is the launch it was rendered with.
  bool invariant = false;
  std::vector<uint8_t> invariantKey;
  uint64_t renderCount = 0;   // how often this pass has actually been rendered (a consumer's key includes its producers' counts)
  size_t frameBytes = 0;
  // A pass folded into its consumers (runChunk, KernelEntry::byte_map): not rendered for the last chunk - its consumers read ITS
  // input through `foldedView` (this pass's format and sampler state over the input's bytes, decoded through `foldedDec`).  The
  // launch is kept, and rendered into `target` if somebody asks for the pass's own bytes (readPass).
  bool folded = false;
  DeviceBuffer foldedDec;       // rcd::kFoldedTableWords words (Tex::dec)
  rcd::Tex foldedView;
  rcd::PassLaunch deferred;
  bool deferredPending = false;
  std::map<std::string, float> extractedParameters;
  std::map<std::string, ShaderParameterInfo> parameterInfo;
};

struct LutTexture {
  DeviceBuffer data;
  int width = 0, height = 0;
};

// Something that holds a pointer to an engine (a FramePipeline): told when the engine shuts down, so that it
// stops using it instead of dereferencing a dead object.
struct EngineClient {
  virtual void engineGone() = 0;
 protected:
  ~EngineClient() = default;
};

class ShaderEngine {
 public:
  ShaderEngine();
  ~ShaderEngine();
  void attachClient(EngineClient* c);
  void detachClient(EngineClient* c);

  // device < 0: keep the current HIP device.  stream may be null (default stream).
  bool init(int device = -1, hipStream_t stream = nullptr);
  void shutdown();

  bool loadShader(const std::string& shaderPath);
  bool loadPreset(const std::string& presetPath);
  std::string getPresetPath() const;

  // One frame.  Returns the device pointer of the output frame (engine owned); returns
  // `input` itself when no shader is active or no pass is usable, nullptr for a null input.
  const void* applyShader(const void* input, uint32_t width, uint32_t height);
  // N frames, `frameStride` bytes apart (0 = tightly packed width*height*4).
  const void* applyShaderBatch(const void* inputs, uint32_t nFrames, uint32_t width, uint32_t height,
                               uint64_t frameStride = 0);

  void setViewport(uint32_t width, uint32_t height);
  void disableShader();
  bool isShaderActive() const { return m_shaderActive; }
  uint32_t getOutputWidth() const { return m_outputWidth; }
  uint32_t getOutputHeight() const { return m_outputHeight; }
  void setMaxShaderResolution(uint32_t maxWidth, uint32_t maxHeight);
  uint32_t getMaxShaderWidth() const { return m_maxShaderWidth; }
  uint32_t getMaxShaderHeight() const { return m_maxShaderHeight; }

  void setUniform(const std::string& name, float value);
  void setUniform(const std::string& name, float x, float y);
  void setUniform(const std::string& name, float x, float y, float z, float w);

  struct ShaderParameter {
    std::string name;
    float value, defaultValue, min, max, step;
    std::string description;
  };
  std::vector<ShaderParameter> getShaderParameters() const;
  bool setShaderParameter(const std::string& name, float value);

  ShaderPreset& getPreset() { return m_preset; }
  const ShaderPreset& getPreset() const { return m_preset; }

  // ---- not in the reference -------------------------------------------------------------
  void setInputFilterLinear(bool linear) { m_inputLinear = linear; }  // FrameProcessor's texture filter
  void setChunkFrames(uint32_t n) {
    m_chunk = n ? n : 1;
    m_chunkAuto = false;
  }
  // Accept presets whose .glsl files are absent when the registry knows the shader (the
  // registry's parameter table is used).  Off by default: the reference fails such a pass.
  void setAllowMissingSources(bool allow) { m_allowMissingSources = allow; }
  // crt-royale pass 6 tests a varying its vertex shader never writes.  false (default): the
  // fragment shader discards everything, as on Mesa llvmpipe; true: the varying reads 0, as on
  // GL drivers that zero undefined varyings, and the resized phosphor mask is rendered.
  void setUndefinedVaryingZero(bool zero) { m_undefVaryingZero = zero; }
  void setGeneralKernelsOnly(bool on) { m_generalOnly = on; }
  // Fold passes that are byte maps of their input into their consumers (default on; crt-royale's pass 0 at 1:1): same bytes in
  // every other pass.  A folded pass's own bytes are rendered on demand by readPass from the input frames of the last call,
  // which must still be there.
  void setFoldPasses(bool on) { m_foldPasses = on; }
  // Per-geometry tables that take long to build (crt-royale's scanline tables: 140 ms) are built on a worker thread while the
  // general kernel forms serve the frames (default); off: the first frame of a new geometry waits for them.
  void setAsyncTableBuilds(bool on) { m_asyncTables = on; }
  bool passFolded(size_t i) const { return i < m_passes.size() && m_passes[i].folded; }
  // float_framebuffer targets stored as four binary16 values (8 bytes per texel) instead of RGBA32F: arithmetic stays
  // float, only the storage of those targets rounds.  Off by default (bit-exact); see DESIGN.md for the tolerance.
  void setFloatTargetFp16(bool on) { m_floatTargetFp16 = on; }
  uint32_t getChunkFrames() const { return m_chunk; }
  // Two lanes (opt-in, default 1): the second half of a batch runs on a helper engine instance with its own HIP stream - same
  // device, preset, parameters and flags - and writes straight into this engine's batch output; the helper's stream waits
  // for this engine's stream before it starts and is waited for before applyShaderBatch returns its pointer, so the caller
  // sees one stream-ordered result.  Kernels of the two lanes overlap on the device (the copy-rate passes of one under the
  // VALU / LDS bound passes of the other).  Presets that sample frame history or PassFeedback, single-shader mode and
  // profiled runs stay on one lane.
  void setLanes(uint32_t n) { m_lanes = n >= 2 ? 2u : 1u; }
  uint32_t getLanes() const { return m_lanes; }
  hipStream_t stream() const { return m_stream; }
  size_t passCount() const { return m_passes.size(); }
  const ShaderPassData* pass(size_t i) const { return i < m_passes.size() ? &m_passes[i] : nullptr; }
  uint64_t outputFrameBytes() const { return (uint64_t)m_outputWidth * m_outputHeight * 4; }
  // Copies pass `i`'s target for frame `frame` of the last processed chunk to host memory.
  bool readPass(size_t i, uint32_t frame, void* host, size_t bytes);
  float frameCount() const { return m_frameCount; }
  // Per-pass device timing with HIP events on the engine's stream (off by default; when on,
  // every pass launch is bracketed by two events that are read back at collectProfile()).
  void setProfiling(bool on);
  struct PassProfile {
    double totalMs = 0.0;   // summed launch durations
    uint64_t launches = 0;  // kernel launches
    uint64_t frames = 0;    // frames those launches processed
  };
  bool collectProfile(std::vector<PassProfile>* out);  // synchronises the stream
  // Algorithmic bytes one frame of pass i moves: each distinct texture it samples once at its
  // stored size, plus its target once (SURVEY.md section 8d convention).
  void passBytes(size_t i, uint64_t* readBytes, uint64_t* writeBytes) const;
  // Frame-history ring (newest first); entries are RGBA8 at the output size they were pushed with.
  size_t historyCount() const { return m_frameHistory.size(); }
  bool readHistory(size_t k, uint32_t* width, uint32_t* height, void* host, size_t bytes);

 private:
  bool m_initialized = false;
  bool m_shaderActive = false;
  int m_device = -1;
  hipStream_t m_stream = nullptr;
  const uint32_t* m_srgbEnc = nullptr;  // sRGB8 encode table of this device (srgb_encode.cpp)

  ShaderPreset m_preset;
  std::string m_presetPath;
  std::vector<ShaderPassData> m_passes;
  std::unordered_map<std::string, LutTexture> m_textureReferences;
  uint32_t m_sourceWidth = 0, m_sourceHeight = 0;
  uint32_t m_viewportWidth = 0, m_viewportHeight = 0;
  uint32_t m_outputWidth = 0, m_outputHeight = 0;
  uint32_t m_maxShaderWidth = 0, m_maxShaderHeight = 0;
  float m_frameCount = 0.0f;
  float m_time = 0.0f;
  bool m_inputLinear = false;
  // frames per launch: per-launch costs (table loads, ramp and tail of the one-workgroup-per-CU kernels) are 25 % of crt-royale
  // at 8, 5 % at 32, 2 % at 64.  Unless the caller sets it, a batch runs 128 frames per launch where 128 frames of the largest
  // pass target stay below 2 GiB (1080p RGBA8 chains), and m_chunk = 64 otherwise (4K targets, float targets).
  uint32_t m_chunk = 64;
  bool m_chunkAuto = true;
  uint32_t m_lanes = 2;
  bool m_lastTwoLanes = false;     // the last batch was split over the two lanes, this engine taking the first m_lastOwnFrames frames
  uint32_t m_lastOwnFrames = 0;
  std::unique_ptr<ShaderEngine> m_helper;   // the second lane (setLanes)
  hipStream_t m_helperStream = nullptr;
  hipEvent_t m_laneFork = nullptr, m_laneJoin = nullptr;
  uint64_t m_configEpoch = 1, m_helperEpoch = 0;   // loadPreset / loadShader bump the epoch: the helper reloads when it lags
  uint8_t* m_externalOut = nullptr;   // on a helper: where the last pass of its next batch goes (the owner's batch output)
  bool syncHelper();
  void destroyHelper();
  uint32_t m_lastChunkFrames = 0;
  uint32_t m_lastChunkFirst = 0;
  bool m_singleShader = false;
  bool m_allowMissingSources = false;
  bool m_undefVaryingZero = false;
  bool m_generalOnly = false;
  bool m_foldPasses = true;
  bool m_asyncTables = true;
  uint32_t m_chunkCapacity = 0;   // frames the intermediate targets are sized for (applyShaderBatch)
  bool consumersTakeDecodeTable(size_t i) const;
  bool m_floatTargetFp16 = false;
  struct Vec4 { float x, y, z, w; };
  std::unordered_map<std::string, Vec4> m_uniforms;
  DeviceBuffer m_batchOutput;
  bool m_profiling = false;
  struct TimedLaunch { size_t pass; uint32_t frames; hipEvent_t start, stop; };
  std::vector<TimedLaunch> m_timed;
  std::vector<PassProfile> m_profile;
  std::vector<uint64_t> m_passReadBytes;

  std::vector<EngineClient*> m_clients;
  mutable std::mutex m_paramMutex;  // the reference shares these maps across threads unguarded
  std::map<std::string, float> m_customParameters;

  int targetFormat(const ShaderPass& pi) const;
  bool loadPresetPasses();
  bool compilePass(size_t i);
  void cleanupPresetPasses();
  void cleanupTextureReferences();
  bool loadTextureReference(const std::string& name, const std::string& path);
  uint32_t calculateScale(uint32_t sourceSize, const std::string& scaleType, float scale, uint32_t viewportSize) const;
  void resolvePassSizes(uint32_t width, uint32_t height);
  bool ensureBuffer(DeviceBuffer& b, size_t bytes);
  float effectiveParameter(const ShaderPassData& pass, const KernelParam& kp,
                           const std::map<std::string, float>& custom) const;
  // frame history (reference ShaderEngine.h:140-143: at most 7 textures of the processed output,
  // newest first), the texture bound to every texture unit as the last draw left it, and the
  // sampler uniforms of pass 0's program (a uniform keeps its value until it is set again)
  struct HistoryFrame { DeviceBuffer buf; uint32_t width = 0, height = 0; };
  std::vector<HistoryFrame> m_frameHistory;
  std::vector<rcd::Tex> m_units;
  std::map<std::string, int> m_pass0Units;
  static constexpr size_t kMaxFrameHistory = 7;
  bool presetSamplesHistory() const;
  bool pushHistory(const void* finalFrame, int frameCount, const rcd::Tex& sourceTex,
                   const std::map<std::string, float>& custom);
  void fillGeometry(size_t passIndex, const rcd::Tex& inputTex, const rcd::PassLaunch& L, PassGeometry* geo) const;
  rcd::Tex lutTexture(const std::string& name) const;
  // Emulates the reference's sampler binding of pass i (:1095-1415): same order, same texture-unit
  // numbering; fills L.extra for the kernel's declared samplers.
  // Returns false on a device error.  *lostDraw: a feedback partner was created during this binding,
  // which in the reference leaves framebuffer 0 bound - the pass's draw misses its (cleared) target.
  bool bindSamplers(size_t passIndex, const KernelEntry& k, const rcd::Tex& inputTex, const rcd::Tex& sourceTex,
                    rcd::PassLaunch* L, bool* lostDraw);
  bool presetSamplesFeedback() const;
  rcd::Tex passTexture(size_t passIndex) const;
  bool buildMipChain(size_t passIndex, const void* level0, uint32_t nFrames);
  // levels 1.. of the chain of `level0` (fmt RGBX8 = the GL_RGB source frame) into `mips`, packed per frame
  bool buildMipLevels(const rcd::Tex& level0, uint32_t nFrames, DeviceBuffer* mips, int* levels, size_t* frameBytes);
  DeviceBuffer m_sourceMips;   // mipmap_input0: the chain of the source frames of the current chunk
  DeviceBuffer m_historyCleared;   // (0, 0, 0, 1) image: what a recycled history texture reads as during its own re-draw
  size_t m_historyClearedBytes = 0;
  bool runChunk(const void* inputs, uint64_t inStride, uint32_t width, uint32_t height, uint32_t nFrames,
                int firstFrameCount, void* finalOut);
};

}  // namespace rc
