// Registry of hand-written HIP pass kernels, keyed by shader identity.
//
// There is no GLSL compiler in this engine: a preset pass can run only if its shader file
// is one the registry knows (identity = the path below ".../shaders_glsl/", falling back to
// the last two path components).  Each entry carries what the reference would have learnt
// from compiling the GLSL: which samplers the shader declares besides its input
// (reference ShaderEngine.cpp:1049-1415 binds them by uniform name) and the #pragma
// parameter table (used when the .glsl file itself is not readable, e.g. on a machine that
// has presets but not the shader sources).
#pragma once
#include <string>
#include <vector>

#include "kernels/pass_launch.h"
#include "pragma_params.h"

namespace rc {

struct PassGeometry {  // what a kernel's setup hook may look at
  int pass_index;
  int in_w, in_h;    // TextureSize == InputSize (reference ShaderEngine.cpp:2401-2437)
  int out_w, out_h, out_fmt;
  int src_w, src_h;  // OriginalSize
  int vp_w, vp_h;
  int n_passes;                 // output size of every pass of the chain (PassPrev<N> sizes)
  int chain_w[32], chain_h[32];
};

struct KernelParam {
  const char* name;
  float def, min, max, step;
  const char* description;
};

struct KernelEntry {
  const char* identity;                 // e.g. "crt/shaders/crt-pi.glsl"
  const char* name;                     // short name, also used by rc_engine_pass_kernel()
  std::vector<KernelParam> params;      // in the order the kernel expects them in PassLaunch::params
  std::vector<const char*> samplers;    // extra sampler uniform names, in PassLaunch::extra order
  rck::LaunchFn launch;
  void (*setup)(const PassGeometry& g, rcd::PassLaunch& L);  // planes + derived constants
  bool frame_invariant;                 // the shader reads nothing that changes from frame to frame (FrameCount, ...): with
                                        // frame-invariant inputs its output is rendered once and kept (shader_engine.cpp runChunk)
  bool reads_input = true;              // false: the shader never samples its `Texture` input
  // Optional: returns an error text if the kernel cannot honour these parameter values
  const char* (*validate)(const float* params) = nullptr;
  // Optional: bytes of device scratch the pass needs per frame (handed over in PassLaunch::scratch)
  uint64_t (*scratch_bytes)(const PassGeometry& g) = nullptr;
  // The shader reads no size uniform (TextureSize / InputSize / OutputSize / ...): its result depends
  // only on the textures bound and the target size.  Needed for the frame-history re-draw, which runs
  // pass 0's program with stale size uniforms (shader_engine.cpp pushHistory).
  bool size_independent = false;
  // ... or it reads them through PassLaunch::uni_* (rc_device.h), which the re-draw fills with pass 0's stale values.
  bool stale_size_uniforms = false;
  // The kernel samples its input as llvmpipe samples a mip-mapped texture (mipmap_input of its pass:
  // GL_LINEAR_MIPMAP_LINEAR + glGenerateMipmap, ShaderEngine.cpp:1022-1033); the engine then builds the chain.
  bool mip_aware = false;
  // The shader never reads TextureSize.y, so the reference's override of that uniform for pass index 3
  // (ShaderEngine.cpp:2418-2421) cannot change its result.
  bool ignores_texture_height = false;
  // ... or the kernel restates its shader under that override (its setup reads PassGeometry::pass_index).
  bool texture_height_override = false;
  // validate with the whole launch in view (input texture state, target format): a reason, or nullptr to go ahead
  const char* (*validate_launch)(const rcd::PassLaunch& L) = nullptr;
  // Folding a pass into its consumers (shader_engine.cpp runChunk).  Producer side: returns true if, for this launch, the pass
  // stores a per-channel byte map of its input's texel under each pixel with alpha 255 into an sRGB8 target, and fills
  // d_dec256[byte] with the target's decode of the mapped byte and d_dec256[256 + byte] with the mapped byte (rcd::kFoldedTableWords
  // words in device memory, written on `s`).
  bool (*byte_map)(const rcd::PassLaunch& L, hipStream_t s, float* d_dec256) = nullptr;
  // Consumer side: which of the kernel's textures may be such a never-written target, read through rcd::Tex::dec instead
  // (bit 0: `Texture`, bit 1 + s: samplers[s]); the kernel samples no other sRGB8 texture.
  uint32_t decode_table_inputs = 0;
  // A single-pass preset of this kernel stays on one lane (ShaderEngine::setLanes): measured, its launches fill the device by
  // themselves and a second lane's only contend with them (crt-pi 124 k frames/s against 118 k, scanline 2.24 M against 2.09 M)
  bool one_lane = false;
};

const KernelEntry* findKernel(const std::string& shaderPath);
const std::vector<KernelEntry>& allKernels();
std::string shaderIdentity(const std::string& shaderPath);

}  // namespace rc
