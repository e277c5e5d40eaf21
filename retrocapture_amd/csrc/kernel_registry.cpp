#include "kernel_registry.h"

#include "royale_setup.h"
#include "varying.h"

namespace rc {
namespace {

void setupTexCoord(const PassGeometry& g, rcd::PassLaunch& L) {
  L.plane[0] = planeU(1.0f, g.out_w, g.out_h, g.out_fmt);
  L.plane[1] = planeV(1.0f, g.out_w, g.out_h, g.out_fmt);
}

void setupCrtPi(const PassGeometry& g, rcd::PassLaunch& L) {
  // VS: TEX0 = TexCoord * 1.0001 (crt-pi.glsl:101)
  L.plane[0] = planeU(1.0001f, g.out_w, g.out_h, g.out_fmt);
  L.plane[1] = planeV(1.0001f, g.out_w, g.out_h, g.out_fmt);
}

std::vector<KernelEntry> build() {
  std::vector<KernelEntry> r;
  r.push_back({"stock.glsl", "stock", {}, {}, rck::launch_stock, setupTexCoord, false});
  r.push_back({"scanlines/shaders/scanline.glsl", "scanline",
               {{"SCANLINE_BASE_BRIGHTNESS", 0.95f, 0.0f, 1.0f, 0.01f, "Scanline Base Brightness"},
                {"SCANLINE_SINE_COMP_A", 0.0f, 0.0f, 1.0f, 0.02f, "Grid Strength"},
                {"SCANLINE_SINE_COMP_B", 0.25f, 0.0f, 1.0f, 0.05f, "Scanline Strength"},
                {"size", 1.0f, 1.0f, 2.0f, 1.0f, "Grid size"}},
               {}, rck::launch_scanline, setupTexCoord, false});
  r.push_back({"crt/shaders/crt-pi.glsl", "crt-pi",
               {{"CURVATURE_X", 0.10f, 0.0f, 1.0f, 0.01f, "Screen curvature - horizontal"},
                {"CURVATURE_Y", 0.15f, 0.0f, 1.0f, 0.01f, "Screen curvature - vertical"},
                {"MASK_BRIGHTNESS", 0.70f, 0.0f, 1.0f, 0.01f, "Mask brightness"},
                {"SCANLINE_WEIGHT", 6.0f, 0.0f, 15.0f, 0.1f, "Scanline weight"},
                {"SCANLINE_GAP_BRIGHTNESS", 0.12f, 0.0f, 1.0f, 0.01f, "Scanline gap brightness"},
                {"BLOOM_FACTOR", 1.5f, 0.0f, 5.0f, 0.01f, "Bloom factor"},
                {"INPUT_GAMMA", 2.4f, 0.0f, 5.0f, 0.01f, "Input gamma"},
                {"OUTPUT_GAMMA", 2.2f, 0.0f, 5.0f, 0.01f, "Output gamma"}},
               {}, rck::launch_crt_pi, setupCrtPi, false});
  registerRoyaleKernels(r);
  return r;
}

}  // namespace

const std::vector<KernelEntry>& allKernels() {
  static const std::vector<KernelEntry> k = build();
  return k;
}

std::string shaderIdentity(const std::string& shaderPath) {
  const std::string marker = "shaders_glsl/";
  size_t k = shaderPath.rfind(marker);
  if (k != std::string::npos) return shaderPath.substr(k + marker.size());
  // last two components, e.g. ".../shaders/crt-pi.glsl"
  size_t a = shaderPath.find_last_of('/');
  if (a == std::string::npos) return shaderPath;
  size_t b = a == 0 ? std::string::npos : shaderPath.find_last_of('/', a - 1);
  return b == std::string::npos ? shaderPath : shaderPath.substr(b + 1);
}

const KernelEntry* findKernel(const std::string& shaderPath) {
  const std::string id = shaderIdentity(shaderPath);
  const KernelEntry* tail_match = nullptr;
  for (const KernelEntry& e : allKernels()) {
    const std::string want = e.identity;
    if (id == want) return &e;
    // identity given relative to something deeper (e.g. "shaders/crt-pi.glsl" for
    // "crt/shaders/crt-pi.glsl"): accept a suffix match on whole path components
    if (want.size() > id.size() && want.compare(want.size() - id.size(), id.size(), id) == 0 &&
        want[want.size() - id.size() - 1] == '/')
      tail_match = &e;
    if (id.size() > want.size() && id.compare(id.size() - want.size(), want.size(), want) == 0 &&
        id[id.size() - want.size() - 1] == '/')
      tail_match = &e;
  }
  return tail_match;
}

}  // namespace rc
