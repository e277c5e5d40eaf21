// RetroArch .glslp preset model and parser for the HIP shader chain.
//
// Behavioural contract: reference src/shader/ShaderPreset.{h,cpp} (ShaderPreset.h:7-29 for
// the field defaults, ShaderPreset.cpp:18-333 for load/parseLine, :335-538 for path
// resolution, :557-661 for saveAs).  The reference's parsing quirks change which sampler
// state and target sizes a preset gets, so they are kept (see shader_preset.cpp).
#pragma once
#include <string>
#include <unordered_map>
#include <vector>

namespace rc {

struct ShaderPass {
  std::string shaderPath;
  bool filterLinear = true;
  std::string wrapMode = "clamp_to_edge";
  bool mipmapInput = false;
  std::string alias;
  bool floatFramebuffer = false;
  bool srgbFramebuffer = false;
  unsigned int frameCountMod = 0;
  std::string scaleTypeX = "source";  // "source" | "viewport" | "absolute"
  float scaleX = 1.0f;
  std::string scaleTypeY = "source";
  float scaleY = 1.0f;
};

struct ShaderTexture {
  std::string path;
  std::string wrapMode = "clamp_to_border";
  bool mipmap = false;
  bool linear = true;
};

class ShaderPreset {
 public:
  bool load(const std::string& presetPath);
  bool save(const std::string& presetPath,
            const std::unordered_map<std::string, float>& customParameters = {}) const;
  bool saveAs(const std::string& presetPath,
              const std::unordered_map<std::string, float>& customParameters = {}) const;

  const std::vector<ShaderPass>& getPasses() const { return m_passes; }
  const std::unordered_map<std::string, ShaderTexture>& getTextures() const { return m_textures; }
  const std::unordered_map<std::string, float>& getParameters() const { return m_parameters; }
  std::string getBasePath() const { return m_basePath; }
  std::string getPresetPath() const { return m_presetPath; }
  void setParameter(const std::string& name, float value) { m_parameters[name] = value; }
  void clear();

  // Root that "shaders/..." and "../..." style paths fall back to: $RETROCAPTURE_SHADER_PATH,
  // else <assets>/shaders/shaders_glsl (reference ShaderPreset.cpp:353-362, Paths.cpp:150-169).
  static std::string shaderRoot();

 private:
  std::vector<ShaderPass> m_passes;
  std::unordered_map<std::string, ShaderTexture> m_textures;
  std::unordered_map<std::string, float> m_parameters;
  std::string m_basePath;
  std::string m_presetPath;

  void parseLine(const std::string& line);
  std::string resolvePath(const std::string& path) const;
};

}  // namespace rc
