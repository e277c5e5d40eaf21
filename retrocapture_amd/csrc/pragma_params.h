// `#pragma parameter` extraction from a RetroArch GLSL file.
// Contract: reference src/shader/ShaderPreprocessor.cpp:11-220 (regex at :36, "bogus_"
// labels skipped at :48, PARAMETER_UNIFORM defined only if the file declares at least one
// parameter at :207-212) and processIncludes :222-363 (only lines that START with #include,
// searched relative to the including file, then <cwd>/shaders/shaders_slang, then <cwd>,
// then with leading "../" stripped against parent directories).
// Everything else that function does is GL compile glue with no meaning for HIP kernels.
#pragma once
#include <map>
#include <string>
#include <vector>

namespace rc {

struct ShaderParameterInfo {  // reference ShaderEngine.h:11-17
  float defaultValue = 0.f;
  float min = 0.f;
  float max = 1.f;
  float step = 0.01f;
  std::string description;
};

struct ShaderSourceInfo {
  bool readable = false;
  std::map<std::string, ShaderParameterInfo> parameterInfo;  // name -> info (sorted, as std::map in the reference)
  std::vector<std::string> declarationOrder;                 // first occurrence order in the text
  bool parameterUniform = false;                             // #define PARAMETER_UNIFORM would be emitted
};

std::string expandIncludes(const std::string& source, const std::string& baseDir, int depth = 0);
ShaderSourceInfo scanShaderSource(const std::string& shaderPath);
ShaderSourceInfo scanShaderText(const std::string& text, const std::string& baseDir);

}  // namespace rc
