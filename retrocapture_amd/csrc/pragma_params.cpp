#include "pragma_params.h"

#include <filesystem>
#include <fstream>
#include <regex>
#include <sstream>

#include "rc_log.h"

namespace fs = std::filesystem;

namespace rc {
namespace {
bool slurp(const std::string& path, std::string* out) {
  std::ifstream f(path, std::ios::binary);
  if (!f.is_open()) return false;
  std::stringstream ss;
  ss << f.rdbuf();
  *out = ss.str();
  return true;
}
bool present(const fs::path& p) {
  std::error_code ec;
  return fs::exists(p, ec);
}
}  // namespace

std::string expandIncludes(const std::string& source, const std::string& baseDir, int depth) {
  static const std::regex include_line(R"([ \t]*#include\s+["<]([^">]+)[">].*)");
  if (depth > 32) return source;
  std::string out;
  size_t pos = 0;
  while (pos <= source.size()) {
    size_t eol = source.find('\n', pos);
    std::string line = source.substr(pos, eol == std::string::npos ? std::string::npos : eol - pos);
    std::smatch m;
    if (std::regex_match(line, m, include_line)) {
      const std::string inc = m[1].str();
      std::string full;
      std::error_code ec;
      if (!inc.empty() && inc[0] == '/') {
        full = inc;
      } else {
        const fs::path cwd = fs::current_path(ec);
        if (!baseDir.empty() && present(fs::path(baseDir) / inc)) full = (fs::path(baseDir) / inc).string();
        if (full.empty() && present(cwd / "shaders" / "shaders_slang" / inc)) full = (cwd / "shaders" / "shaders_slang" / inc).string();
        if (full.empty() && present(cwd / inc)) full = (cwd / inc).string();
        if (full.empty() && !baseDir.empty()) {
          fs::path b(baseDir);
          std::string clean = inc;
          while (clean.find("../") == 0) {
            clean = clean.substr(3);
            b = b.parent_path();
          }
          if (present(b / clean)) full = (b / clean).string();
        }
      }
      std::string body;
      if (!full.empty() && slurp(full, &body)) {
        out += expandIncludes(body, fs::path(full).parent_path().string(), depth + 1);
      } else {
        RC_LOG_WARN("Included file not found: " + inc);
      }
    } else {
      out += line;
    }
    if (eol == std::string::npos) break;
    out += '\n';
    pos = eol + 1;
  }
  return out;
}

ShaderSourceInfo scanShaderText(const std::string& text, const std::string& baseDir) {
  ShaderSourceInfo info;
  info.readable = true;
  const std::string src = expandIncludes(text, baseDir);
  static const std::regex pragma(
      "#pragma\\s+parameter\\s+(\\w+)\\s+\"([^\"]*)\"\\s+(-?[\\d.]+)\\s+(-?[\\d.]+)\\s+(-?[\\d.]+)\\s+(-?[\\d.]+)");
  for (auto it = std::sregex_iterator(src.begin(), src.end(), pragma); it != std::sregex_iterator(); ++it) {
    const std::string name = (*it)[1].str();
    if (name.find("bogus_") != std::string::npos) continue;
    ShaderParameterInfo p;
    p.description = (*it)[2].str();
    try {
      p.defaultValue = std::stof((*it)[3].str());
      p.min = std::stof((*it)[4].str());
      p.max = std::stof((*it)[5].str());
      p.step = std::stof((*it)[6].str());
    } catch (...) {
      p = ShaderParameterInfo();
      p.description = (*it)[2].str();
    }
    if (!info.parameterInfo.count(name)) info.declarationOrder.push_back(name);
    info.parameterInfo[name] = p;  // a later duplicate overwrites, as map assignment does in the reference
  }
  info.parameterUniform = !info.parameterInfo.empty();
  return info;
}

ShaderSourceInfo scanShaderSource(const std::string& shaderPath) {
  std::string text;
  if (!slurp(shaderPath, &text)) return ShaderSourceInfo();
  return scanShaderText(text, fs::path(shaderPath).parent_path().string());
}

}  // namespace rc
