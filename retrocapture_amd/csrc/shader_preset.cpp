// .glslp preset parser. See shader_preset.h for the reference contract.
//
// Quirks of the reference parser that are deliberately reproduced because they decide
// sampler state / pass count (reference ShaderPreset.cpp):
//   Q1  a value is the text after the first '=', trimmed of blank/tab/quote at both ends
//       only; booleans are true iff that text lower-cased is exactly "true" or "1"
//       (:129-130, :197) - so `"true" # comment` is FALSE; numbers go through stof, which
//       accepts a numeric prefix.
//   Q2  any key containing a decimal digit is a per-pass key whose index is stoi() of the
//       key from its first digit; the pass list GROWS to fit (:176-183); keys whose
//       prefix is not recognised are dropped - so `frame_count_mod0` is ignored and global
//       parameters with a digit in their name are lost.
//   Q3  prefix matching is by `find(prefix) == 0` in a fixed order, e.g. `scale_type_x`
//       before `scale_x` before `scale_type` before `scale` (:186-259).
//   Q4  texture keys (`<name>`, `<name>_linear|_wrap_mode|_mipmap`) are recognised first, but
//       only for names already declared by a preceding `textures =` line (:134-171).
#include "shader_preset.h"

#include <algorithm>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <sstream>

#include "rc_log.h"

namespace fs = std::filesystem;

namespace rc {
namespace {

std::string trimmed(const std::string& s, const char* chars) {
  size_t b = s.find_first_not_of(chars);
  if (b == std::string::npos) return std::string();
  size_t e = s.find_last_not_of(chars);
  return s.substr(b, e - b + 1);
}

bool truthy(std::string v) {
  std::transform(v.begin(), v.end(), v.begin(), [](unsigned char c) { return (char)std::tolower(c); });
  return v == "true" || v == "1";
}

float to_float(const std::string& v) {
  try {
    return std::stof(v);
  } catch (...) {
    return 0.0f;
  }
}

bool starts(const std::string& key, const char* prefix) { return key.find(prefix) == 0; }

bool ends_with(const std::string& key, const std::string& suffix, std::string* base) {
  if (key.size() < suffix.size()) return false;
  // the reference uses rfind(suffix) and requires it to end the key; an empty base is "no match"
  size_t pos = key.rfind(suffix);
  if (pos == std::string::npos || pos + suffix.size() != key.size()) return false;
  *base = key.substr(0, pos);
  return !base->empty();
}

bool pathExists(const fs::path& p) {
  std::error_code ec;
  return fs::exists(p, ec);
}

}  // namespace

void ShaderPreset::clear() {
  m_passes.clear();
  m_textures.clear();
  m_parameters.clear();
}

std::string ShaderPreset::shaderRoot() {
  if (const char* env = std::getenv("RETROCAPTURE_SHADER_PATH"))
    if (pathExists(env)) return env;
  // Paths::getReadOnlyAssetsDir (reference Paths.cpp:150-260, Linux branch)
  fs::path assets;
  const char* over = std::getenv("RETROCAPTURE_ASSETS_DIR");
  std::error_code ec;
  if (over && *over && pathExists(over)) {
    assets = over;
  } else if (pathExists(fs::current_path(ec) / "shaders" / "shaders_glsl")) {
    assets = fs::current_path(ec);
  } else {
    const char* xdg = std::getenv("XDG_DATA_DIRS");
    std::stringstream ss(xdg && *xdg ? xdg : "/usr/local/share:/usr/share");
    std::string entry;
    while (assets.empty() && std::getline(ss, entry, ':'))
      if (!entry.empty() && pathExists(fs::path(entry) / "retrocapture")) assets = fs::path(entry) / "retrocapture";
    if (assets.empty()) {
      fs::path exe = fs::read_symlink("/proc/self/exe", ec).parent_path();
      if (pathExists(exe.parent_path() / "share" / "retrocapture")) assets = exe.parent_path() / "share" / "retrocapture";
      else assets = exe / "assets";
    }
  }
  return (assets / "shaders" / "shaders_glsl").string();
}

bool ShaderPreset::load(const std::string& presetPath) {
  clear();
  fs::path path(presetPath);
  std::error_code ec;
  if (path.is_relative()) path = fs::absolute(path, ec);
  m_basePath = path.parent_path().string();
  if (m_basePath.empty()) m_basePath = fs::current_path(ec).string();
  m_presetPath = path.string();

  std::ifstream file(presetPath);
  if (!file.is_open()) {
    RC_LOG_ERROR("Failed to open preset: " + presetPath);
    return false;
  }
  std::string line;
  while (std::getline(file, line)) {
    line = trimmed(line, " \t\r\n");
    if (line.empty() || line[0] == '#') continue;
    // `shaders = N` and `textures = A;B` are matched literally with the single blank
    // (reference :60, :75); `shaders=3` falls through to the generic key path.
    if (line.find("shaders =") == 0) {
      std::string v = trimmed(line.substr(line.find('=') + 1), " \t\"");
      m_passes.resize((size_t)std::max(0, (int)to_float(v)));
      continue;
    }
    if (line.find("textures =") == 0) {
      std::string v = trimmed(line.substr(line.find('=') + 1), " \t\"");
      std::istringstream names(v);
      std::string name;
      while (std::getline(names, name, ';')) {
        name = trimmed(name, " \t\"");
        if (!name.empty()) m_textures[name] = ShaderTexture();
      }
      continue;
    }
    parseLine(line);
  }
  RC_LOG_INFO("Preset parsed: " + std::to_string(m_passes.size()) + " passes, " +
              std::to_string(m_textures.size()) + " textures");
  return !m_passes.empty();
}

void ShaderPreset::parseLine(const std::string& line) {
  size_t eq = line.find('=');
  if (eq == std::string::npos) return;
  const std::string key = trimmed(line.substr(0, eq), " \t");
  const std::string value = trimmed(line.substr(eq + 1), " \t\"");  // Q1

  // Q4: declared textures first.
  std::string base;
  if (ends_with(key, "_linear", &base) && m_textures.count(base)) {
    m_textures[base].linear = truthy(value);
    return;
  }
  if (ends_with(key, "_wrap_mode", &base) && m_textures.count(base)) {
    m_textures[base].wrapMode = value;
    return;
  }
  if (ends_with(key, "_mipmap", &base) && m_textures.count(base)) {
    m_textures[base].mipmap = truthy(value);
    return;
  }
  if (m_textures.count(key)) {
    m_textures[key].path = resolvePath(value);
    return;
  }

  size_t digit = key.find_first_of("0123456789");
  if (digit != std::string::npos) {  // Q2
    int idx = std::stoi(key.substr(digit));
    if (idx >= (int)m_passes.size()) m_passes.resize((size_t)idx + 1);
    ShaderPass& p = m_passes[(size_t)idx];
    // Q3: order matters.
    if (starts(key, "shader")) p.shaderPath = resolvePath(value);
    else if (starts(key, "filter_linear")) p.filterLinear = truthy(value);
    else if (starts(key, "wrap_mode")) p.wrapMode = value;
    else if (starts(key, "mipmap_input")) p.mipmapInput = truthy(value);
    else if (starts(key, "alias")) p.alias = value;
    else if (starts(key, "float_framebuffer")) p.floatFramebuffer = truthy(value);
    else if (starts(key, "srgb_framebuffer")) p.srgbFramebuffer = truthy(value);
    else if (starts(key, "scale_type_x")) p.scaleTypeX = value;
    else if (starts(key, "scale_x")) p.scaleX = to_float(value);
    else if (starts(key, "scale_type_y")) p.scaleTypeY = value;
    else if (starts(key, "scale_y")) p.scaleY = to_float(value);
    else if (starts(key, "scale_type")) p.scaleTypeX = p.scaleTypeY = value;
    else if (starts(key, "scale")) p.scaleX = p.scaleY = to_float(value);
    // anything else with a digit (frame_count_mod0, texture_wrap_mode11, param names with
    // digits ...) is dropped, having possibly grown the pass list.
    return;
  }

  // digit-free keys: late texture forms, else a global float parameter
  const bool is_sampler = starts(key, "Sampler");
  const bool has_wrap = key.find("_wrap_mode") != std::string::npos;
  const bool has_mip = key.find("_mipmap") != std::string::npos;
  if (is_sampler && !has_wrap && !has_mip) {
    ShaderTexture t;
    t.path = resolvePath(value);
    m_textures[key] = t;
  } else if (is_sampler && has_wrap) {
    std::string name = key.substr(0, key.find("_wrap_mode"));
    if (m_textures.count(name)) m_textures[name].wrapMode = value;
  } else if (is_sampler && has_mip) {
    std::string name = key.substr(0, key.find("_mipmap"));
    if (m_textures.count(name)) m_textures[name].mipmap = truthy(value);
  } else if (key.find("_linear") != std::string::npos) {
    std::string name = key.substr(0, key.find("_linear"));
    if (m_textures.count(name)) m_textures[name].linear = truthy(value);
  } else if (starts(key, "frame_count_mod")) {
    // unreachable with a digit in the key; without one there is no pass to apply it to
  } else {
    m_parameters[key] = to_float(value);
  }
}

// Path search order of the reference (ShaderPreset.cpp:335-538).
std::string ShaderPreset::resolvePath(const std::string& path) const {
  if (path.empty() || path[0] == '/') return path;
  std::error_code ec;
  const fs::path cwd = fs::current_path(ec);
  const fs::path root(shaderRoot());
  const fs::path base(m_basePath);

  fs::path candidate = (base / path).lexically_normal();
  if (pathExists(candidate)) return candidate.string();

  if (path.find("shaders/") == 0) {
    const std::string sub = path.substr(8);
    if (pathExists(base / sub)) return (base / sub).string();
    if (pathExists(root / sub)) return (root / sub).string();
  }

  std::string clean = path;
  int ups = 0;
  while (clean.find("../") == 0) {
    clean = clean.substr(3);
    ++ups;
  }
  if (ups > 0) {
    candidate = (root / clean).lexically_normal();
    if (pathExists(candidate)) return candidate.string();
    // same file name anywhere below the named directory
    size_t slash = clean.find_last_of('/');
    if (slash != std::string::npos) {
      const fs::path dir = root / clean.substr(0, slash);
      const std::string file = clean.substr(slash + 1);
      if (pathExists(dir) && fs::is_directory(dir, ec)) {
        for (fs::recursive_directory_iterator it(dir, ec), end; !ec && it != end; it.increment(ec))
          if (fs::is_regular_file(*it, ec) && it->path().filename() == file) return it->path().string();
      }
    }
    fs::path b = base.is_relative() ? cwd / base : base;
    const std::string bs = b.string();
    size_t k = bs.find("shaders_glsl");
    if (k != std::string::npos) {
      // NB: the reference keeps only 11 of the 12 characters of "shaders_glsl" here
      // (substr(0, pos + 11)); kept so a lookup that fails there fails here too.
      candidate = (fs::path(bs.substr(0, k + 11)) / clean).lexically_normal();
      if (pathExists(candidate)) return candidate.string();
    }
    for (int i = 0; i < ups; ++i) b = b.parent_path();
    candidate = (b / clean).lexically_normal();
    if (pathExists(candidate)) return candidate.string();
  }

  candidate = (cwd / path).lexically_normal();
  if (pathExists(candidate)) return candidate.string();
  if (pathExists(root / clean)) return (root / clean).string();
  if (!note_missing_source()) RC_LOG_WARN("Shader not found: " + path + " (tried: " + candidate.string() + ")");
  return candidate.string();
}

bool ShaderPreset::save(const std::string& presetPath,
                        const std::unordered_map<std::string, float>& customParameters) const {
  return saveAs(presetPath, customParameters);
}

// Rewrites the loaded preset line by line, replacing the value of every line whose key is
// a known global parameter or a custom parameter (reference ShaderPreset.cpp:557-661).
bool ShaderPreset::saveAs(const std::string& presetPath,
                          const std::unordered_map<std::string, float>& customParameters) const {
  if (m_presetPath.empty()) {
    RC_LOG_ERROR("No preset loaded to save");
    return false;
  }
  std::ifstream in(m_presetPath);
  if (!in.is_open()) {
    RC_LOG_ERROR("Failed to open original preset for reading: " + m_presetPath);
    return false;
  }
  std::vector<std::string> lines;
  for (std::string l; std::getline(in, l);) lines.push_back(l);
  in.close();

  std::unordered_map<std::string, float> values = m_parameters;
  for (const auto& kv : customParameters) values[kv.first] = kv.second;

  std::ofstream out(presetPath);
  if (!out.is_open()) {
    RC_LOG_ERROR("Failed to create preset file: " + presetPath);
    return false;
  }
  for (const std::string& original : lines) {
    std::string lineOut = original;
    size_t eq = original.find('=');
    if (eq != std::string::npos) {
      const std::string key = trimmed(original.substr(0, eq), " \t");
      auto it = values.find(key);
      if (it != values.end()) {
        std::string num = std::to_string(it->second);  // "%f", then strip trailing zeros
        size_t dot = num.find('.');
        if (dot != std::string::npos) {
          while (num.size() > dot + 1 && num.back() == '0') num.pop_back();
          if (num.back() == '.') num.pop_back();
        }
        const std::string rhs = original.substr(eq + 1);
        size_t first = rhs.find_first_not_of(" \t\"");
        if (first != std::string::npos) {
          size_t last = rhs.find_last_not_of(" \t\"");
          std::string suffix = (last != std::string::npos && last < rhs.size() - 1) ? rhs.substr(last + 1) : "";
          lineOut = key + " = " + rhs.substr(0, first) + num + suffix;
        } else {
          lineOut = key + " = " + num;
        }
      }
    }
    out << lineOut << "\n";
  }
  return true;
}

}  // namespace rc
