// TEST INFRASTRUCTURE (oracle/): print what the reference's own preset parser
// (src/shader/ShaderPreset.cpp, compiled unmodified into oracle/_ref) produces for each
// preset given on the command line, one JSON object per line. tests/ compare the product's
// restated parser with this across the whole shader corpus.
#include <cstdio>
#include <string>
#include <unordered_map>

#include "shader/ShaderPreset.h"

static std::string esc(const std::string& s) {
  std::string o;
  for (char c : s) {
    if (c == '"' || c == '\\') o += '\\';
    if ((unsigned char)c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; continue; }
    o += c;
  }
  return o;
}

int main(int argc, char** argv) {
  // dump_preset --saveas <in.glslp> <out.glslp> [name=value ...]: the reference's ShaderPreset::saveAs
  // (ShaderPreset.cpp:557-661) with the given custom parameters, for the save / reload round-trip test
  if (argc >= 4 && std::string(argv[1]) == "--saveas") {
    ShaderPreset p;
    if (!p.load(argv[2])) return 2;
    std::unordered_map<std::string, float> custom;
    for (int a = 4; a < argc; ++a) {
      std::string kv = argv[a];
      size_t eq = kv.find('=');
      if (eq == std::string::npos) return 3;
      custom[kv.substr(0, eq)] = std::stof(kv.substr(eq + 1));
    }
    return p.saveAs(argv[3], custom) ? 0 : 4;
  }
  for (int a = 1; a < argc; ++a) {
    ShaderPreset p;
    bool ok = p.load(argv[a]);
    printf("{\"preset\":\"%s\",\"ok\":%s,\"passes\":[", esc(argv[a]).c_str(), ok ? "true" : "false");
    bool first = true;
    for (auto& s : p.getPasses()) {
      printf("%s{\"shader\":\"%s\",\"filter_linear\":%s,\"wrap\":\"%s\",\"mipmap\":%s,\"alias\":\"%s\","
             "\"float_fb\":%s,\"srgb_fb\":%s,\"fcm\":%u,\"stx\":\"%s\",\"sx\":%.9g,\"sty\":\"%s\",\"sy\":%.9g}",
             first ? "" : ",", esc(s.shaderPath).c_str(), s.filterLinear ? "true" : "false",
             esc(s.wrapMode).c_str(), s.mipmapInput ? "true" : "false", esc(s.alias).c_str(),
             s.floatFramebuffer ? "true" : "false", s.srgbFramebuffer ? "true" : "false",
             s.frameCountMod, esc(s.scaleTypeX).c_str(), s.scaleX, esc(s.scaleTypeY).c_str(), s.scaleY);
      first = false;
    }
    printf("],\"textures\":{");
    first = true;
    for (auto& t : p.getTextures()) {
      printf("%s\"%s\":{\"path\":\"%s\",\"wrap\":\"%s\",\"mipmap\":%s,\"linear\":%s}", first ? "" : ",",
             esc(t.first).c_str(), esc(t.second.path).c_str(), esc(t.second.wrapMode).c_str(),
             t.second.mipmap ? "true" : "false", t.second.linear ? "true" : "false");
      first = false;
    }
    printf("},\"params\":{");
    first = true;
    for (auto& q : p.getParameters()) {
      printf("%s\"%s\":%.9g", first ? "" : ",", esc(q.first).c_str(), q.second);
      first = false;
    }
    printf("}}\n");
  }
  return 0;
}
