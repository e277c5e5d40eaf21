// TEST INFRASTRUCTURE (oracle/): run a RetroArch .glslp preset's GLSL passes on Mesa
// llvmpipe and dump every pass's render target, to produce golden vectors.
//
// The reference's ShaderEngine cannot be built in this image without writing a stand-in
// for GLFW (src/renderer/glad_loader.cpp:13 includes <GLFW/glfw3.h>, which the image
// lacks), so it is NOT built.  What IS used from the reference:
//   * its preset parser, compiled unmodified from where it lies
//     (src/shader/ShaderPreset.cpp + src/utils/{Logger,Paths}.cpp; no GL, no stand-ins);
//   * its GLSL shader assets, read as data at run time.
// The GL pass plumbing below is this repo's own restatement of what the reference does
// around each draw (file:line cited per step); it issues the same GL calls so that the
// same driver executes the same shader text under the same sampler / target state.
//
//   glchain --preset P.glslp --input in.rgb24 --w W --h H --vw VW --vh VH
//           [--frames N] [--lut NAME=file.rgba:W:H ...] [--param NAME=V ...] --out DIR
// Writes DIR/pass<k>.bin (stored texels: RGBA8 bytes or RGBA32F floats, GL row order)
// and DIR/meta.txt.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <regex>
#include <sstream>
#include <string>
#include <vector>

#include "headless_gl.h"
#include "shader/ShaderPreset.h"  // the reference's own parser (built into oracle/_ref)

using namespace glrun;

namespace {

struct Pass {
  ShaderPass info;
  GLuint program = 0, fbo = 0, tex = 0;
  GLuint written_tex = 0;
  GLuint fb_fbo = 0, fb_tex = 0;  // PassFeedback ping-pong partner (ShaderEngine.h feedbackTexture / feedbackFramebuffer)
  bool feedback_enabled = false;
  uint32_t w = 0, h = 0;
  std::map<std::string, float> params;  // #pragma parameter defaults
};

std::string read_file(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  std::stringstream s;
  s << f.rdbuf();
  return s.str();
}

std::string dirname_of(const std::string& p) {
  size_t k = p.find_last_of('/');
  return k == std::string::npos ? "." : p.substr(0, k);
}

// reference: ShaderPreprocessor::processIncludes (ShaderPreprocessor.cpp:222-363):
// only lines that START with #include are expanded, relative to the including file.
std::string expand_includes(const std::string& src, const std::string& dir, int depth = 0) {
  if (depth > 16) return src;
  std::regex inc(R"([ \t]*#include\s+["<]([^">]+)[">].*)");
  std::stringstream in(src), out;
  std::string line;
  while (std::getline(in, line)) {
    std::smatch m;
    if (std::regex_match(line, m, inc)) {
      std::string path = m[1].str()[0] == '/' ? m[1].str() : dir + "/" + m[1].str();
      std::ifstream f(path);
      if (f) {
        out << expand_includes(read_file(path), dirname_of(path), depth + 1) << "\n";
      }
      continue;
    }
    out << line << "\n";
  }
  return out.str();
}

// reference: ShaderPreprocessor::preprocess (ShaderPreprocessor.cpp:11-220).
void preprocess(const std::string& path, std::string& vs, std::string& fs,
                std::map<std::string, float>& params) {
  std::string src = expand_includes(read_file(path), dirname_of(path));
  // #pragma parameter NAME "desc" default min max step  (cpp:36); names containing
  // "bogus_" are labels (cpp:48).
  std::regex prag(
      "#pragma\\s+parameter\\s+(\\w+)\\s+\"([^\"]*)\"\\s+(-?[\\d.]+)\\s+(-?[\\d.]+)\\s+(-?[\\d.]+)\\s+(-?[\\d.]+)");
  for (auto it = std::sregex_iterator(src.begin(), src.end(), prag); it != std::sregex_iterator();
       ++it) {
    std::string name = (*it)[1].str();
    if (name.find("bogus_") != std::string::npos) continue;
    float v = 0.f;
    try {
      v = std::stof((*it)[3].str());
    } catch (...) {
    }
    params[name] = v;
  }
  // blank the pragma lines (cpp:83-95)
  for (size_t p = src.find("#pragma parameter"); p != std::string::npos;
       p = src.find("#pragma parameter", p)) {
    size_t e = src.find('\n', p);
    if (e == std::string::npos) e = src.size();
    for (size_t j = p; j < e; ++j) src[j] = ' ';
    p = e;
  }
  // keep the file's own #version if any, else the driver pick, which on a 3.3+ desktop
  // context is "#version 330" (glad_loader.cpp:297-322)
  std::string version = "#version 330\n";
  std::regex ver(R"(#version\s+\d+[^\n]*)");
  std::smatch vm;
  if (std::regex_search(src, vm, ver)) {
    version = vm.str() + "\n";
    src = std::regex_replace(src, ver, "", std::regex_constants::format_first_only);
  }
  // the 420pack extension line is emitted when the driver exposes it (cpp:162-172);
  // llvmpipe 4.5 does.
  std::string ext = "#extension GL_ARB_shading_language_420pack : require\n";
  std::string pu = params.empty() ? "" : "#define PARAMETER_UNIFORM\n";  // cpp:207-212
  vs = version + ext + "#define VERTEX\n" + pu + src;
  fs = version + ext + "#define FRAGMENT\n" + pu + src;
}

GLuint compile(GLenum type, const std::string& src, const std::string& what) {
  GLuint s = CreateShader(type);
  const char* p = src.c_str();
  ShaderSource(s, 1, &p, nullptr);
  CompileShader(s);
  GLint ok = 0;
  GetShaderiv(s, GL_COMPILE_STATUS, &ok);
  if (!ok) {
    char log[8192];
    GetShaderInfoLog(s, sizeof(log), nullptr, log);
    fprintf(stderr, "glchain: compile failed (%s):\n%s\n", what.c_str(), log);
    exit(2);
  }
  return s;
}

GLint uloc(GLuint prog, const std::string& n) { return GetUniformLocation(prog, n.c_str()); }

GLenum uniform_type(GLuint prog, const char* name) {
  GLint n = 0;
  GetProgramiv(prog, GL_ACTIVE_UNIFORMS, &n);
  for (GLint i = 0; i < n; ++i) {
    char nm[256];
    GLint sz;
    GLenum ty;
    GetActiveUniform(prog, i, sizeof(nm), nullptr, &sz, &ty, nm);
    if (!strcmp(nm, name)) return ty;
  }
  return 0;
}

GLenum wrap_enum(const std::string& w) {  // ShaderEngine.cpp:3208-3228
  if (w == "repeat") return GL_REPEAT;
  if (w == "mirrored_repeat") return GL_MIRRORED_REPEAT;
  if (w == "clamp_to_border") return GL_CLAMP_TO_BORDER;
  return GL_CLAMP_TO_EDGE;
}

uint32_t calc_scale(uint32_t src, const std::string& type, float scale, uint32_t vp) {
  // ShaderEngine.cpp:1881-1910
  if (type.empty() || type == "source") {
    if (scale == 0.f) scale = 1.f;
    return (uint32_t)std::round(src * scale);
  }
  if (type == "viewport") {
    if (scale == 0.f) scale = 1.f;
    return (uint32_t)std::round(vp * scale);
  }
  if (type == "absolute") return (uint32_t)std::round(scale);
  return src;
}

}  // namespace

int main(int argc, char** argv) {
  std::string preset_path, input_path, out_dir = ".";
  uint32_t W = 0, H = 0, VW = 0, VH = 0;
  int frames = 1;
  std::map<std::string, std::string> lut_files;
  std::map<std::string, float> custom;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() { return std::string(argv[++i]); };
    if (a == "--preset") preset_path = next();
    else if (a == "--input") input_path = next();
    else if (a == "--w") W = atoi(next().c_str());
    else if (a == "--h") H = atoi(next().c_str());
    else if (a == "--vw") VW = atoi(next().c_str());
    else if (a == "--vh") VH = atoi(next().c_str());
    else if (a == "--frames") frames = atoi(next().c_str());
    else if (a == "--out") out_dir = next();
    else if (a == "--lut") {
      std::string v = next();
      lut_files[v.substr(0, v.find('='))] = v.substr(v.find('=') + 1);
    } else if (a == "--param") {
      std::string v = next();
      custom[v.substr(0, v.find('='))] = std::stof(v.substr(v.find('=') + 1));
    }
  }
  if (preset_path.empty() || !W || !H || !VW || !VH) {
    fprintf(stderr, "usage: see header\n");
    return 1;
  }
  if (!create_context()) return 3;

  ShaderPreset preset;
  if (!preset.load(preset_path)) {
    fprintf(stderr, "glchain: preset parse failed\n");
    return 4;
  }
  std::vector<Pass> passes(preset.getPasses().size());
  for (size_t i = 0; i < passes.size(); ++i) {
    Pass& p = passes[i];
    p.info = preset.getPasses()[i];
    std::string vs, fs;
    preprocess(p.info.shaderPath, vs, fs, p.params);
    p.program = CreateProgram();
    AttachShader(p.program, compile(GL_VERTEX_SHADER, vs, p.info.shaderPath + " [VS]"));
    AttachShader(p.program, compile(GL_FRAGMENT_SHADER, fs, p.info.shaderPath + " [FS]"));
    // ShaderEngine.cpp:707-719
    BindAttribLocation(p.program, 0, "Position");
    BindAttribLocation(p.program, 0, "VertexCoord");
    BindAttribLocation(p.program, 1, "TexCoord");
    // :712-718: the motion-blur shaders' per-history-frame coordinates all alias location 1
    for (const char* n : {"PrevTexCoord", "Prev1TexCoord", "Prev2TexCoord", "Prev3TexCoord", "Prev4TexCoord", "Prev5TexCoord", "Prev6TexCoord"})
      BindAttribLocation(p.program, 1, n);
    BindAttribLocation(p.program, 2, "COLOR");
    LinkProgram(p.program);
    GLint ok = 0;
    GetProgramiv(p.program, GL_LINK_STATUS, &ok);
    if (!ok) {
      char log[4096];
      GetProgramInfoLog(p.program, sizeof(log), nullptr, log);
      fprintf(stderr, "glchain: link failed pass %zu: %s\n", i, log);
      return 2;
    }
  }

  // LUTs: decoded to RGBA8 by the fixture script (PNG -> RGBA, ShaderEngine.cpp:2535-2706),
  // uploaded as GL_RGBA.
  std::map<std::string, GLuint> luts;
  for (auto& t : preset.getTextures()) {
    auto it = lut_files.find(t.first);
    if (it == lut_files.end()) {
      fprintf(stderr, "glchain: no --lut given for %s\n", t.first.c_str());
      return 5;
    }
    std::string spec = it->second;  // file:W:H
    size_t c2 = spec.rfind(':'), c1 = spec.rfind(':', c2 - 1);
    int lw = atoi(spec.substr(c1 + 1, c2 - c1 - 1).c_str()), lh = atoi(spec.substr(c2 + 1).c_str());
    std::string data = read_file(spec.substr(0, c1));
    GLuint tx;
    GenTextures(1, &tx);
    BindTexture(GL_TEXTURE_2D, tx);
    PixelStorei(GL_UNPACK_ALIGNMENT, 1);
    TexImage2D(GL_TEXTURE_2D, 0, GL_RGBA, lw, lh, 0, GL_RGBA, GL_UNSIGNED_BYTE, data.data());
    TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
    TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
    luts[t.first] = tx;
  }

  // quad: ShaderEngine.cpp:2945-2985
  float quad[] = {-1, -1, 0, 1, 0, 0, 1, -1, 0, 1, 1, 0, 1, 1, 0, 1, 1, 1, -1, 1, 0, 1, 0, 1};
  unsigned idx[] = {0, 1, 2, 2, 3, 0};
  GLuint vao, vbo, ebo;
  GenVertexArrays(1, &vao);
  BindVertexArray(vao);
  GenBuffers(1, &vbo);
  BindBuffer(GL_ARRAY_BUFFER, vbo);
  BufferData(GL_ARRAY_BUFFER, sizeof(quad), quad, GL_STATIC_DRAW);
  GenBuffers(1, &ebo);
  BindBuffer(GL_ELEMENT_ARRAY_BUFFER, ebo);
  BufferData(GL_ELEMENT_ARRAY_BUFFER, sizeof(idx), idx, GL_STATIC_DRAW);
  VertexAttribPointer(0, 4, GL_FLOAT, GL_FALSE, 24, nullptr);
  EnableVertexAttribArray(0);
  VertexAttribPointer(1, 2, GL_FLOAT, GL_FALSE, 24, (void*)16);
  EnableVertexAttribArray(1);
  BindVertexArray(0);

  // source texture: FrameProcessor.cpp:172-205 (GL_RGB internal format, RGB24 upload,
  // NEAREST); row 0 of the buffer is t = 0.
  std::string in = read_file(input_path);
  // one RGB24 frame (re-applied every frame) or `frames` frames back to back (a moving source)
  const bool in_seq = frames > 1 && in.size() == (size_t)W * H * 3 * (size_t)frames;
  if (!in_seq && in.size() != (size_t)W * H * 3) {
    fprintf(stderr, "glchain: input must be %ux%u RGB24 (x1 or x frames)\n", W, H);
    return 6;
  }
  GLuint src_tex;
  GenTextures(1, &src_tex);
  BindTexture(GL_TEXTURE_2D, src_tex);
  PixelStorei(GL_UNPACK_ALIGNMENT, 1);
  TexImage2D(GL_TEXTURE_2D, 0, GL_RGB, W, H, 0, GL_RGB, GL_UNSIGNED_BYTE, in.data());
  TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
  TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
  TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
  TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);

  // frame history: ShaderEngine.h:140-143 (ring of at most 7 textures, newest first)
  std::vector<GLuint> history;
  std::vector<uint32_t> history_w, history_h;
  GLuint copy_fbo = 0;
  const size_t kMaxHistory = 7;

  const bool force_f32 = getenv("GLCHAIN_F32") != nullptr;
  float frame_count = 0.f, time_s = 0.f;
  for (int f = 0; f < frames; ++f) {
    frame_count += 1.0f;  // ShaderEngine.cpp:1688-1689
    time_s += 0.016f;
    if (in_seq && f > 0) {  // FrameProcessor.cpp:177 glTexSubImage2D of the next captured frame
      BindTexture(GL_TEXTURE_2D, src_tex);
      PixelStorei(GL_UNPACK_ALIGNMENT, 1);
      TexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, W, H, GL_RGB, GL_UNSIGNED_BYTE, in.data() + (size_t)f * W * H * 3);
    }
    GLuint cur_tex = src_tex;
    uint32_t cw = W, ch = H;
    for (size_t i = 0; i < passes.size(); ++i) {
      Pass& p = passes[i];
      const ShaderPass& pi = p.info;
      // output size: ShaderEngine.cpp:858-894
      bool last = (i == passes.size() - 1);
      std::string tx = pi.scaleTypeX, ty = pi.scaleTypeY;
      float sx = pi.scaleX, sy = pi.scaleY;
      if (last && tx != "viewport" && (tx.empty() || (tx == "source" && sx == 1.0f))) {
        tx = "viewport";
        sx = 1.0f;
      }
      if (last && ty != "viewport" && (ty.empty() || (ty == "source" && sy == 1.0f))) {
        ty = "viewport";
        sy = 1.0f;
      }
      uint32_t ow = calc_scale(cw, tx, sx, VW), oh = calc_scale(ch, ty, sy, VH);
      if (!p.fbo || p.w != ow || p.h != oh) {
        // createFramebuffer: ShaderEngine.cpp:2872-2923
        GenTextures(1, &p.tex);
        BindTexture(GL_TEXTURE_2D, p.tex);
        // GLCHAIN_F32=1: every pass renders to RGBA32F, so that the shaders' arithmetic can be compared
        // at float precision (8-bit targets hide last-bit differences); not the reference's formats
        // GLCHAIN_F32_LAST=1: only the last pass (to look at one pass's floats behind the reference's own formats)
        const bool f32 = pi.floatFramebuffer || force_f32 || (getenv("GLCHAIN_F32_LAST") != nullptr && i + 1 == passes.size());
        GLenum ifmt = f32 ? GL_RGBA32F : pi.srgbFramebuffer ? GL_SRGB8_ALPHA8 : (getenv("GLCHAIN_RGBA8") ? GL_RGBA8 : GL_RGBA);
        if (getenv("GLCHAIN_NODITHER")) Disable(GL_DITHER);
        TexImage2D(GL_TEXTURE_2D, 0, ifmt, ow, oh, 0, GL_RGBA, f32 ? GL_FLOAT : GL_UNSIGNED_BYTE, nullptr);
        TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
        TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
        TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
        TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
        GenFramebuffers(1, &p.fbo);
        BindFramebuffer(GL_FRAMEBUFFER, p.fbo);
        FramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, p.tex, 0);
        if (CheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) {
          fprintf(stderr, "glchain: FBO incomplete pass %zu\n", i);
          return 7;
        }
        p.w = ow;
        p.h = oh;
      }
      BindFramebuffer(GL_FRAMEBUFFER, p.fbo);
      if (pi.srgbFramebuffer && !force_f32) Enable(GL_FRAMEBUFFER_SRGB);  // ShaderEngine.cpp:944-952
      else Disable(GL_FRAMEBUFFER_SRGB);
      Viewport(0, 0, ow, oh);
      ColorMask(1, 1, 1, 1);
      ClearColor(0, 0, 0, 0);
      if (!getenv("GLCHAIN_NOCLEAR")) Clear(GL_COLOR_BUFFER_BIT);
      Disable(GL_BLEND);
      Disable(GL_CULL_FACE);
      Disable(GL_DEPTH_TEST);
      UseProgram(p.program);
      GLuint pr = p.program;
      GLint l;
      // ---- uniforms: ShaderEngine.cpp:1912-2533
      if ((l = uloc(pr, "SourceSize")) >= 0) Uniform4f(l, cw, ch, 1.f / cw, 1.f / ch);
      if ((l = uloc(pr, "OriginalSize")) >= 0) Uniform4f(l, W, H, 1.f / W, 1.f / H);
      if ((l = uloc(pr, "OutputSize")) >= 0) {
        GLenum t = uniform_type(pr, "OutputSize");
        if (t == GL_FLOAT_VEC3) Uniform3f(l, ow, oh, 1.f / ow);
        else if (t == GL_FLOAT_VEC4) Uniform4f(l, ow, oh, 1.f / ow, 1.f / oh);
        else Uniform2f(l, ow, oh);
      }
      for (size_t k = 0; k < i; ++k) {
        if ((l = uloc(pr, "PassOutputSize" + std::to_string(k))) >= 0)
          Uniform4f(l, passes[k].w, passes[k].h, 1.f / passes[k].w, 1.f / passes[k].h);
        if ((l = uloc(pr, "PassInputSize" + std::to_string(k))) >= 0) {
          float iw = k == 0 ? W : passes[k - 1].w, ih = k == 0 ? H : passes[k - 1].h;
          Uniform4f(l, iw, ih, 1.f / iw, 1.f / ih);
        }
      }
      if ((l = uloc(pr, "PassScale")) >= 0) Uniform1f(l, (pi.scaleX + pi.scaleY) / 2.f);
      if ((l = uloc(pr, "PassScaleX")) >= 0) Uniform1f(l, pi.scaleX);
      if ((l = uloc(pr, "PassScaleY")) >= 0) Uniform1f(l, pi.scaleY);
      if ((l = uloc(pr, "PassFilter")) >= 0) Uniform1f(l, pi.filterLinear ? 1.f : 0.f);
      {
        float fc = frame_count;
        if (pi.frameCountMod > 0) fc = fmodf(frame_count, (float)pi.frameCountMod);
        if ((l = uloc(pr, "FrameCount")) >= 0) {
          if (uniform_type(pr, "FrameCount") == GL_INT) Uniform1i(l, (GLint)fc);
          else Uniform1f(l, fc);
        }
      }
      if ((l = uloc(pr, "MVPMatrix")) >= 0) {
        float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        UniformMatrix4fv(l, 1, GL_FALSE, m);
      }
      if ((l = uloc(pr, "FrameDirection")) >= 0) Uniform1i(l, 1);
      for (int k = 0; k <= 7; ++k)
        if ((l = uloc(pr, "OriginalHistorySize" + std::to_string(k))) >= 0)
          Uniform4f(l, cw, ch, 1.f / cw, 1.f / ch);  // no history in this runner
      for (auto& kv : p.params) {
        if ((l = uloc(pr, kv.first)) < 0) continue;
        float v = kv.second;
        auto c = custom.find(kv.first);
        if (c != custom.end()) v = c->second;
        else {
          auto g = preset.getParameters().find(kv.first);
          if (g != preset.getParameters().end()) v = g->second;
        }
        Uniform1f(l, v);
      }
      {  // hard-coded overrides, ShaderEngine.cpp:2260-2374
        static const struct { const char* n; float v; } ov[] = {
            {"BLURSCALEX", .30f}, {"LOWLUMSCAN", 6.f}, {"HILUMSCAN", 8.f}, {"BRIGHTBOOST", 1.25f},
            {"MASK_DARK", .25f}, {"MASK_FADE", .8f}, {"RESSWITCH_ENABLE", 1.f},
            {"RESSWITCH_GLITCH_TRESHOLD", .1f}, {"RESSWITCH_GLITCH_BAR_STR", .6f},
            {"RESSWITCH_GLITCH_BAR_SIZE", .5f}, {"RESSWITCH_GLITCH_BAR_SMOOTH", 1.f},
            {"RESSWITCH_GLITCH_SHAKE_MAX", .25f}, {"RESSWITCH_GLITCH_ROT_MAX", .2f},
            {"RESSWITCH_GLITCH_WOB_MAX", .1f}, {"AS", .20f}, {"asat", .33f}, {"PR", .32f},
            {"PG", .32f}, {"PB", .32f}, {"internal_res", 1.f}, {"auto_res", 0.f}};
        for (auto& o : ov)
          if ((l = uloc(pr, o.n)) >= 0) Uniform1f(l, o.v);
      }
      if ((l = uloc(pr, "TextureSize")) >= 0)
        Uniform2f(l, cw, (oh != ch && i == 3) ? (float)oh : (float)ch);  // cpp:2418-2426
      if ((l = uloc(pr, "InputSize")) >= 0) Uniform2f(l, cw, ch);
      if ((l = uloc(pr, "IN.video_size")) >= 0) Uniform2f(l, W, H);
      if ((l = uloc(pr, "IN.texture_size")) >= 0) Uniform2f(l, cw, ch);
      if ((l = uloc(pr, "IN.output_size")) >= 0) Uniform2f(l, ow, oh);
      if ((l = uloc(pr, "IN.frame_count")) >= 0) Uniform1f(l, frame_count);
      if ((l = uloc(pr, "FRAMEINDEX")) >= 0) Uniform1f(l, frame_count);
      if ((l = uloc(pr, "TIME")) >= 0) Uniform1f(l, time_s);
      for (auto& kv : preset.getParameters())
        if ((l = uloc(pr, kv.first)) >= 0) Uniform1f(l, kv.second);

      // ---- input texture + its sampler state: ShaderEngine.cpp:991-1036
      ActiveTexture(GL_TEXTURE0);
      BindTexture(GL_TEXTURE_2D, cur_tex);
      GLenum filt = pi.filterLinear ? GL_LINEAR : GL_NEAREST;
      TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, filt);
      TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, filt);
      if (pi.mipmapInput) {
        TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER,
                      pi.filterLinear ? GL_LINEAR_MIPMAP_LINEAR : GL_NEAREST_MIPMAP_NEAREST);
        GenerateMipmap(GL_TEXTURE_2D);
      }
      TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, wrap_enum(pi.wrapMode));
      TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, wrap_enum(pi.wrapMode));
      for (const char* n : {"Texture", "Source", "Input", "s_p", "tex", "image"})
        if ((l = uloc(pr, n)) >= 0) {
          Uniform1i(l, 0);
          break;
        }
      int unit = 1;
      if (i == 0) {
        // frame history for the first pass: ShaderEngine.cpp:1095-1159.  A Prev sampler is bound
        // only when that much history exists; otherwise its uniform keeps its old value.
        bool needs_history = false;
        for (int k = 0; k < 7; ++k)
          if (uloc(pr, k == 0 ? std::string("PrevTexture") : "Prev" + std::to_string(k) + "Texture") >= 0) needs_history = true;
        for (int k = 0; k < 7; ++k) {
          std::vector<std::string> names = {k == 0 ? std::string("PrevTexture") : "Prev" + std::to_string(k) + "Texture",
                                            "PassPrev" + std::to_string(k) + "Texture"};
          for (auto& nm : names)
            if ((l = uloc(pr, nm)) >= 0) {
              if (needs_history && (size_t)k < history.size() && history[k] != 0) {
                ActiveTexture(GL_TEXTURE0 + unit);
                BindTexture(GL_TEXTURE_2D, history[k]);
                Uniform1i(l, unit++);
              }
              break;
            }
        }
      }
      if (i > 0) {
        // previous passes: ShaderEngine.cpp:1163-1228
        for (size_t pp = 0; pp < i; ++pp) {
          std::string n = std::to_string(i - pp);
          std::vector<std::string> names = {"PassPrev" + n + "Texture",
                                            pp == 0 ? "PrevTexture" : "Prev" + std::to_string(pp) + "Texture"};
          for (auto& nm : names)
            if ((l = uloc(pr, nm)) >= 0) {
              ActiveTexture(GL_TEXTURE0 + unit);
              BindTexture(GL_TEXTURE_2D, passes[pp].tex);
              Uniform1i(l, unit++);
              break;
            }
          if ((l = uloc(pr, "PassPrev" + n + "TextureSize")) >= 0) Uniform2f(l, passes[pp].w, passes[pp].h);
          if ((l = uloc(pr, "PassPrev" + n + "InputSize")) >= 0)
            Uniform2f(l, pp == 0 ? W : passes[pp - 1].w, pp == 0 ? H : passes[pp - 1].h);
          if ((l = uloc(pr, "PassPrev" + n + "OutputSize")) >= 0) Uniform2f(l, passes[pp].w, passes[pp].h);
        }
        for (size_t N = i + 1; N <= i + 12; ++N)  // cpp:1234-1245
          if ((l = uloc(pr, "PassPrev" + std::to_string(N) + "Texture")) >= 0) {
            ActiveTexture(GL_TEXTURE0 + unit);
            BindTexture(GL_TEXTURE_2D, src_tex);
            Uniform1i(l, unit++);
          }
        for (size_t pp = 0; pp < i; ++pp) {  // aliases, cpp:1251-1277
          const std::string& al = passes[pp].info.alias;
          if (al.empty()) continue;
          if ((l = uloc(pr, al)) >= 0) {
            ActiveTexture(GL_TEXTURE0 + unit);
            BindTexture(GL_TEXTURE_2D, passes[pp].tex);
            Uniform1i(l, unit++);
          }
          if ((l = uloc(pr, al + "Size")) >= 0)
            Uniform4f(l, passes[pp].w, passes[pp].h, 1.f / passes[pp].w, 1.f / passes[pp].h);
        }
      }
      // PassFeedback<N>: the previous frame's output of pass N <= i, second texture allocated on
      // first sight (createFramebuffer, same size and format), swapped at the end of the frame:
      // ShaderEngine.cpp:1285-1347
      for (size_t fp = 0; fp <= i; ++fp) {
        const std::string n = std::to_string(fp);
        GLint tl = uloc(pr, "PassFeedback" + n);
        if (tl < 0) tl = uloc(pr, "PassFeedback" + n + "Texture");
        GLint sl = uloc(pr, "PassFeedback" + n + "Size");
        if (sl < 0) sl = uloc(pr, "PassFeedback" + n + "TextureSize");
        if (tl < 0 && sl < 0) continue;
        Pass& t = passes[fp];
        t.feedback_enabled = true;
        if (t.fb_tex == 0 && t.w > 0 && t.h > 0) {
          GenTextures(1, &t.fb_tex);
          BindTexture(GL_TEXTURE_2D, t.fb_tex);
          GLenum ifmt = t.info.floatFramebuffer ? GL_RGBA32F : t.info.srgbFramebuffer ? GL_SRGB8_ALPHA8 : GL_RGBA;
          TexImage2D(GL_TEXTURE_2D, 0, ifmt, t.w, t.h, 0, GL_RGBA, t.info.floatFramebuffer ? GL_FLOAT : GL_UNSIGNED_BYTE, nullptr);
          TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
          TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
          TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
          TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
          GenFramebuffers(1, &t.fb_fbo);
          BindFramebuffer(GL_FRAMEBUFFER, t.fb_fbo);
          FramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, t.fb_tex, 0);
          CheckFramebufferStatus(GL_FRAMEBUFFER);
          BindFramebuffer(GL_FRAMEBUFFER, 0);  // createFramebuffer leaves FBO 0 bound (cpp:2931-2932)
          BindTexture(GL_TEXTURE_2D, 0);
        }
        if (tl >= 0 && t.fb_tex != 0) {
          ActiveTexture(GL_TEXTURE0 + unit);
          BindTexture(GL_TEXTURE_2D, t.fb_tex);
          Uniform1i(tl, unit++);
        }
        if (sl >= 0) Uniform4f(sl, t.w, t.h, t.w > 0 ? 1.f / t.w : 0.f, t.h > 0 ? 1.f / t.h : 0.f);
      }
      if ((l = uloc(pr, "OrigTexture")) >= 0) {  // cpp:1351-1358
        ActiveTexture(GL_TEXTURE0 + unit);
        BindTexture(GL_TEXTURE_2D, src_tex);
        Uniform1i(l, unit++);
      }
      for (auto& lt : luts) {  // cpp:1361-1415
        ActiveTexture(GL_TEXTURE0 + unit);
        BindTexture(GL_TEXTURE_2D, lt.second);
        const ShaderTexture& st = preset.getTextures().at(lt.first);
        GLenum lf = st.linear ? GL_LINEAR : GL_NEAREST;
        TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, lf);
        TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, lf);
        if (st.mipmap) {
          TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER,
                        st.linear ? GL_LINEAR_MIPMAP_LINEAR : GL_NEAREST_MIPMAP_NEAREST);
          GenerateMipmap(GL_TEXTURE_2D);
        }
        TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, wrap_enum(st.wrapMode));
        TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, wrap_enum(st.wrapMode));
        if ((l = uloc(pr, lt.first)) >= 0) Uniform1i(l, unit);
        unit++;
      }
      BindVertexArray(vao);
      ActiveTexture(GL_TEXTURE0);
      BindTexture(GL_TEXTURE_2D, cur_tex);
      DrawElements(GL_TRIANGLES, 6, GL_UNSIGNED_INT, nullptr);  // cpp:1448
      BindVertexArray(0);
      cur_tex = p.tex;
      p.written_tex = p.tex;  // what this frame rendered into (the feedback swap below renames it)
      cw = ow;
      ch = oh;
    }
    BindFramebuffer(GL_FRAMEBUFFER, 0);
    Disable(GL_FRAMEBUFFER_SRGB);
    BindTexture(GL_TEXTURE_2D, 0);  // unit 0 is the active one after the last draw (cpp:1703-1704)
    UseProgram(0);
    for (auto& fp : passes) {  // PassFeedback ping-pong swap: cpp:1710-1718
      if (!fp.feedback_enabled || fp.fb_tex == 0) continue;
      std::swap(fp.tex, fp.fb_tex);
      std::swap(fp.fbo, fp.fb_fbo);
    }
    // history push: the final output is re-drawn through pass 0's program, with pass 0's uniforms
    // as they stand, into a history texture: ShaderEngine.cpp:1735-1865
    if (cur_tex != 0 && cw > 0 && ch > 0) {
      GLuint ht = 0;
      if (history.size() < kMaxHistory) {
        GenTextures(1, &ht);
        BindTexture(GL_TEXTURE_2D, ht);
        TexImage2D(GL_TEXTURE_2D, 0, GL_RGBA, cw, ch, 0, GL_RGBA, GL_UNSIGNED_BYTE, nullptr);
        TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
        TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
        TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
        TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
        BindTexture(GL_TEXTURE_2D, 0);
      } else {
        ht = history.back();
        history.pop_back();
        history_w.pop_back();
        history_h.pop_back();
        if (history_w.empty() || history_w.back() != cw || history_h.back() != ch) {
          BindTexture(GL_TEXTURE_2D, ht);
          TexImage2D(GL_TEXTURE_2D, 0, GL_RGBA, cw, ch, 0, GL_RGBA, GL_UNSIGNED_BYTE, nullptr);
          BindTexture(GL_TEXTURE_2D, 0);
        }
      }
      if (!copy_fbo) GenFramebuffers(1, &copy_fbo);
      BindFramebuffer(GL_FRAMEBUFFER, copy_fbo);
      FramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, ht, 0);
      if (CheckFramebufferStatus(GL_FRAMEBUFFER) == GL_FRAMEBUFFER_COMPLETE) {
        Viewport(0, 0, cw, ch);
        ClearColor(0.f, 0.f, 0.f, 1.f);
        Clear(GL_COLOR_BUFFER_BIT);
        if (passes[0].program) {
          UseProgram(passes[0].program);
          BindVertexArray(vao);
          ActiveTexture(GL_TEXTURE0);
          BindTexture(GL_TEXTURE_2D, cur_tex);
          GLint tl = uloc(passes[0].program, "Texture");
          if (tl < 0) tl = uloc(passes[0].program, "Source");
          if (tl >= 0) Uniform1i(tl, 0);
          EnableVertexAttribArray(0);
          EnableVertexAttribArray(1);
          DrawElements(GL_TRIANGLES, 6, GL_UNSIGNED_INT, nullptr);
          BindVertexArray(0);
          BindTexture(GL_TEXTURE_2D, 0);
          UseProgram(0);
        }
      }
      BindFramebuffer(GL_FRAMEBUFFER, 0);
      history.insert(history.begin(), ht);
      history_w.insert(history_w.begin(), cw);
      history_h.insert(history_h.begin(), ch);
    }
    Finish();
  }

  // dump the stored texels of every pass
  std::ofstream meta(out_dir + "/meta.txt");
  PixelStorei(GL_PACK_ALIGNMENT, 1);
  for (size_t i = 0; i < passes.size(); ++i) {
    Pass& p = passes[i];
    BindTexture(GL_TEXTURE_2D, p.written_tex ? p.written_tex : p.tex);
    std::string fn = out_dir + "/pass" + std::to_string(i) + ".bin";
    FILE* fo = fopen(fn.c_str(), "wb");
    const bool last_f32 = getenv("GLCHAIN_F32_LAST") != nullptr && i + 1 == passes.size();
    if (p.info.floatFramebuffer || force_f32 || last_f32) {
      std::vector<float> d((size_t)p.w * p.h * 4);
      GetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_FLOAT, d.data());
      fwrite(d.data(), 4, d.size(), fo);
    } else {
      // bytes as stored: an sRGB8 texture returns its encoded bytes through glGetTexImage
      std::vector<unsigned char> d((size_t)p.w * p.h * 4);
      GetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_UNSIGNED_BYTE, d.data());
      fwrite(d.data(), 1, d.size(), fo);
    }
    fclose(fo);
    meta << "pass " << i << " " << p.w << " " << p.h << " "
         << ((p.info.floatFramebuffer || force_f32 || last_f32) ? "f32" : p.info.srgbFramebuffer ? "srgb8" : "rgba8") << " lin="
         << p.info.filterLinear << " wrap=" << p.info.wrapMode << " alias=" << p.info.alias
         << " shader=" << p.info.shaderPath << "\n";
    for (auto& kv : p.params) meta << "  param " << kv.first << " " << kv.second << "\n";
  }
  for (size_t k = 0; k < history.size(); ++k) {
    BindTexture(GL_TEXTURE_2D, history[k]);
    std::vector<unsigned char> d((size_t)history_w[k] * history_h[k] * 4);
    GetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_UNSIGNED_BYTE, d.data());
    FILE* fo = fopen((out_dir + "/history" + std::to_string(k) + ".bin").c_str(), "wb");
    fwrite(d.data(), 1, d.size(), fo);
    fclose(fo);
    meta << "history " << k << " " << history_w[k] << " " << history_h[k] << "\n";
  }
  GLenum e = GetError();
  if (e) fprintf(stderr, "glchain: GL error 0x%x\n", e);
  return 0;
}
