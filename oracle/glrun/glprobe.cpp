// TEST INFRASTRUCTURE (oracle/): evaluate a GLSL expression on Mesa llvmpipe over a
// buffer of float4 inputs and return the float4 results, unrounded.
//
// Used to pin the oracle's transcendental restatements (pow/exp2/log2/sin/cos/...)
// and its sampler/rounding rules against what the GL that executes the reference's
// shaders really computes ("extract them empirically").
//
//   glprobe <body.glsl> <W> <H> [u8|srgb8|f32] [texfile TW TH rgba8|srgb8|rgb8|f32 linear|nearest WRAP] < in.f32 > out
//
// The optional second texture is bound as `uniform sampler2D S` (with `uniform vec2 SSize`)
// with the given internal format, filter and wrap (edge|border|repeat|mirror), so the
// oracle's sampler restatement can be checked against llvmpipe's texture unit.
//
// body.glsl must define `vec4 f(vec4 v)`; it may use `TC` (the interpolated TexCoord varying), `uniform sampler2D T` (the
// input as an RGBA32F NEAREST texture) and `uniform vec2 Size`.
// Input: W*H float4.  Output: W*H float4 (f32) or W*H RGBA8 bytes (u8 / srgb8 target,
// to observe UNORM8 rounding and the sRGB encode).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <sstream>
#include <string>
#include <vector>

#include "headless_gl.h"

using namespace glrun;

static GLuint compile(GLenum type, const std::string& src) {
  GLuint s = CreateShader(type);
  const char* p = src.c_str();
  ShaderSource(s, 1, &p, nullptr);
  CompileShader(s);
  GLint ok = 0;
  GetShaderiv(s, GL_COMPILE_STATUS, &ok);
  if (!ok) {
    char log[4096];
    GetShaderInfoLog(s, sizeof(log), nullptr, log);
    fprintf(stderr, "glprobe: compile failed:\n%s\n", log);
    exit(2);
  }
  return s;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: glprobe body.glsl W H [u8|srgb8|f32]\n");
    return 1;
  }
  std::ifstream bf(argv[1]);
  std::stringstream bs;
  bs << bf.rdbuf();
  int W = atoi(argv[2]), H = atoi(argv[3]);
  std::string target = argc > 4 ? argv[4] : "f32";
  if (!create_context()) return 3;

  std::vector<float> in((size_t)W * H * 4);
  if (fread(in.data(), sizeof(float), in.size(), stdin) != in.size()) {
    fprintf(stderr, "glprobe: short input\n");
    return 4;
  }

  std::string vs =
      "#version 330\nin vec4 P;\nin vec2 TexCoord;\nout vec2 TC;\nvoid main(){ gl_Position = P; TC = TexCoord; }\n";
  std::string fs =
      "#version 330\n#extension GL_ARB_texture_query_lod : enable\nuniform sampler2D T;\nuniform sampler2D S;\nuniform vec2 Size;\nuniform vec2 SSize;\nin vec2 TC;\nout vec4 O;\n" + bs.str() +
      "\nvoid main(){ O = f(texelFetch(T, ivec2(gl_FragCoord.xy), 0)); }\n";
  GLuint prog = CreateProgram();
  AttachShader(prog, compile(GL_VERTEX_SHADER, vs));
  AttachShader(prog, compile(GL_FRAGMENT_SHADER, fs));
  BindAttribLocation(prog, 0, "P");
  BindAttribLocation(prog, 1, "TexCoord");
  LinkProgram(prog);
  GLint ok = 0;
  GetProgramiv(prog, GL_LINK_STATUS, &ok);
  if (!ok) {
    char log[4096];
    GetProgramInfoLog(prog, sizeof(log), nullptr, log);
    fprintf(stderr, "glprobe: link failed:\n%s\n", log);
    return 2;
  }

  GLuint tin = 0, tout = 0, fbo = 0, vao = 0, vbo = 0, ebo = 0;
  GenTextures(1, &tin);
  BindTexture(GL_TEXTURE_2D, tin);
  TexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, W, H, 0, GL_RGBA, GL_FLOAT, in.data());
  TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
  TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);

  GLuint ts = 0;
  int TW = 0, TH = 0;
  if (argc >= 11) {
    TW = atoi(argv[6]);
    TH = atoi(argv[7]);
    std::string tf = argv[8], filt = argv[9], wrap = argv[10];
    std::ifstream f(argv[5], std::ios::binary);
    std::vector<char> td((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    GenTextures(1, &ts);
    BindTexture(GL_TEXTURE_2D, ts);
    PixelStorei(GL_UNPACK_ALIGNMENT, 1);
    if (tf == "f32")
      TexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, TW, TH, 0, GL_RGBA, GL_FLOAT, td.data());
    else if (tf == "rgb8")
      TexImage2D(GL_TEXTURE_2D, 0, GL_RGB, TW, TH, 0, GL_RGB, GL_UNSIGNED_BYTE, td.data());
    else
      TexImage2D(GL_TEXTURE_2D, 0, tf == "srgb8" ? GL_SRGB8_ALPHA8 : GL_RGBA, TW, TH, 0, GL_RGBA,
                 GL_UNSIGNED_BYTE, td.data());
    // "trilinear" / "mipnearest": the state ShaderEngine sets for mipmap_input (ShaderEngine.cpp:1022-1033)
    const bool mip = filt == "trilinear" || filt == "mipnearest";
    GLenum fl = (filt == "linear" || filt == "trilinear") ? GL_LINEAR : GL_NEAREST;
    TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, fl);
    TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, fl);
    if (mip) {
      TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, filt == "trilinear" ? GL_LINEAR_MIPMAP_LINEAR : GL_NEAREST_MIPMAP_NEAREST);
      GenerateMipmap(GL_TEXTURE_2D);
      if (const char* dump = getenv("GLPROBE_DUMP_LEVELS")) {   // every generated level, raw stored bytes / floats
        FILE* df = fopen(dump, "wb");
        for (int lv = 0, lw = TW, lh = TH;; ++lv) {
          std::vector<char> buf((size_t)lw * lh * (tf == "f32" ? 16 : 4));
          GetTexImage(GL_TEXTURE_2D, lv, GL_RGBA, tf == "f32" ? GL_FLOAT : GL_UNSIGNED_BYTE, buf.data());
          fwrite(buf.data(), 1, buf.size(), df);
          if (lw == 1 && lh == 1) break;
          lw = lw > 1 ? lw / 2 : 1;
          lh = lh > 1 ? lh / 2 : 1;
        }
        fclose(df);
      }
    }
    GLenum wr = wrap == "border" ? GL_CLAMP_TO_BORDER
                : wrap == "repeat" ? GL_REPEAT
                : wrap == "mirror" ? GL_MIRRORED_REPEAT
                                   : GL_CLAMP_TO_EDGE;
    TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, wr);
    TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, wr);
  }

  GLenum ifmt = target == "u8" ? GL_RGBA8 : target == "srgb8" ? GL_SRGB8_ALPHA8 : GL_RGBA32F;
  GenTextures(1, &tout);
  BindTexture(GL_TEXTURE_2D, tout);
  TexImage2D(GL_TEXTURE_2D, 0, ifmt, W, H, 0, GL_RGBA,
             ifmt == GL_RGBA32F ? GL_FLOAT : GL_UNSIGNED_BYTE, nullptr);
  TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
  TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
  GenFramebuffers(1, &fbo);
  BindFramebuffer(GL_FRAMEBUFFER, fbo);
  FramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, tout, 0);
  if (CheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) {
    fprintf(stderr, "glprobe: FBO incomplete\n");
    return 5;
  }
  if (target == "srgb8") Enable(GL_FRAMEBUFFER_SRGB);

  // Same vertex layout and index order as the reference quad (ShaderEngine.cpp:2945-2960).
  float quad[] = {-1, -1, 0, 1, 0, 0, 1, -1, 0, 1, 1, 0, 1, 1, 0, 1, 1, 1, -1, 1, 0, 1, 0, 1};
  unsigned idx[] = {0, 1, 2, 2, 3, 0};
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

  Viewport(0, 0, W, H);
  UseProgram(prog);
  ActiveTexture(GL_TEXTURE0);
  BindTexture(GL_TEXTURE_2D, tin);
  Uniform1i(GetUniformLocation(prog, "T"), 0);
  if (ts) {
    ActiveTexture(GL_TEXTURE1);
    BindTexture(GL_TEXTURE_2D, ts);
    GLint l = GetUniformLocation(prog, "S");
    if (l >= 0) Uniform1i(l, 1);
    l = GetUniformLocation(prog, "SSize");
    if (l >= 0) Uniform2f(l, (float)TW, (float)TH);
    ActiveTexture(GL_TEXTURE0);
  }
  GLint sl = GetUniformLocation(prog, "Size");
  if (sl >= 0) Uniform2f(sl, (float)W, (float)H);
  DrawElements(GL_TRIANGLES, 6, GL_UNSIGNED_INT, nullptr);
  Finish();

  PixelStorei(GL_PACK_ALIGNMENT, 1);
  if (ifmt == GL_RGBA32F) {
    std::vector<float> out((size_t)W * H * 4);
    ReadPixels(0, 0, W, H, GL_RGBA, GL_FLOAT, out.data());
    fwrite(out.data(), sizeof(float), out.size(), stdout);
  } else {
    // Read the stored bytes without any conversion on the way out.
    Disable(GL_FRAMEBUFFER_SRGB);
    std::vector<unsigned char> out((size_t)W * H * 4);
    BindTexture(GL_TEXTURE_2D, tout);
    GetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_UNSIGNED_BYTE, out.data());
    fwrite(out.data(), 1, out.size(), stdout);
  }
  GLenum e = GetError();
  if (e) fprintf(stderr, "glprobe: GL error 0x%x\n", e);
  return 0;
}
