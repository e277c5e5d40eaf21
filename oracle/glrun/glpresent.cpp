// TEST INFRASTRUCTURE (oracle/): the reference's off-screen uses of OpenGLRenderer::renderTexture,
// executed on Mesa llvmpipe to produce golden vectors for rc_present (include/rc_shaderchain.h).
//
// The reference draws one textured quad with a four-line fragment program (reference:
// src/renderer/OpenGLRenderer.cpp:141-158, desktop GL >= 3 branch; quad :292-307; uniforms :408-430):
//     coord = flipY ? (u, 1 - v) : (u, v);  t = texture(coord);
//     rgb = t.rgb * brightness;  rgb = (rgb - 0.5) * contrast + 0.5;  out = (rgb, t.a)
// into (a) the shader-source pre-pass target, GL_RGB, with the overscan viewport
// (src/core/FrameCapturePipeline.cpp:160-250), (b) the output-resolution target, GL_RGBA
// (:413-505), (c) the image-adjustment target, GL_RGBA (:739-804).  The program below is this
// repository's restatement of those four lines; state set-up follows the cited call sites.
//
//   glpresent in.raw SW SH rgb|rgba nearest|linear DW DH rgb|rgba|f32 VPX VPY VPW VPH FLIPY BRIGHT CONTRAST > out
//
// in.raw: SW*SH*3 (rgb) or *4 (rgba) bytes, row 0 first.  out: DW*DH RGBA8 bytes (GL_RGB target: alpha
// reads back 255) or DW*DH float4 (f32 target: pins the arithmetic before the UNORM8 store).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "headless_gl.h"

using namespace glrun;

static GLuint compile(GLenum type, const char* src) {
  GLuint s = CreateShader(type);
  ShaderSource(s, 1, &src, nullptr);
  CompileShader(s);
  GLint ok = 0;
  GetShaderiv(s, GL_COMPILE_STATUS, &ok);
  if (!ok) {
    char log[4096];
    GetShaderInfoLog(s, sizeof(log), nullptr, log);
    fprintf(stderr, "glpresent: compile failed:\n%s\n", log);
    exit(2);
  }
  return s;
}

int main(int argc, char** argv) {
  if (argc < 16) {
    fprintf(stderr, "usage: glpresent in.raw SW SH rgb|rgba nearest|linear DW DH rgb|rgba|f32 VPX VPY VPW VPH FLIPY BRIGHT CONTRAST\n");
    return 1;
  }
  const int SW = atoi(argv[2]), SH = atoi(argv[3]);
  const std::string sfmt = argv[4], filt = argv[5];
  const int DW = atoi(argv[6]), DH = atoi(argv[7]);
  const std::string dfmt = argv[8];
  const int vpx = atoi(argv[9]), vpy = atoi(argv[10]), vpw = atoi(argv[11]), vph = atoi(argv[12]);
  const int flip = atoi(argv[13]);
  const float bright = (float)atof(argv[14]), contrast = (float)atof(argv[15]);
  std::ifstream f(argv[1], std::ios::binary);
  std::vector<char> src((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  if (src.size() != (size_t)SW * SH * (sfmt == "rgb" ? 3 : 4)) {
    fprintf(stderr, "glpresent: input size mismatch\n");
    return 4;
  }
  if (!create_context()) return 3;

  const char* vs =
      "#version 330 core\nlayout (location = 0) in vec2 aPos;\nlayout (location = 1) in vec2 aTexCoord;\nout vec2 TexCoord;\n"
      "void main() { gl_Position = vec4(aPos, 0.0, 1.0); TexCoord = aTexCoord; }\n";
  const char* fs =
      "#version 330 core\nin vec2 TexCoord;\nout vec4 FragColor;\nuniform sampler2D ourTexture;\nuniform int flipY;\n"
      "uniform float brightness;\nuniform float contrast;\n"
      "void main() {\n"
      "  vec2 coord = (flipY == 1) ? vec2(TexCoord.x, 1.0 - TexCoord.y) : TexCoord;\n"
      "  vec4 t = texture(ourTexture, coord);\n"
      "  vec3 c = t.rgb * brightness;\n"
      "  c = (c - 0.5) * contrast + 0.5;\n"
      "  FragColor = vec4(c, t.a);\n"
      "}\n";
  GLuint prog = CreateProgram();
  AttachShader(prog, compile(GL_VERTEX_SHADER, vs));
  AttachShader(prog, compile(GL_FRAGMENT_SHADER, fs));
  LinkProgram(prog);
  GLint ok = 0;
  GetProgramiv(prog, GL_LINK_STATUS, &ok);
  if (!ok) return 2;

  GLuint tin = 0, tout = 0, fbo = 0, vao = 0, vbo = 0, ebo = 0;
  PixelStorei(GL_UNPACK_ALIGNMENT, 1);
  GenTextures(1, &tin);
  BindTexture(GL_TEXTURE_2D, tin);
  if (sfmt == "rgb")
    TexImage2D(GL_TEXTURE_2D, 0, GL_RGB, SW, SH, 0, GL_RGB, GL_UNSIGNED_BYTE, src.data());
  else
    TexImage2D(GL_TEXTURE_2D, 0, GL_RGBA, SW, SH, 0, GL_RGBA, GL_UNSIGNED_BYTE, src.data());
  const GLenum fl = filt == "linear" ? GL_LINEAR : GL_NEAREST;
  TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, fl);
  TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, fl);
  TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
  TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);

  GenTextures(1, &tout);
  BindTexture(GL_TEXTURE_2D, tout);
  if (dfmt == "f32")
    TexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, DW, DH, 0, GL_RGBA, GL_FLOAT, nullptr);
  else if (dfmt == "rgb")
    TexImage2D(GL_TEXTURE_2D, 0, GL_RGB, DW, DH, 0, GL_RGB, GL_UNSIGNED_BYTE, nullptr);
  else
    TexImage2D(GL_TEXTURE_2D, 0, GL_RGBA, DW, DH, 0, GL_RGBA, GL_UNSIGNED_BYTE, nullptr);
  TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
  TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
  TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
  TexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
  GenFramebuffers(1, &fbo);
  BindFramebuffer(GL_FRAMEBUFFER, fbo);
  FramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, tout, 0);
  if (CheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) {
    fprintf(stderr, "glpresent: FBO incomplete\n");
    return 5;
  }

  float quad[] = {-1, -1, 0, 0, 1, -1, 1, 0, 1, 1, 1, 1, -1, 1, 0, 1};
  unsigned idx[] = {0, 1, 2, 2, 3, 0};
  GenVertexArrays(1, &vao);
  BindVertexArray(vao);
  GenBuffers(1, &vbo);
  BindBuffer(GL_ARRAY_BUFFER, vbo);
  BufferData(GL_ARRAY_BUFFER, sizeof(quad), quad, GL_STATIC_DRAW);
  GenBuffers(1, &ebo);
  BindBuffer(GL_ELEMENT_ARRAY_BUFFER, ebo);
  BufferData(GL_ELEMENT_ARRAY_BUFFER, sizeof(idx), idx, GL_STATIC_DRAW);
  VertexAttribPointer(0, 2, GL_FLOAT, GL_FALSE, 16, nullptr);
  EnableVertexAttribArray(0);
  VertexAttribPointer(1, 2, GL_FLOAT, GL_FALSE, 16, (void*)8);
  EnableVertexAttribArray(1);

  Viewport(0, 0, DW, DH);
  ClearColor(0, 0, 0, 0);
  Clear(GL_COLOR_BUFFER_BIT);
  Disable(GL_BLEND);
  UseProgram(prog);
  ActiveTexture(GL_TEXTURE0);
  BindTexture(GL_TEXTURE_2D, tin);
  Uniform1i(GetUniformLocation(prog, "ourTexture"), 0);
  Uniform1i(GetUniformLocation(prog, "flipY"), flip ? 1 : 0);
  Uniform1f(GetUniformLocation(prog, "brightness"), bright);
  Uniform1f(GetUniformLocation(prog, "contrast"), contrast);
  Viewport(vpx, vpy, vpw, vph);
  DrawElements(GL_TRIANGLES, 6, GL_UNSIGNED_INT, nullptr);
  Finish();

  PixelStorei(GL_PACK_ALIGNMENT, 1);
  BindTexture(GL_TEXTURE_2D, tout);
  if (dfmt == "f32") {
    std::vector<float> out((size_t)DW * DH * 4);
    GetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_FLOAT, out.data());
    fwrite(out.data(), sizeof(float), out.size(), stdout);
  } else {
    std::vector<unsigned char> out((size_t)DW * DH * 4);
    GetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_UNSIGNED_BYTE, out.data());
    fwrite(out.data(), 1, out.size(), stdout);
  }
  GLenum e = GetError();
  if (e) fprintf(stderr, "glpresent: GL error 0x%x\n", e);
  return 0;
}
