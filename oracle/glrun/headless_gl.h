// TEST INFRASTRUCTURE (oracle/): headless OpenGL 3.3-core context on Mesa llvmpipe.
//
// There is no X server, EGL, OSMesa, GLFW or SDL2 in the build image, but Mesa's
// software rasteriser (swrast_dri.so = llvmpipe) and <GL/internal/dri_interface.h>
// are installed, so a context is created by talking to the DRI "swrast" driver
// interface directly.  This is the same GL the reference requests
// (reference: src/output/WindowManager.cpp:61-63 asks GLFW for a 3.3 core profile).
//
// Nothing here is part of the product; it exists so the reference's GLSL shader
// assets can be executed by a real GL implementation to produce golden vectors.
#pragma once
#include <GL/gl.h>
#include <GL/glext.h>

namespace glrun {

// Creates the llvmpipe context and makes it current. Returns false on failure.
bool create_context(int gl_major = 3, int gl_minor = 3);

// glXGetProcAddress equivalent (Mesa's glapi dispatch table).
void* get_proc(const char* name);

// GL 2.0+ entry points used by the runner (resolved by create_context()).
#define GLRUN_FUNCS(X) \
  X(PFNGLCREATESHADERPROC, CreateShader) X(PFNGLSHADERSOURCEPROC, ShaderSource) \
  X(PFNGLCOMPILESHADERPROC, CompileShader) X(PFNGLGETSHADERIVPROC, GetShaderiv) \
  X(PFNGLGETSHADERINFOLOGPROC, GetShaderInfoLog) X(PFNGLCREATEPROGRAMPROC, CreateProgram) \
  X(PFNGLATTACHSHADERPROC, AttachShader) X(PFNGLBINDATTRIBLOCATIONPROC, BindAttribLocation) \
  X(PFNGLLINKPROGRAMPROC, LinkProgram) X(PFNGLGETPROGRAMIVPROC, GetProgramiv) \
  X(PFNGLGETPROGRAMINFOLOGPROC, GetProgramInfoLog) X(PFNGLUSEPROGRAMPROC, UseProgram) \
  X(PFNGLGETUNIFORMLOCATIONPROC, GetUniformLocation) X(PFNGLGETACTIVEUNIFORMPROC, GetActiveUniform) \
  X(PFNGLUNIFORM1IPROC, Uniform1i) X(PFNGLUNIFORM1FPROC, Uniform1f) X(PFNGLUNIFORM2FPROC, Uniform2f) \
  X(PFNGLUNIFORM3FPROC, Uniform3f) X(PFNGLUNIFORM4FPROC, Uniform4f) \
  X(PFNGLUNIFORMMATRIX4FVPROC, UniformMatrix4fv) \
  X(PFNGLGENFRAMEBUFFERSPROC, GenFramebuffers) X(PFNGLBINDFRAMEBUFFERPROC, BindFramebuffer) \
  X(PFNGLFRAMEBUFFERTEXTURE2DPROC, FramebufferTexture2D) \
  X(PFNGLCHECKFRAMEBUFFERSTATUSPROC, CheckFramebufferStatus) \
  X(PFNGLDELETEFRAMEBUFFERSPROC, DeleteFramebuffers) \
  X(PFNGLGENVERTEXARRAYSPROC, GenVertexArrays) X(PFNGLBINDVERTEXARRAYPROC, BindVertexArray) \
  X(PFNGLGENBUFFERSPROC, GenBuffers) X(PFNGLBINDBUFFERPROC, BindBuffer) \
  X(PFNGLBUFFERDATAPROC, BufferData) X(PFNGLVERTEXATTRIBPOINTERPROC, VertexAttribPointer) \
  X(PFNGLENABLEVERTEXATTRIBARRAYPROC, EnableVertexAttribArray) \
  X(PFNGLACTIVETEXTUREPROC, ActiveTexture) X(PFNGLGENERATEMIPMAPPROC, GenerateMipmap) \
  X(PFNGLDELETESHADERPROC, DeleteShader) X(PFNGLDELETEPROGRAMPROC, DeleteProgram)

#define X(T, N) extern T N;
GLRUN_FUNCS(X)
#undef X

// GL 1.x entry points (also fetched through glapi; libGL.so is not linked).
#define GLRUN_FUNCS1(X) \
  X(const GLubyte*, GetString, (GLenum n), (n)) \
  X(void, GenTextures, (GLsizei n, GLuint* t), (n, t)) \
  X(void, BindTexture, (GLenum a, GLuint t), (a, t)) \
  X(void, DeleteTextures, (GLsizei n, const GLuint* t), (n, t)) \
  X(void, TexParameteri, (GLenum a, GLenum p, GLint v), (a, p, v)) \
  X(void, TexParameterfv, (GLenum a, GLenum p, const GLfloat* v), (a, p, v)) \
  X(void, PixelStorei, (GLenum p, GLint v), (p, v)) \
  X(void, Viewport, (GLint x, GLint y, GLsizei w, GLsizei h), (x, y, w, h)) \
  X(void, ClearColor, (GLfloat r, GLfloat g, GLfloat b, GLfloat a), (r, g, b, a)) \
  X(void, Clear, (GLbitfield m), (m)) \
  X(void, ColorMask, (GLboolean r, GLboolean g, GLboolean b, GLboolean a), (r, g, b, a)) \
  X(void, Enable, (GLenum c), (c)) X(void, Disable, (GLenum c), (c)) \
  X(void, Finish, (void), ()) X(GLenum, GetError, (void), ()) \
  X(void, DrawElements, (GLenum m, GLsizei c, GLenum t, const void* i), (m, c, t, i)) \
  X(void, ReadPixels, (GLint x, GLint y, GLsizei w, GLsizei h, GLenum f, GLenum t, void* d), (x, y, w, h, f, t, d)) \
  X(void, GetTexImage, (GLenum a, GLint l, GLenum f, GLenum t, void* d), (a, l, f, t, d)) \
  X(void, TexImage2D, (GLenum a, GLint l, GLint i, GLsizei w, GLsizei h, GLint b, GLenum f, GLenum t, const void* d), (a, l, i, w, h, b, f, t, d)) \
  X(void, TexSubImage2D, (GLenum a, GLint l, GLint x, GLint y, GLsizei w, GLsizei h, GLenum f, GLenum t, const void* d), (a, l, x, y, w, h, f, t, d))

#define X(R, N, A, C) R N A;
GLRUN_FUNCS1(X)
#undef X

}  // namespace glrun
