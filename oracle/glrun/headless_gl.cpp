// TEST INFRASTRUCTURE (oracle/): see headless_gl.h.
#include "headless_gl.h"

#include <GL/internal/dri_interface.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstring>

namespace glrun {

#define X(T, N) T N = nullptr;
GLRUN_FUNCS(X)
#undef X

#define X(R, N, A, C) static R(*p_##N) A = nullptr;
GLRUN_FUNCS1(X)
#undef X
#define X(R, N, A, C) \
  R N A { return p_##N C; }
GLRUN_FUNCS1(X)
#undef X

namespace {

typedef void* (*glapi_gpa_t)(const char*);
glapi_gpa_t g_gpa = nullptr;

// The swrast loader wants a window system to blit to / read from. We render only
// into FBOs, so the "window" is a 16x16 dummy that discards everything.
void drawable_info(__DRIdrawable*, int* x, int* y, int* w, int* h, void*) {
  *x = 0;
  *y = 0;
  *w = 16;
  *h = 16;
}
void put_image(__DRIdrawable*, int, int, int, int, int, char*, void*) {}
void get_image(__DRIdrawable*, int, int, int w, int h, char* data, void*) {
  memset(data, 0, (size_t)w * (size_t)h * 4);
}

__DRIswrastLoaderExtension make_loader() {
  __DRIswrastLoaderExtension l;
  memset(&l, 0, sizeof(l));
  l.base.name = __DRI_SWRAST_LOADER;
  l.base.version = 1;
  l.getDrawableInfo = drawable_info;
  l.putImage = put_image;
  l.getImage = get_image;
  return l;
}

}  // namespace

void* get_proc(const char* name) { return g_gpa ? g_gpa(name) : nullptr; }

bool create_context(int gl_major, int gl_minor) {
  static __DRIswrastLoaderExtension loader = make_loader();
  static const __DRIextension* loader_exts[] = {&loader.base, nullptr};

  const char* drv = "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so";
  void* h = dlopen(drv, RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    fprintf(stderr, "glrun: dlopen(%s): %s\n", drv, dlerror());
    return false;
  }
  typedef const __DRIextension** (*get_exts_t)(void);
  get_exts_t get_exts = (get_exts_t)dlsym(h, "__driDriverGetExtensions_swrast");
  if (!get_exts) {
    fprintf(stderr, "glrun: swrast driver has no __driDriverGetExtensions_swrast\n");
    return false;
  }
  const __DRIextension** exts = get_exts();
  const __DRIcoreExtension* core = nullptr;
  const __DRIswrastExtension* swrast = nullptr;
  for (int i = 0; exts[i]; ++i) {
    if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension*)exts[i];
    if (!strcmp(exts[i]->name, __DRI_SWRAST)) swrast = (const __DRIswrastExtension*)exts[i];
  }
  if (!core || !swrast) {
    fprintf(stderr, "glrun: DRI core/swrast extension missing\n");
    return false;
  }
  const __DRIconfig** configs = nullptr;
  __DRIscreen* screen = swrast->createNewScreen2(0, loader_exts, exts, &configs, nullptr);
  if (!screen || !configs || !configs[0]) {
    fprintf(stderr, "glrun: createNewScreen2 failed\n");
    return false;
  }
  unsigned err = 0;
  unsigned attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, (unsigned)gl_major,
                        __DRI_CTX_ATTRIB_MINOR_VERSION, (unsigned)gl_minor};
  __DRIcontext* ctx = swrast->createContextAttribs(screen, __DRI_API_OPENGL_CORE, configs[0],
                                                   nullptr, 2, attribs, &err, nullptr);
  if (!ctx) {
    fprintf(stderr, "glrun: createContextAttribs(core %d.%d) failed, err=%u\n", gl_major,
            gl_minor, err);
    return false;
  }
  __DRIdrawable* drawable = swrast->createNewDrawable(screen, configs[0], nullptr);
  if (!drawable || !core->bindContext(ctx, drawable, drawable)) {
    fprintf(stderr, "glrun: bindContext failed\n");
    return false;
  }
  void* glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
  if (!glapi) {
    fprintf(stderr, "glrun: dlopen(libglapi.so.0): %s\n", dlerror());
    return false;
  }
  g_gpa = (glapi_gpa_t)dlsym(glapi, "_glapi_get_proc_address");
  if (!g_gpa) return false;

  bool ok = true;
#define X(T, N)                                      \
  N = (T)g_gpa("gl" #N);                             \
  if (!N) {                                          \
    fprintf(stderr, "glrun: missing gl" #N "\n");    \
    ok = false;                                      \
  }
  GLRUN_FUNCS(X)
#undef X
#define X(R, N, A, C)                                \
  p_##N = (R(*) A)g_gpa("gl" #N);                    \
  if (!p_##N) {                                      \
    fprintf(stderr, "glrun: missing gl" #N "\n");    \
    ok = false;                                      \
  }
  GLRUN_FUNCS1(X)
#undef X
  return ok;
}

}  // namespace glrun
