/* TEST INFRASTRUCTURE - see rc_oracle.h.
 *
 * OpenGLRenderer::renderTexture as FrameCapturePipeline uses it off-screen (reference
 * src/renderer/OpenGLRenderer.cpp:378-470, fragment program :141-158, quad :292-307):
 *   - the shader-source pre-pass: NEAREST downscale + overscan crop of the captured frame into a
 *     GL_RGB target through an enlarged, offset viewport (src/core/FrameCapturePipeline.cpp:160-250);
 *   - the output-resolution resize into a GL_RGBA target (:413-505);
 *   - the brightness / contrast bake into a GL_RGBA target (:739-804).
 * One textured quad covering the viewport (vp_x, vp_y, vp_w, vp_h); target pixels outside it keep
 * `clear`.  Arithmetic as measured on llvmpipe (oracle/glrun/glpresent.cpp, float target):
 *   rgb = ((t.rgb * brightness) - 0.5) * contrast + 0.5, every operation rounded (no fma); alpha = t.a.
 */
#include <math.h>
#include <string.h>

#include "rc_oracle.h"

static o_varying present_plane(float a_l_or_b, float a_r_or_t, int along_x, const o_present_args* a) {
  /* the rectangle path's single plane (rc_varying.c), for a quad at the viewport rectangle instead of
   * the whole target: anchored at the top-right vertex (vp_x + vp_w, vp_y + vp_h) */
  o_varying v;
  const float fw = (float)a->vp_w, fh = (float)a->vp_h;
  const float ooa = 1.0f / (fw * fh);
  const float hy = fh * ooa, wx = fw * ooa;
  const float dx = along_x ? hy * (a_r_or_t - a_l_or_b) : hy * 0.0f;
  const float dy = along_x ? wx * 0.0f : wx * (a_r_or_t - a_l_or_b);
  const float xr = (float)(a->vp_x + a->vp_w), yt = (float)(a->vp_y + a->vp_h);
  const float a0 = a_r_or_t - (dx * (xr - 0.5f) + dy * (yt - 0.5f));
  v.dx_lo = v.dx_up = dx;
  v.dy_lo = v.dy_up = dy;
  v.a0_lo = v.a0_up = a0;
  return v;
}

void o_present(const o_present_args* a) {
  unsigned csr = o_fp_enter();
  const o_varying tu = present_plane(0.f, 1.f, 1, a), tv = present_plane(0.f, 1.f, 0, a);
  const int x_lo = a->vp_x > 0 ? a->vp_x : 0, y_lo = a->vp_y > 0 ? a->vp_y : 0;
  const int x_hi = a->vp_x + a->vp_w < a->dst_w ? a->vp_x + a->vp_w : a->dst_w;
  const int y_hi = a->vp_y + a->vp_h < a->dst_h ? a->vp_y + a->vp_h : a->dst_h;
  for (int y = 0; y < a->dst_h; ++y)
    for (int x = 0; x < a->dst_w; ++x) {
      o_vec4 c = a->clear;
      if (x >= x_lo && x < x_hi && y >= y_lo && y < y_hi) {
        const float u = o_varying_at(&tu, x, y, 1);
        float v = o_varying_at(&tv, x, y, 1);
        if (a->flip_y) v = 1.0f - v;
        const o_vec4 t = o_sample(a->src, u, v);
        c.x = ((t.x * a->brightness) - 0.5f) * a->contrast + 0.5f;
        c.y = ((t.y * a->brightness) - 0.5f) * a->contrast + 0.5f;
        c.z = ((t.z * a->brightness) - 0.5f) * a->contrast + 0.5f;
        c.w = t.w;
      }
      if (a->dst_fmt == O_FMT_F32) {
        float* d = (float*)a->dst + ((size_t)y * a->dst_w + x) * 4;
        d[0] = c.x; d[1] = c.y; d[2] = c.z; d[3] = c.w;
      } else {
        uint8_t* d = (uint8_t*)a->dst + ((size_t)y * a->dst_w + x) * 4;
        d[0] = o_store_unorm8(c.x); d[1] = o_store_unorm8(c.y); d[2] = o_store_unorm8(c.z);
        d[3] = a->dst_fmt == O_FMT_RGBX8 ? 255 : o_store_unorm8(c.w);
      }
    }
  o_fp_leave(csr);
}

void o_overscan_viewport(int fbo_w, int fbo_h, float pct_x, float pct_y, int vp[4]) {
  const float ox = fmaxf(0.0f, fminf(0.45f, pct_x / 100.0f)), oy = fmaxf(0.0f, fminf(0.45f, pct_y / 100.0f));
  const float fx = 1.0f - 2.0f * ox, fy = 1.0f - 2.0f * oy;
  const float w = (float)fbo_w / fx, h = (float)fbo_h / fy;
  vp[0] = (int)(((float)fbo_w - w) / 2.0f);
  vp[1] = (int)(((float)fbo_h - h) / 2.0f);
  vp[2] = (int)w;
  vp[3] = (int)h;
}
