#include "present_setup.h"

#include <algorithm>
#include <cmath>

namespace rc {

namespace {
uint32_t packClear(const float c[4]) {
  uint32_t p = 0;
  for (int k = 0; k < 4; ++k) {
    float x = c[k];
    uint32_t b = 0;
    if (x > 0.0f) b = (uint32_t)std::nearbyintf((x > 1.0f ? 1.0f : x) * 255.0f);
    p |= b << (8 * k);
  }
  return p;
}
}  // namespace

bool makePresentLaunch(const PresentDesc& d, const void* dSrc, void* dDst, uint32_t nFrames, rck::PresentLaunch* out) {
  if (!out || d.srcW == 0 || d.srcH == 0 || d.dstW == 0 || d.dstH == 0 || d.dstKind < 0 || d.dstKind > 2) return false;
  if ((uint64_t)d.srcW * d.srcH * 4 >= (1ull << 32) || (uint64_t)d.dstW * d.dstH * 4 >= (1ull << 32)) return false;  // 32-bit texel offsets
  const int vpX = d.vpW ? d.vpX : 0, vpY = d.vpW ? d.vpY : 0;
  const int vpW = d.vpW ? d.vpW : (int)d.dstW, vpH = d.vpW ? d.vpH : (int)d.dstH;
  if (vpW <= 0 || vpH <= 0) return false;
  rck::PresentLaunch L{};
  L.src.base = dSrc;
  L.src.frame_stride = (uint64_t)d.srcW * d.srcH * 4;
  L.src.w = (int)d.srcW;
  L.src.h = (int)d.srcH;
  L.src.fmt = d.srcRgb ? rcd::FMT_RGBX8 : rcd::FMT_RGBA8;
  L.src.linear = d.srcLinear ? 1 : 0;
  L.src.wrap = rcd::WRAP_EDGE;
  L.dst = dDst;
  L.dst_w = (int)d.dstW;
  L.dst_h = (int)d.dstH;
  L.dst_kind = d.dstKind;
  L.dst_frame_stride = (uint64_t)d.dstW * d.dstH * (d.dstKind == rck::PRESENT_RGB24 ? 3 : 4);
  L.cov_x0 = std::max(vpX, 0);
  L.cov_y0 = std::max(vpY, 0);
  L.cov_x1 = std::min(vpX + vpW, (int)d.dstW);
  L.cov_y1 = std::min(vpY + vpH, (int)d.dstH);
  L.flip_y = d.flipY ? 1 : 0;
  L.out_flip_rows = d.outFlipRows ? 1 : 0;
  L.brightness = d.brightness;
  L.contrast = d.contrast;
  L.bake = d.bake ? 1 : 0;
  L.bake_brightness = d.bakeBrightness;
  L.bake_contrast = d.bakeContrast;
  // llvmpipe's rectangle path: one TexCoord plane per axis, anchored at the quad's top-right vertex
  // (vpX + vpW, vpY + vpH); same operations as varying.cpp's RGBA8 branch with the viewport's extent.
  const float fw = (float)vpW, fh = (float)vpH;
  const float ooa = 1.0f / (fw * fh);
  const float hy = fh * ooa, wx = fw * ooa;
  const float xr = (float)(vpX + vpW), yt = (float)(vpY + vpH);
  {
    const float dx = hy * (1.0f - 0.0f), dy = wx * 0.0f;
    L.u_dx = dx;
    L.u_a0 = 1.0f - (dx * (xr - 0.5f) + dy * (yt - 0.5f));
  }
  {
    const float dx = hy * 0.0f, dy = wx * (1.0f - 0.0f);
    L.v_dy = dy;
    L.v_a0 = 1.0f - (dx * (xr - 0.5f) + dy * (yt - 0.5f));
  }
  L.clear = packClear(d.clear);
  L.n_frames = (int)nFrames;
  *out = L;
  return true;
}

void overscanViewport(uint32_t fboW, uint32_t fboH, float pctX, float pctY, int vp[4]) {
  const float ox = std::max(0.0f, std::min(0.45f, pctX / 100.0f)), oy = std::max(0.0f, std::min(0.45f, pctY / 100.0f));
  const float fx = 1.0f - 2.0f * ox, fy = 1.0f - 2.0f * oy;
  const float w = (float)fboW / fx, h = (float)fboH / fy;
  vp[0] = (int)(((float)fboW - w) / 2.0f);
  vp[1] = (int)(((float)fboH - h) / 2.0f);
  vp[2] = (int)w;
  vp[3] = (int)h;
}

}  // namespace rc
