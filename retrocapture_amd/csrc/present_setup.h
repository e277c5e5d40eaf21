// Host side of kernels/present.hip: turns one off-screen OpenGLRenderer::renderTexture call
// (reference src/renderer/OpenGLRenderer.cpp:378-470) into a PresentLaunch.
#pragma once
#include <cstdint>

#include "kernels/present.h"

namespace rc {

struct PresentDesc {
  uint32_t srcW = 0, srcH = 0;
  bool srcRgb = false;      // source is a GL_RGB texture (the captured frame): alpha samples as 1
  bool srcLinear = true;
  uint32_t dstW = 0, dstH = 0;
  int dstKind = rck::PRESENT_RGBA8;
  int vpX = 0, vpY = 0, vpW = 0, vpH = 0;  // vpW == 0: the whole target
  bool flipY = false;
  float brightness = 1.0f, contrast = 1.0f;
  float clear[4] = {0.f, 0.f, 0.f, 0.f};
  bool bake = false;
  float bakeBrightness = 1.0f, bakeContrast = 1.0f;
  bool outFlipRows = false;
};

// false: invalid geometry (empty source / target, non-positive viewport, dstKind out of range)
bool makePresentLaunch(const PresentDesc& d, const void* dSrc, void* dDst, uint32_t nFrames, rck::PresentLaunch* out);

// The pre-pass viewport for an overscan crop of `pct` percent per side (reference
// src/core/FrameCapturePipeline.cpp:205-216): x, y, w, h.
void overscanViewport(uint32_t fboW, uint32_t fboH, float pctX, float pctY, int vp[4]);

}  // namespace rc
