// Plane equations of vertex-shader outputs over the full-target quad.
//
// The reference draws every pass as the quad (-1,-1)..(1,1) with TexCoord (0,0)..(1,1) as
// two triangles (reference ShaderEngine.cpp:2945-2960, :1448) and an identity MVP
// (:2152-2162), so every varying is an affine function of the pixel index.  The planes are
// computed here with the float operation order of the GL the reference is measured on, which
// differs by target format: plain RGBA8 targets are rasterised as one rectangle (one plane
// anchored at the top-right vertex), sRGB8 / float targets as the two triangles (see
// DESIGN.md "Varyings").  Compiled with -ffp-contract=off.
#pragma once
#include "kernels/rc_device.h"

namespace rc {
// Vertex values at bottom-left, bottom-right, top-right, top-left.
rcd::Plane makePlane(float a_bl, float a_br, float a_tr, float a_tl, int W, int H, int out_fmt);
// the same for the quad the GL's own blits draw (mip level generation)
rcd::Plane makePlaneFan(float a_bl, float a_br, float a_tr, float a_tl, int W, int H, int out_fmt);
// TexCoord-proportional varyings: value = k * TexCoord.x (or .y)
inline rcd::Plane planeU(float k, int W, int H, int fmt) { return makePlane(0.f * k, 1.f * k, 1.f * k, 0.f * k, W, H, fmt); }
inline rcd::Plane planeV(float k, int W, int H, int fmt) { return makePlane(0.f * k, 0.f * k, 1.f * k, 1.f * k, W, H, fmt); }
}  // namespace rc
