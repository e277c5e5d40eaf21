// Launch descriptor of kernels/present.hip (OpenGLRenderer::renderTexture off-screen).
#pragma once
#include <hip/hip_runtime.h>

#include "rc_device.h"

namespace rck {

enum PresentKind : int { PRESENT_RGBA8 = 0, PRESENT_RGBX8 = 1, PRESENT_RGB24 = 2 };

struct PresentLaunch {
  rcd::Tex src;               // RGBX8 (captured frame: alpha samples as 1) or RGBA8; clamp to edge
  void* dst;
  uint64_t dst_frame_stride;
  int dst_w, dst_h, dst_kind;
  int cov_x0, cov_y0, cov_x1, cov_y1;  // target pixels the viewport quad covers; the rest keep `clear`
  int flip_y;                 // the program's flipY uniform (coordinate flip)
  int out_flip_rows;          // rows stored bottom-up (readback flip), independent of flipY
  float brightness, contrast;
  int bake;                   // second draw at the same size on the first one's RGBA8 result
  float bake_brightness, bake_contrast;
  float u_a0, u_dx, v_a0, v_dy;  // TexCoord planes of the viewport quad (host: present_setup)
  uint32_t clear;             // RGBA8
  int n_frames;
};

hipError_t launch_present(const PresentLaunch& L, hipStream_t s);

}  // namespace rck
