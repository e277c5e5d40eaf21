/* TEST INFRASTRUCTURE - see rc_oracle.h.
 *
 * Frame ingest / egress as the reference's CPU code does them:
 *   ingest  reference src/processing/FrameProcessor.cpp:43-222 - RGB24 is uploaded as is, BGRA / RGBA
 *           are swizzled by the GL into a GL_RGB texture (alpha dropped, samples as 1), YUYV422 goes
 *           through libswscale (sws_getContext(..., AV_PIX_FMT_YUYV422 -> AV_PIX_FMT_RGB24, SWS_POINT),
 *           :249-284).
 *   egress  reference src/core/FrameCapturePipeline.cpp:1060-1080 - drop every fourth byte.
 * Parity: the byte shuffles are exact by construction.  YUYV422 is "PARITY UNPINNED": libswscale
 * (FFmpeg, a system dependency of the reference, not vendored and not installed here) converts with
 * SIMD code whose rounding is build-specific; what is restated is the published BT.601 limited-range
 * integer conversion with libswscale's own ITU-601 coefficients (yuv2rgb.c ff_yuv2rgb_coeffs:
 * 104597, 132201, 25675, 53279; luma gain 65536*255/219), chroma shared by each pixel pair as
 * libswscale does for RGB output without SWS_FULL_CHR_H_INT.
 */
#include <stdint.h>
#include <stddef.h>

static inline uint8_t clip8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* fmt: 0 RGB24, 1 BGRA, 2 RGBA, 3 YUYV422; dst: n_px RGBA8 pixels, alpha 255 */
void o_ingest(const uint8_t* src, int fmt, size_t n_px, uint8_t* dst) {
  for (size_t p = 0; p < n_px; ++p) {
    uint8_t r, g, b;
    if (fmt == 0) { r = src[3 * p]; g = src[3 * p + 1]; b = src[3 * p + 2]; }
    else if (fmt == 1) { b = src[4 * p]; g = src[4 * p + 1]; r = src[4 * p + 2]; }
    else if (fmt == 2) { r = src[4 * p]; g = src[4 * p + 1]; b = src[4 * p + 2]; }
    else {
      const uint8_t* m = src + 4 * (p / 2);
      const int y = m[(p & 1) ? 2 : 0], u = m[1], v = m[3];
      const int c = 76309 * (y - 16) + 32768, d = u - 128, e = v - 128;
      r = clip8((c + 104597 * e) >> 16);
      g = clip8((c - 25675 * d - 53279 * e) >> 16);
      b = clip8((c + 132201 * d) >> 16);
    }
    dst[4 * p] = r; dst[4 * p + 1] = g; dst[4 * p + 2] = b; dst[4 * p + 3] = 255;
  }
}

void o_egress_rgb24(const uint8_t* src, int w, int h, int n, int flip_y, uint8_t* dst) {
  for (int f = 0; f < n; ++f)
    for (int y = 0; y < h; ++y) {
      const uint8_t* s = src + ((size_t)f * h + (flip_y ? h - 1 - y : y)) * w * 4;
      uint8_t* d = dst + ((size_t)f * h + y) * w * 3;
      for (int x = 0; x < w; ++x) { d[3 * x] = s[4 * x]; d[3 * x + 1] = s[4 * x + 1]; d[3 * x + 2] = s[4 * x + 2]; }
    }
}
