// Host-to-host frame path around a ShaderEngine: what the reference does per captured frame
// between the capture buffer and the encoder's RGB24 buffer - FrameProcessor::processFrame upload
// (reference src/processing/FrameProcessor.cpp:43-222), ShaderEngine::applyShader, and the readback
// with its PBO double buffering (reference src/renderer/PBOManager.cpp:86-170, "lags one frame",
// src/core/FrameCapturePipeline.cpp:974-1084) - as a ring of slots on three HIP streams:
//   copy-in stream   pinned host frame  -> device        (hipMemcpyAsync H2D)
//   engine stream    rc ingest kernel -> [source pre-pass] -> shader chain -> rc egress kernel
//                    (or the fused resize / image-adjust / strip kernel, kernels/present.hip)
//   copy-out stream  device RGB24 -> pinned host          (hipMemcpyAsync D2H)
// linked by events, so the copies of frame n+1 / n-1 overlap the kernels of frame n.  Frames come
// back in submission order.  Presets with frame history or PassFeedback stay correct: the engine
// stream serialises the chain.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "shader_engine.h"

namespace rc {

class FramePipeline : public EngineClient {
 public:
  FramePipeline(ShaderEngine* engine, int slots);
  ~FramePipeline();
  bool ok() const { return m_ok; }
  // Copies the host frame (tightly packed, pixfmt as rc_pixfmt) into the next free slot and queues it.
  // Returns false if every slot is still waiting to be received, or on a device error.
  bool submit(const void* hostFrame, int pixfmt, uint32_t width, uint32_t height);
  // Pinned staging memory of the slot the next submit() will use, sized for such a frame: a caller that
  // captures straight into it and passes the same pointer to submit() saves the host-side copy.
  void* inputBuffer(int pixfmt, uint32_t width, uint32_t height);
  // Oldest submitted frame: RGB24, row 0 first, in pinned memory owned by the pipeline.  The frame stays the
  // caller's until the NEXT receive() (or the pipeline's destruction): its slot is not reused by submit() /
  // inputBuffer() before that, so a ring of N slots carries at most N - 1 frames in flight besides the one the
  // caller holds.  wait = false returns false when the frame is not finished yet.
  bool receive(const void** hostRgb24, uint32_t* width, uint32_t* height, bool wait);
  // The engine is shutting down (ShaderEngine::shutdown / destructor): drain, then refuse further work.
  void engineGone() override;
  int inFlight() const { return m_inFlight; }
  void setFlipY(bool flip) { m_flipY = flip; }
  // Optional stages of the reference's frame path (src/core/FrameCapturePipeline.cpp): NEAREST
  // downscale to a logical capture size + overscan crop before the chain (:160-250), output
  // resolution (:413-505) and brightness / contrast bake (:739-804) after it.
  void setSourcePrepass(uint32_t logicalW, uint32_t logicalH, float overscanPctX, float overscanPctY) {
    m_logicalW = logicalW;
    m_logicalH = logicalH;
    m_overscanX = overscanPctX;
    m_overscanY = overscanPctY;
  }
  void setOutputResolution(uint32_t w, uint32_t h) {
    m_outW = w;
    m_outH = h;
  }
  void setImageAdjust(float brightness, float contrast) {
    m_brightness = brightness;
    m_contrast = contrast;
  }

 private:
  struct Slot {
    void* hostIn = nullptr;
    void* hostOut = nullptr;
    void* devIn = nullptr;
    void* devRgba = nullptr;
    void* devPre = nullptr;   // pre-pass target (the chain's source when the pre-pass is on)
    void* devOut = nullptr;
    size_t hostInBytes = 0, hostOutBytes = 0, devInBytes = 0, devRgbaBytes = 0, devPreBytes = 0, devOutBytes = 0;
    hipEvent_t h2dDone = nullptr, computeDone = nullptr, d2hDone = nullptr;
    uint32_t outW = 0, outH = 0;
  };
  bool grow(void** p, size_t* have, size_t need, bool host);
  ShaderEngine* m_engine;
  std::vector<Slot> m_slots;
  hipStream_t m_in = nullptr, m_out = nullptr;
  int m_head = 0, m_tail = 0, m_inFlight = 0;
  int m_held = -1;   // slot whose output the caller received last and may still be reading
  bool full() const { return m_inFlight + (m_held >= 0 ? 1 : 0) >= (int)m_slots.size(); }
  bool m_ok = false, m_flipY = false;
  uint32_t m_logicalW = 0, m_logicalH = 0, m_outW = 0, m_outH = 0;
  float m_overscanX = 0.0f, m_overscanY = 0.0f, m_brightness = 1.0f, m_contrast = 1.0f;
};

}  // namespace rc
