#include "frame_pipeline.h"

#include <cstring>

#include "kernels/pass_launch.h"
#include "present_setup.h"
#include "rc_log.h"

namespace rc {
namespace {
bool hipOk(hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  RC_LOG_ERROR(std::string("frame pipeline: ") + what + ": " + hipGetErrorString(e));
  return false;
}
size_t frameBytes(int pixfmt, uint32_t w, uint32_t h) {
  const size_t px = (size_t)w * h;
  switch (pixfmt) {
    case 0: return px * 3;
    case 1:
    case 2: return px * 4;
    case 3: return px * 2;
    default: return 0;
  }
}
}  // namespace

FramePipeline::FramePipeline(ShaderEngine* engine, int slots) : m_engine(engine) {
  if (!engine) return;
  if (slots < 2) slots = 2;   // one slot can be held by the caller (see receive)
  if (slots > 8) slots = 8;
  m_slots.resize((size_t)slots);
  if (!hipOk(hipStreamCreateWithFlags(&m_in, hipStreamNonBlocking), "stream") ||
      !hipOk(hipStreamCreateWithFlags(&m_out, hipStreamNonBlocking), "stream"))
    return;
  for (Slot& s : m_slots)
    if (!hipOk(hipEventCreateWithFlags(&s.h2dDone, hipEventDisableTiming), "event") ||
        !hipOk(hipEventCreateWithFlags(&s.computeDone, hipEventDisableTiming), "event") ||
        !hipOk(hipEventCreateWithFlags(&s.d2hDone, hipEventDisableTiming), "event"))
      return;
  m_ok = true;
  m_engine->attachClient(this);
}

void FramePipeline::engineGone() {
  if (m_in) (void)hipStreamSynchronize(m_in);
  if (m_engine) (void)hipStreamSynchronize(m_engine->stream());
  if (m_out) (void)hipStreamSynchronize(m_out);
  m_engine = nullptr;
  m_ok = false;   // submit / receive / inputBuffer now fail; frames already received stay valid
}

FramePipeline::~FramePipeline() {
  if (m_engine) m_engine->detachClient(this);
  if (m_in) (void)hipStreamSynchronize(m_in);
  if (m_out) (void)hipStreamSynchronize(m_out);
  if (m_engine) (void)hipStreamSynchronize(m_engine->stream());
  for (Slot& s : m_slots) {
    if (s.hostIn) (void)hipHostFree(s.hostIn);
    if (s.hostOut) (void)hipHostFree(s.hostOut);
    if (s.devIn) (void)hipFree(s.devIn);
    if (s.devRgba) (void)hipFree(s.devRgba);
    if (s.devPre) (void)hipFree(s.devPre);
    if (s.devOut) (void)hipFree(s.devOut);
    if (s.h2dDone) (void)hipEventDestroy(s.h2dDone);
    if (s.computeDone) (void)hipEventDestroy(s.computeDone);
    if (s.d2hDone) (void)hipEventDestroy(s.d2hDone);
  }
  if (m_in) (void)hipStreamDestroy(m_in);
  if (m_out) (void)hipStreamDestroy(m_out);
}

bool FramePipeline::grow(void** p, size_t* have, size_t need, bool host) {
  if (*p && *have >= need) return true;
  if (*p) (void)(host ? hipHostFree(*p) : hipFree(*p));
  *p = nullptr;
  *have = 0;
  if (!hipOk(host ? hipHostMalloc(p, need, hipHostMallocDefault) : hipMalloc(p, need), "allocation")) return false;
  *have = need;
  return true;
}

bool FramePipeline::submit(const void* hostFrame, int pixfmt, uint32_t width, uint32_t height) {
  if (!m_ok || !hostFrame || full()) return false;
  const size_t inBytes = frameBytes(pixfmt, width, height);
  if (inBytes == 0 || (pixfmt == 3 && (width & 1u))) return false;
  Slot& s = m_slots[(size_t)m_head];
  hipStream_t es = m_engine->stream();
  if (!grow(&s.hostIn, &s.hostInBytes, inBytes, true) || !grow(&s.devIn, &s.devInBytes, inBytes, false) ||
      !grow(&s.devRgba, &s.devRgbaBytes, (size_t)width * height * 4, false))
    return false;
  if (hostFrame != s.hostIn) std::memcpy(s.hostIn, hostFrame, inBytes);  // the caller's buffer need not be pinned
  if (!hipOk(hipMemcpyAsync(s.devIn, s.hostIn, inBytes, hipMemcpyHostToDevice, m_in), "H2D") ||
      !hipOk(hipEventRecord(s.h2dDone, m_in), "event") || !hipOk(hipStreamWaitEvent(es, s.h2dDone, 0), "wait"))
    return false;
  if (!hipOk(rck::launch_ingest(s.devIn, pixfmt, width, height, 1, s.devRgba, es), "ingest")) return false;
  // source pre-pass (FrameCapturePipeline.cpp:160-250): only while a shader is active
  const void* chainSrc = s.devRgba;
  uint32_t cw = width, ch = height;
  const bool needsOverscan = m_overscanX > 0.001f || m_overscanY > 0.001f;
  const bool needsDownscale = m_logicalW > 0 && m_logicalH > 0 && m_logicalW < width && m_logicalH < height;
  if (m_engine->isShaderActive() && (needsDownscale || needsOverscan)) {
    PresentDesc d;
    d.srcW = width;
    d.srcH = height;
    d.srcRgb = true;
    d.srcLinear = false;   // the reference forces NEAREST on the source for this draw
    d.dstW = needsDownscale ? m_logicalW : width;
    d.dstH = needsDownscale ? m_logicalH : height;
    d.dstKind = rck::PRESENT_RGBX8;
    int vp[4];
    overscanViewport(d.dstW, d.dstH, m_overscanX, m_overscanY, vp);
    d.vpX = vp[0];
    d.vpY = vp[1];
    d.vpW = vp[2];
    d.vpH = vp[3];
    rck::PresentLaunch L;
    if (!grow(&s.devPre, &s.devPreBytes, (size_t)d.dstW * d.dstH * 4, false) ||
        !makePresentLaunch(d, s.devRgba, s.devPre, 1, &L) || !hipOk(rck::launch_present(L, es), "pre-pass"))
      return false;
    chainSrc = s.devPre;
    cw = d.dstW;
    ch = d.dstH;
  }
  const void* out = m_engine->applyShader(chainSrc, cw, ch);
  uint32_t ow = cw, oh = ch;
  const bool shaded = out != chainSrc;  // an inactive engine hands the input back (reference behaviour)
  if (shaded) {
    ow = m_engine->getOutputWidth();
    oh = m_engine->getOutputHeight();
  }
  // output resolution (:413-505) and image adjustments (:739-804), fused with the alpha strip
  const bool resize = m_outW > 0 && m_outH > 0;
  const bool adjust = m_brightness != 1.0f || m_contrast != 1.0f;
  const uint32_t fw = resize ? m_outW : ow, fh = resize ? m_outH : oh;
  const size_t outBytes = (size_t)fw * fh * 3;
  if (!grow(&s.devOut, &s.devOutBytes, outBytes, false) || !grow(&s.hostOut, &s.hostOutBytes, outBytes, true)) return false;
  bool queued;
  if (resize || adjust) {
    PresentDesc d;
    d.srcW = ow;
    d.srcH = oh;
    d.srcRgb = !shaded;      // the captured frame is a GL_RGB texture, NEAREST (FrameProcessor's default filter)
    d.srcLinear = shaded;    // a render target's filter state as created (ShaderEngine.cpp:2896-2899)
    d.dstW = fw;
    d.dstH = fh;
    d.dstKind = rck::PRESENT_RGB24;
    d.outFlipRows = m_flipY;
    if (resize) {
      d.bake = adjust;
      d.bakeBrightness = m_brightness;
      d.bakeContrast = m_contrast;
    } else {
      d.brightness = m_brightness;
      d.contrast = m_contrast;
    }
    rck::PresentLaunch L;
    queued = makePresentLaunch(d, out, s.devOut, 1, &L) && hipOk(rck::launch_present(L, es), "present");
  } else {
    queued = hipOk(rck::launch_egress_rgb24(out, ow, oh, 1, m_flipY ? 1 : 0, s.devOut, es), "egress");
  }
  ow = fw;
  oh = fh;
  if (!queued ||
      !hipOk(hipEventRecord(s.computeDone, es), "event") || !hipOk(hipStreamWaitEvent(m_out, s.computeDone, 0), "wait") ||
      !hipOk(hipMemcpyAsync(s.hostOut, s.devOut, outBytes, hipMemcpyDeviceToHost, m_out), "D2H") ||
      !hipOk(hipEventRecord(s.d2hDone, m_out), "event"))
    return false;
  s.outW = ow;
  s.outH = oh;
  m_head = (m_head + 1) % (int)m_slots.size();
  ++m_inFlight;
  return true;
}

void* FramePipeline::inputBuffer(int pixfmt, uint32_t width, uint32_t height) {
  if (!m_ok || full()) return nullptr;
  const size_t inBytes = frameBytes(pixfmt, width, height);
  Slot& s = m_slots[(size_t)m_head];
  if (inBytes == 0 || !grow(&s.hostIn, &s.hostInBytes, inBytes, true)) return nullptr;
  return s.hostIn;
}

bool FramePipeline::receive(const void** hostRgb24, uint32_t* width, uint32_t* height, bool wait) {
  if (!m_ok || m_inFlight == 0) return false;
  Slot& s = m_slots[(size_t)m_tail];
  if (wait) {
    if (!hipOk(hipEventSynchronize(s.d2hDone), "sync")) return false;
  } else if (hipEventQuery(s.d2hDone) != hipSuccess) {
    return false;
  }
  if (hostRgb24) *hostRgb24 = s.hostOut;
  if (width) *width = s.outW;
  if (height) *height = s.outH;
  m_held = m_tail;   // the caller's until the next receive(): submit() keeps off it (the previous one is released)
  m_tail = (m_tail + 1) % (int)m_slots.size();
  --m_inFlight;
  return true;
}

}  // namespace rc
