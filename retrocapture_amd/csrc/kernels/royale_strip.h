// Shared by the crt-royale "strip" kernels (pass_royale_bloom.hip, ...): kernels for the common case where every
// texture coordinate of a pass is separable - the horizontal one a function of the target column only, the
// vertical one of the target row only (an axis-aligned quad: the varying's plane has a zero slope in the other
// direction) - and differs between the two triangles of the quad only by which plane constants apply.
// Then everything a LINEAR / NEAREST clamp-to-edge sampler derives from a coordinate (first texel, weight) is a
// per-column or per-row quantity, computed ONCE per geometry by a small kernel with the samplers' own
// operations (so the results are the sampler's, bit for bit), kept per thread in registers (columns: a thread
// walks a vertical strip of one column) or read as wave-uniform scalars (rows).  The per-pixel work that is left
// is texel fetch, decode and the float lerps / sums themselves, in the GL's order.
#pragma once
#include <cstring>
#include <chrono>
#include <future>
#include <map>
#include <memory>
#include <mutex>

#include "royale_common.h"

namespace rcstrip {
using namespace rcd;

struct LinTap {
  int i0;    // first texel of the pair (may be -1: both indices are clamped to the texture when fetched)
  float w;   // weight of the second texel
};
// sample_linear_f<., WRAP_EDGE> on one axis (rc_device.h linear_coord_edge_pair: the pair (-1, 0) at the left / top edge)
__device__ __forceinline__ LinTap lin_tap(float s, int n) {
  const float u = linear_coord_edge_pair(s, n);
  const float f = __builtin_floorf(u);
  return LinTap{(int)f, u - f};
}
// sample_nearest<., WRAP_EDGE> on one axis
__device__ __forceinline__ int near_tap(float s, int n) { return clampi((int)__builtin_floorf(s * (float)n), 0, n - 1); }

// strip = 64 columns x kRows rows of one frame; strips are numbered row-major inside a frame, frames in order
template <int kRows>
struct StripGrid {
  int W, H, cgs, rss, per_frame, total;
  __device__ __forceinline__ StripGrid(int w, int h, int n_frames)
      : W(w), H(h), cgs((w + 63) >> 6), rss((h + kRows - 1) / kRows), per_frame(cgs * rss), total(per_frame * n_frames) {}
  __device__ __forceinline__ void locate(int strip, int* z, int* xw, int* ys) const {
    *z = strip / per_frame;
    const int rem = strip - *z * per_frame, rs = rem / cgs;
    *xw = (rem - rs * cgs) * 64;
    *ys = rs * kRows;
  }
};

// Everything of a PassLaunch that per-geometry tables can depend on (pointers excluded)
struct GeoKey {
  int device;
  int out[3], src[2], vp[2], flags;
  int tex[1 + kMaxExtra][5];
  Plane plane[kMaxPlanes];
  float params[kMaxParams];
  bool operator<(const GeoKey& o) const { return std::memcmp(this, &o, sizeof(GeoKey)) < 0; }
};
inline bool make_geo_key(const PassLaunch& L, GeoKey* k) {
  std::memset(static_cast<void*>(k), 0, sizeof(*k));
  if (hipGetDevice(&k->device) != hipSuccess) return false;
  k->out[0] = L.out_w; k->out[1] = L.out_h; k->out[2] = L.out_fmt;
  k->src[0] = L.src_w; k->src[1] = L.src_h; k->vp[0] = L.vp_w; k->vp[1] = L.vp_h; k->flags = L.flags & ~RC_FLAG_ASYNC_TABLES;
  for (int i = 0; i <= kMaxExtra; ++i) {
    const Tex& t = i == 0 ? L.in : L.extra[i - 1];
    k->tex[i][0] = t.w; k->tex[i][1] = t.h; k->tex[i][2] = t.fmt; k->tex[i][3] = t.linear; k->tex[i][4] = t.wrap;
  }
  std::memcpy(k->plane, L.plane, sizeof(k->plane));
  std::memcpy(k->params, L.params, sizeof(k->params));
  return true;
}
// planes [first, first + 2 * pairs): even ones horizontal coordinates (no slope in y), odd ones vertical (none in x)
inline bool separable(const PassLaunch& L, int first, int pairs) {
  for (int p = 0; p < pairs; ++p) {
    const Plane &u = L.plane[first + 2 * p], &v = L.plane[first + 2 * p + 1];
    if (u.dy_lo != 0.0f || u.dy_up != 0.0f || v.dx_lo != 0.0f || v.dx_up != 0.0f) return false;
  }
  return true;
}

// Per-geometry device tables, built on first use by `build` (which launches its kernels on the stream it is given and fills
// T->usable; that stream is synchronised before the tables are handed out, so that a launch on ANY stream - the engine's second
// lane - finds them complete).  At most kGeoCacheEntries geometries are kept per table kind: when a new one arrives (a parameter
// being swept, a window being resized) the least recently used one leaves the cache.  Callers hold the tables through a shared
// pointer for as long as they prepare a launch; the device memory is released when the last holder lets go, after a device
// synchronisation (launches that read the tables may still be in flight).
// `slow_build`: the build takes milliseconds (the scanline pass's exhaustive bound sweep: 140 ms).  When the launch asks for it
// (RC_FLAG_ASYNC_TABLES, the engine's default) such a build runs on a worker thread with a stream of its own, off the frame
// path: nullptr is returned - the caller renders with its general form, same bytes - until the tables are ready.  Without the
// flag the call waits for the build.  Such builders must not read memory the launch's engine owns (they outlive the call).
constexpr size_t kGeoCacheEntries = 32;
template <class Tables>
struct GeoCached {
  std::shared_ptr<Tables> tables;
  std::shared_future<void> building;   // valid while / after a worker built the tables
  uint64_t last_use = 0;
};
template <class Tables, class Build>
std::shared_ptr<const Tables> geo_tables(const PassLaunch& L, hipStream_t s, std::mutex& mu, std::map<GeoKey, GeoCached<Tables>>& cache, Build build,
                                         bool slow_build = false) {
  GeoKey key;
  if (!make_geo_key(L, &key)) return nullptr;
  const bool async = slow_build && (L.flags & RC_FLAG_ASYNC_TABLES);
  std::unique_lock<std::mutex> lock(mu);
  static uint64_t clock = 0;
  auto it = cache.find(key);
  if (it != cache.end()) {
    it->second.last_use = ++clock;
    if (it->second.building.valid()) {
      if (async && it->second.building.wait_for(std::chrono::seconds(0)) != std::future_status::ready) return nullptr;
      const std::shared_future<void> f = it->second.building;   // (wait without the lock: the worker does not take it, other callers may)
      const std::shared_ptr<Tables> t = it->second.tables;
      lock.unlock();
      f.wait();
      return t->usable ? t : nullptr;
    }
    return it->second.tables->usable ? it->second.tables : nullptr;
  }
  if (cache.size() >= kGeoCacheEntries) {
    auto victim = cache.end();
    for (auto c = cache.begin(); c != cache.end(); ++c)
      if ((!c->second.building.valid() || c->second.building.wait_for(std::chrono::seconds(0)) == std::future_status::ready) &&
          (victim == cache.end() || c->second.last_use < victim->second.last_use))
        victim = c;
    if (victim != cache.end()) cache.erase(victim);
  }
  GeoCached<Tables> e;
  e.tables = std::shared_ptr<Tables>(new Tables(), [](Tables* t) {
    (void)hipDeviceSynchronize();
    t->release();
    delete t;
  });
  e.last_use = ++clock;
  if (async) {
    // the worker: its own stream on the launch's device; the tables are complete when the future is
    const std::shared_ptr<Tables> t = e.tables;
    const PassLaunch Lc = L;
    const int device = key.device;
    e.building = std::async(std::launch::async, [t, Lc, device, build]() {
                   hipStream_t ws = nullptr;
                   if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ws, hipStreamNonBlocking) != hipSuccess) return;
                   build(Lc, ws, t.get());
                   (void)hipStreamSynchronize(ws);
                   (void)hipStreamDestroy(ws);
                 }).share();
    cache.emplace(key, e);
    return nullptr;
  }
  build(L, s, e.tables.get());
  (void)hipStreamSynchronize(s);
  auto ins = cache.emplace(key, e);
  return ins.first->second.tables->usable ? ins.first->second.tables : nullptr;
}

}  // namespace rcstrip
