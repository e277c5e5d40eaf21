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
#include <map>
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
  k->src[0] = L.src_w; k->src[1] = L.src_h; k->vp[0] = L.vp_w; k->vp[1] = L.vp_h; k->flags = L.flags;
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

// Per-geometry device tables, built on first use by `build` (which launches its kernels on `s`, synchronises once
// to read back whether the geometry qualifies, and fills T->usable).  At most kGeoCacheEntries geometries are kept per
// table kind: when a new one arrives (a parameter being swept, a window being resized) the least recently used one is
// released - after a device synchronisation, since launches that read its tables may still be in flight.
constexpr size_t kGeoCacheEntries = 32;
template <class Tables>
struct GeoCached {
  Tables tables;
  uint64_t last_use = 0;
};
template <class Tables, class Build>
const Tables* geo_tables(const PassLaunch& L, hipStream_t s, std::mutex& mu, std::map<GeoKey, GeoCached<Tables>>& cache, Build build) {
  GeoKey key;
  if (!make_geo_key(L, &key)) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  static uint64_t clock = 0;
  auto it = cache.find(key);
  if (it != cache.end()) {
    it->second.last_use = ++clock;
    return it->second.tables.usable ? &it->second.tables : nullptr;
  }
  if (cache.size() >= kGeoCacheEntries) {
    auto victim = cache.begin();
    for (auto c = cache.begin(); c != cache.end(); ++c)
      if (c->second.last_use < victim->second.last_use) victim = c;
    (void)hipDeviceSynchronize();
    victim->second.tables.release();
    cache.erase(victim);
  }
  GeoCached<Tables> e;
  build(L, s, &e.tables);
  e.last_use = ++clock;
  auto ins = cache.emplace(key, e);
  return ins.first->second.tables.usable ? &ins.first->second.tables : nullptr;
}

}  // namespace rcstrip
