"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE: the checker, never the product)."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None
_THREADS = 1


def set_threads(n):
    """Rows of every pass are split over n threads (ctypes releases the GIL)."""
    global _THREADS
    _THREADS = max(1, int(n))

FMT = {"rgba8": 0, "srgb8": 1, "rgbx8": 2, "f32": 3}
WRAP = {"clamp_to_edge": 0, "clamp_to_border": 1, "repeat": 2, "mirrored_repeat": 3}


class OTex(C.Structure):
    _fields_ = [("data", C.c_void_p), ("w", C.c_int), ("h", C.c_int), ("fmt", C.c_int),
                ("linear", C.c_int), ("wrap", C.c_int), ("n_levels", C.c_int), ("mip", C.c_void_p * 15)]


class OPassArgs(C.Structure):
    _fields_ = [("inp", C.POINTER(OTex)), ("extra", C.POINTER(OTex) * 8), ("src_w", C.c_int),
                ("src_h", C.c_int), ("out_w", C.c_int), ("out_h", C.c_int), ("out_fmt", C.c_int),
                ("frame_count", C.c_int), ("params", C.POINTER(C.c_float)), ("dst", C.c_void_p),
                ("y0", C.c_int), ("y1", C.c_int), ("pass_index", C.c_int), ("n_passes", C.c_int),
                ("chain_w", C.c_int * 16), ("chain_h", C.c_int * 16), ("vp_w", C.c_int), ("vp_h", C.c_int),
                ("flags", C.c_int), ("uni_tex_w", C.c_int), ("uni_tex_h", C.c_int), ("uni_out_w", C.c_int), ("uni_out_h", C.c_int)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
        _LIB = C.CDLL(path)
        for f in ("o_exp2", "o_log2", "o_exp", "o_log", "o_sin", "o_cos"):
            getattr(_LIB, f).restype = C.c_float
            getattr(_LIB, f).argtypes = [C.c_float]
        _LIB.o_pow.restype = C.c_float
        _LIB.o_pow.argtypes = [C.c_float, C.c_float]
    return _LIB


class Tex:
    """A texture as a pass sees it: array + format + the sampler state set on it."""

    def __init__(self, arr, fmt, linear, wrap, mipmap=False):
        self.arr = np.ascontiguousarray(arr)
        self.c = OTex(self.arr.ctypes.data, self.arr.shape[1], self.arr.shape[0], FMT[fmt],
                      int(bool(linear)), WRAP.get(wrap, 0))
        self.fmt = fmt
        self.levels = [self.arr]
        if mipmap:      # mipmap_input: the chain llvmpipe's glGenerateMipmap builds (oracle/rc_sampler.c)
            self.levels = gen_mipmaps(self.arr, fmt)
            self.c.n_levels = len(self.levels)
            for k, l in enumerate(self.levels):
                self.c.mip[k] = l.ctypes.data


def gen_mipmaps(level0, fmt):
    """[level 0, level 1, ...] as o_gen_mipmaps builds them (sRGB8 / F32 textures)."""
    L = lib()
    h, w = level0.shape[:2]
    L.o_mip_levels.restype = C.c_int
    n = L.o_mip_levels(w, h)
    levels = [np.ascontiguousarray(level0)]
    for k in range(1, n):
        levels.append(np.zeros((max(1, h >> k), max(1, w >> k), 4), level0.dtype))
    ptrs = (C.c_void_p * n)(*[l.ctypes.data for l in levels])
    L.o_gen_mipmaps.restype = None
    L.o_gen_mipmaps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_int]
    L.o_gen_mipmaps(levels[0].ctypes.data, w, h, FMT[fmt], ptrs, n)
    return levels


def _call(name, tex, out_w, out_h, y0, y1, out_fmt, params, frame_count, extra, src_w, src_h, chain, pass_index,
          vp, flags, threads, uni):
    L = lib()
    fn = getattr(L, "o_pass_" + name)
    fn.restype = None
    fn.argtypes = [C.POINTER(OPassArgs)]
    dst = np.zeros((out_h, out_w, 4), np.float32 if out_fmt == "f32" else np.uint8)
    p = (C.c_float * max(1, len(params)))(*params)
    a = OPassArgs()
    a.inp = C.pointer(tex.c)
    for i, e in enumerate(extra):
        a.extra[i] = C.pointer(e.c)
    a.src_w = src_w or tex.c.w
    a.src_h = src_h or tex.c.h
    a.out_w, a.out_h, a.out_fmt = out_w, out_h, FMT[out_fmt]
    a.frame_count = frame_count
    a.params = p
    a.dst = dst.ctypes.data
    a.pass_index = pass_index
    a.n_passes = len(chain) if chain else 1
    for k, (cw, ch) in enumerate(chain or []):
        a.chain_w[k], a.chain_h[k] = cw, ch
    a.vp_w, a.vp_h = vp if vp else (out_w, out_h)
    a.flags = flags
    if uni:
        a.uni_tex_w, a.uni_tex_h, a.uni_out_w, a.uni_out_h = uni
    rows = y1 - y0
    if threads > 1 and rows >= 4 * threads:
        import threading
        ths = []
        for k in range(threads):
            b = OPassArgs()
            C.memmove(C.byref(b), C.byref(a), C.sizeof(OPassArgs))
            b.y0, b.y1 = y0 + rows * k // threads, y0 + rows * (k + 1) // threads
            t = threading.Thread(target=fn, args=(C.byref(b),))
            t.start()
            ths.append((t, b))
        for t, _ in ths:
            t.join()
    else:
        a.y0, a.y1 = y0, y1
        fn(C.byref(a))
    return dst


def run_pass(name, tex, out_w, out_h, out_fmt="rgba8", params=(), frame_count=1, extra=(), src_w=None, src_h=None,
             chain=None, pass_index=0, vp=None, flags=0, uni=None):
    """Render one pass with the oracle; returns (out_h, out_w, 4) uint8 or float32.  uni = (tex_w, tex_h, out_w, out_h): the
    size uniforms as the shader reads them where they are not the real sizes (o_pass_args::uni_*)."""
    return _call(name, tex, out_w, out_h, 0, out_h, out_fmt, params, frame_count, extra, src_w, src_h, chain,
                 pass_index, vp, flags, _THREADS, uni)


def run_pass_rows(name, tex, out_w, out_h, y0, y1, out_fmt="rgba8", params=(), frame_count=1, extra=(),
                  src_w=None, src_h=None, chain=None, pass_index=0, vp=None, flags=0):
    """Rows [y0, y1) of a pass rendered at full target size (for full-size spot checks)."""
    return _call(name, tex, out_w, out_h, y0, y1, out_fmt, params, frame_count, extra, src_w, src_h, chain,
                 pass_index, vp, flags, 1, None)[y0:y1].copy()


class OVec4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class OPresentArgs(C.Structure):
    _fields_ = [("src", C.POINTER(OTex)), ("dst_w", C.c_int), ("dst_h", C.c_int), ("dst_fmt", C.c_int),
                ("vp_x", C.c_int), ("vp_y", C.c_int), ("vp_w", C.c_int), ("vp_h", C.c_int), ("flip_y", C.c_int),
                ("brightness", C.c_float), ("contrast", C.c_float), ("clear", OVec4), ("dst", C.c_void_p)]


def present(tex, dst_w, dst_h, dst_fmt="rgba8", vp=None, flip_y=False, brightness=1.0, contrast=1.0,
            clear=(0.0, 0.0, 0.0, 0.0)):
    """OpenGLRenderer::renderTexture off-screen (oracle/rc_present.c); returns (dst_h, dst_w, 4)."""
    L = lib()
    L.o_present.restype = None
    L.o_present.argtypes = [C.POINTER(OPresentArgs)]
    dst = np.zeros((dst_h, dst_w, 4), np.float32 if dst_fmt == "f32" else np.uint8)
    a = OPresentArgs()
    a.src = C.pointer(tex.c)
    a.dst_w, a.dst_h, a.dst_fmt = dst_w, dst_h, FMT[dst_fmt]
    a.vp_x, a.vp_y, a.vp_w, a.vp_h = vp if vp else (0, 0, dst_w, dst_h)
    a.flip_y = int(bool(flip_y))
    a.brightness, a.contrast = brightness, contrast
    a.clear = OVec4(*clear)
    a.dst = dst.ctypes.data
    L.o_present(C.byref(a))
    return dst


def overscan_viewport(fbo_w, fbo_h, pct_x, pct_y):
    L = lib()
    L.o_overscan_viewport.restype = None
    L.o_overscan_viewport.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_int)]
    vp = (C.c_int * 4)()
    L.o_overscan_viewport(fbo_w, fbo_h, pct_x, pct_y, vp)
    return tuple(vp)
