"""ctypes mirror of the reference's ShaderEngine API (reference src/shader/ShaderEngine.h:42-98)
over the C ABI in include/rc_shaderchain.h.  Method names and meanings follow the reference
class; frames are device pointers (ints) or objects exposing ``data_ptr()`` (torch tensors).
"""
import ctypes as C
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class RcError(RuntimeError):
    pass


def library_path():
    return os.path.join(_HERE, "librcshaderchain.so")


class _RcParam(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("description", C.c_char * 128), ("value", C.c_float),
                ("default_value", C.c_float), ("min", C.c_float), ("max", C.c_float), ("step", C.c_float)]


class _RcPassInfo(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_int), ("has_kernel", C.c_int),
                ("filter_linear", C.c_int), ("wrap", C.c_int), ("kernel", C.c_char * 48), ("alias", C.c_char * 48)]


class _RcPresentDesc(C.Structure):
    _fields_ = [("src_w", C.c_uint32), ("src_h", C.c_uint32), ("src_rgb", C.c_int), ("src_linear", C.c_int),
                ("dst_w", C.c_uint32), ("dst_h", C.c_uint32), ("dst_kind", C.c_int),
                ("vp_x", C.c_int32), ("vp_y", C.c_int32), ("vp_w", C.c_int32), ("vp_h", C.c_int32), ("flip_y", C.c_int),
                ("brightness", C.c_float), ("contrast", C.c_float), ("clear", C.c_float * 4), ("bake", C.c_int),
                ("bake_brightness", C.c_float), ("bake_contrast", C.c_float), ("out_flip_rows", C.c_int)]


class _RcPassProfile(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("launches", C.c_uint64), ("frames", C.c_uint64),
                ("read_bytes_per_frame", C.c_uint64), ("write_bytes_per_frame", C.c_uint64),
                ("folded", C.c_uint32), ("reserved", C.c_uint32)]


# every symbol include/rc_shaderchain.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("rc_engine_create", C.c_void_p, [C.c_int, C.c_void_p]),
    ("rc_engine_destroy", None, [C.c_void_p]),
    ("rc_engine_load_preset", C.c_int, [C.c_void_p, C.c_char_p]),
    ("rc_engine_load_shader", C.c_int, [C.c_void_p, C.c_char_p]),
    ("rc_engine_preset_path", C.c_size_t, [C.c_void_p, C.c_char_p, C.c_size_t]),
    ("rc_engine_disable", None, [C.c_void_p]),
    ("rc_engine_is_active", C.c_int, [C.c_void_p]),
    ("rc_engine_set_viewport", None, [C.c_void_p, C.c_uint32, C.c_uint32]),
    ("rc_engine_set_max_resolution", None, [C.c_void_p, C.c_uint32, C.c_uint32]),
    ("rc_engine_apply", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p),
                                  C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("rc_engine_apply_batch", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("rc_engine_output_width", C.c_uint32, [C.c_void_p]),
    ("rc_engine_output_height", C.c_uint32, [C.c_void_p]),
    ("rc_engine_sync", C.c_int, [C.c_void_p]),
    ("rc_engine_param_count", C.c_int, [C.c_void_p]),
    ("rc_engine_param_get", C.c_int, [C.c_void_p, C.c_int, C.POINTER(_RcParam)]),
    ("rc_engine_param_set", C.c_int, [C.c_void_p, C.c_char_p, C.c_float]),
    ("rc_engine_set_uniform1", None, [C.c_void_p, C.c_char_p, C.c_float]),
    ("rc_engine_set_uniform2", None, [C.c_void_p, C.c_char_p, C.c_float, C.c_float]),
    ("rc_engine_set_uniform4", None, [C.c_void_p, C.c_char_p, C.c_float, C.c_float, C.c_float, C.c_float]),
    ("rc_engine_save_preset", C.c_int, [C.c_void_p, C.c_char_p]),
    ("rc_engine_pass_count", C.c_int, [C.c_void_p]),
    ("rc_engine_pass_info", C.c_int, [C.c_void_p, C.c_int, C.POINTER(_RcPassInfo)]),
    ("rc_engine_read_pass", C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_size_t]),
    ("rc_engine_set_profiling", None, [C.c_void_p, C.c_int]),
    ("rc_engine_pass_profile", C.c_int, [C.c_void_p, C.c_int, C.POINTER(_RcPassProfile)]),
    ("rc_engine_set_chunk_frames", None, [C.c_void_p, C.c_uint32]),
    ("rc_engine_set_lanes", None, [C.c_void_p, C.c_uint32]),
    ("rc_engine_set_allow_missing_sources", None, [C.c_void_p, C.c_int]),
    ("rc_engine_set_undefined_varying_zero", None, [C.c_void_p, C.c_int]),
    ("rc_engine_set_general_kernels_only", None, [C.c_void_p, C.c_int]),
    ("rc_engine_set_fold_passes", None, [C.c_void_p, C.c_int]),
    ("rc_engine_set_async_table_builds", None, [C.c_void_p, C.c_int]),
    ("rc_engine_set_float_target_fp16", None, [C.c_void_p, C.c_int]),
    ("rc_engine_history_count", C.c_int, [C.c_void_p]),
    ("rc_engine_read_history", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p,
                                         C.c_size_t]),
    ("rc_ingest", C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("rc_egress_rgb24", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]),
    ("rc_pixfmt_frame_bytes", C.c_size_t, [C.c_int, C.c_uint32, C.c_uint32]),
    ("rc_present", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(_RcPresentDesc), C.c_uint32, C.c_void_p]),
    ("rc_present_frame_bytes", C.c_size_t, [C.c_int, C.c_uint32, C.c_uint32]),
    ("rc_overscan_viewport", None, [C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.POINTER(C.c_int32)]),
    ("rc_pipeline_create", C.c_void_p, [C.c_void_p, C.c_int]),
    ("rc_pipeline_destroy", None, [C.c_void_p]),
    ("rc_pipeline_submit", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32]),
    ("rc_pipeline_receive", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_int]),
    ("rc_pipeline_input_buffer", C.c_void_p, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32]),
    ("rc_pipeline_in_flight", C.c_int, [C.c_void_p]),
    ("rc_pipeline_set_flip_y", None, [C.c_void_p, C.c_int]),
    ("rc_pipeline_set_source_prepass", None, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float]),
    ("rc_pipeline_set_output_resolution", None, [C.c_void_p, C.c_uint32, C.c_uint32]),
    ("rc_pipeline_set_image_adjust", None, [C.c_void_p, C.c_float, C.c_float]),
    ("rc_selftest_fastmath", C.c_int, [C.c_int, C.POINTER(C.c_uint64)]),
    ("rc_selftest_copy_rate", C.c_int, [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    ("rc_selftest_srgb8_host", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("rc_selftest_royale_scan_tables", C.c_int, [C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]),
    ("rc_selftest_srgb8_device", C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("rc_selftest_crt_geom_vertex", C.c_int, [C.c_void_p, C.c_void_p]),
    ("rc_selftest_royale_scan_bounds", C.c_int, [C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]),
    ("rc_selftest_srgb8_host_form", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    ("rc_selftest_srgb8_device_form", C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]),
    ("rc_last_error", C.c_char_p, []),
    ("rc_version", C.c_char_p, []),
    ("rc_kernel_list", C.c_size_t, [C.c_char_p, C.c_size_t]),
    ("rc_preset_dump_json", C.c_size_t, [C.c_char_p, C.c_char_p, C.c_size_t]),
    ("rc_preset_save_as", C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]),
    ("rc_shader_params_json", C.c_size_t, [C.c_char_p, C.c_char_p, C.c_size_t]),
    ("rc_png_decode_rgba8", C.c_int, [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
]


def load_library():
    """Loads librcshaderchain.so (built in-tree by __graft_entry__.build()); fails loudly."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise RcError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % path)
        # PyTorch wheels bundle their own libamdhip64; if this process is going to use torch for
        # device memory as well, its HIP runtime must be the one (and only one) that gets loaded,
        # so it is imported first and our library then binds to the already-loaded runtime.
        # A process without torch simply uses /opt/rocm's.
        if os.environ.get("RC_NO_TORCH_PRELOAD") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        lib = C.CDLL(path)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def _ptr(frame):
    if frame is None:
        return None
    if hasattr(frame, "data_ptr"):
        return frame.data_ptr()
    return int(frame)


def _json_call(fn, path):
    buf = C.create_string_buffer(1 << 16)
    n = fn(path.encode(), buf, len(buf))
    if n >= len(buf):
        buf = C.create_string_buffer(n + 1)
        fn(path.encode(), buf, len(buf))
    return json.loads(buf.value.decode())


def preset_dump(path):
    """Parsed preset (passes / textures / params) as the engine's parser sees it."""
    return _json_call(load_library().rc_preset_dump_json, path)


def shader_params(path):
    return _json_call(load_library().rc_shader_params_json, path)


PIXFMT = {"rgb24": 0, "bgra": 1, "rgba": 2, "yuyv422": 3}


def ingest(src, pixfmt, width, height, n_frames, dst_rgba8, stream=None):
    """Device-side pixel-format conversion into the chain's RGBA8 frames (FrameProcessor.cpp:43-222)."""
    rc = load_library().rc_ingest(_ptr(src), PIXFMT[pixfmt], width, height, n_frames, _ptr(dst_rgba8),
                                  C.c_void_p(stream) if stream else None)
    if rc != 0:
        raise RcError("rc_ingest failed (%d)" % rc)


def egress_rgb24(src_rgba8, width, height, n_frames, dst_rgb24, flip_y=False, stream=None):
    """RGBA8 -> RGB24 alpha strip of the readback (FrameCapturePipeline.cpp:1060-1080)."""
    rc = load_library().rc_egress_rgb24(_ptr(src_rgba8), width, height, n_frames, int(bool(flip_y)), _ptr(dst_rgb24),
                                        C.c_void_p(stream) if stream else None)
    if rc != 0:
        raise RcError("rc_egress_rgb24 failed (%d)" % rc)


PRESENT_KIND = {"rgba8": 0, "rgbx8": 1, "rgb24": 2}


def present(src, src_w, src_h, dst, dst_w, dst_h, n_frames=1, src_rgb=False, src_linear=True, dst_kind="rgba8",
            viewport=None, flip_y=False, brightness=1.0, contrast=1.0, clear=(0.0, 0.0, 0.0, 0.0), bake=None,
            out_flip_rows=False, stream=None):
    """OpenGLRenderer::renderTexture off-screen (OpenGLRenderer.cpp:378-470) on device buffers: the source
    pre-pass, the output-resolution resize and the brightness / contrast bake of FrameCapturePipeline.
    bake = (brightness, contrast) of a second draw on the first one's result, fused."""
    d = _RcPresentDesc()
    d.src_w, d.src_h, d.src_rgb, d.src_linear = src_w, src_h, int(bool(src_rgb)), int(bool(src_linear))
    d.dst_w, d.dst_h, d.dst_kind = dst_w, dst_h, PRESENT_KIND[dst_kind]
    if viewport:
        d.vp_x, d.vp_y, d.vp_w, d.vp_h = viewport
    d.flip_y = int(bool(flip_y))
    d.brightness, d.contrast = brightness, contrast
    for k in range(4):
        d.clear[k] = clear[k]
    if bake:
        d.bake, d.bake_brightness, d.bake_contrast = 1, bake[0], bake[1]
    d.out_flip_rows = int(bool(out_flip_rows))
    rc = load_library().rc_present(_ptr(src), _ptr(dst), C.byref(d), n_frames, C.c_void_p(stream) if stream else None)
    if rc != 0:
        raise RcError("rc_present failed (%d)" % rc)


def overscan_viewport(fbo_w, fbo_h, pct_x, pct_y):
    """glViewport (x, y, w, h) of the pre-pass for an overscan crop (FrameCapturePipeline.cpp:205-216).  No GPU needed."""
    vp = (C.c_int32 * 4)()
    load_library().rc_overscan_viewport(fbo_w, fbo_h, pct_x, pct_y, vp)
    return tuple(vp)


class FramePipeline:
    """Host-to-host frame path (upload, ingest, chain, egress, readback) pipelined over `slots` frames."""

    def __init__(self, engine, slots=3):
        self._lib = load_library()
        self._engine = engine          # keeps the engine alive
        self._h = self._lib.rc_pipeline_create(engine._need(), int(slots))
        if not self._h:
            raise RcError("rc_pipeline_create failed: " + self._lib.rc_last_error().decode())

    def close(self):
        if self._h:
            self._lib.rc_pipeline_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit(self, host_array, pixfmt, width, height):
        """host_array: contiguous numpy uint8 array holding one frame.  False when every slot is busy."""
        return self._lib.rc_pipeline_submit(self._h, host_array.ctypes.data, PIXFMT[pixfmt], width, height) == 0

    def receive(self, wait=True):
        """Oldest finished frame as an (h, w, 3) numpy view of pipeline-owned pinned memory (copy it if you
        keep it past the next submits), or None."""
        import numpy as np
        ptr, w, h = C.c_void_p(), C.c_uint32(), C.c_uint32()
        rc = self._lib.rc_pipeline_receive(self._h, C.byref(ptr), C.byref(w), C.byref(h), int(bool(wait)))
        if rc != 0:
            return None
        buf = (C.c_uint8 * (w.value * h.value * 3)).from_address(ptr.value)
        return np.frombuffer(buf, np.uint8).reshape(h.value, w.value, 3)

    def inputBuffer(self, pixfmt, width, height):
        """numpy view of the next slot's pinned staging memory (None if every slot is busy); fill it and
        pass it to submit() to avoid the host-side copy."""
        import numpy as np
        ptr = self._lib.rc_pipeline_input_buffer(self._h, PIXFMT[pixfmt], width, height)
        if not ptr:
            return None
        n = self._lib.rc_pixfmt_frame_bytes(PIXFMT[pixfmt], width, height)
        return np.frombuffer((C.c_uint8 * n).from_address(ptr), np.uint8)

    def inFlight(self):
        return self._lib.rc_pipeline_in_flight(self._h)

    def setFlipY(self, flip):
        self._lib.rc_pipeline_set_flip_y(self._h, int(bool(flip)))

    def setSourcePrepass(self, logical_w=0, logical_h=0, overscan_pct_x=0.0, overscan_pct_y=0.0):
        self._lib.rc_pipeline_set_source_prepass(self._h, logical_w, logical_h, overscan_pct_x, overscan_pct_y)

    def setOutputResolution(self, width=0, height=0):
        self._lib.rc_pipeline_set_output_resolution(self._h, width, height)

    def setImageAdjust(self, brightness=1.0, contrast=1.0):
        self._lib.rc_pipeline_set_image_adjust(self._h, brightness, contrast)


def selftest_fastmath(device=0):
    """Mismatch counts (log2 division, safe-range division, constant divisors) of the device self-test."""
    out = (C.c_uint64 * 3)()
    rc = load_library().rc_selftest_fastmath(int(device), out)
    if rc != 0:
        raise RcError("rc_selftest_fastmath failed (%d)" % rc)
    return list(out)


def copy_rate(device=0, mib=1024, reps=20):
    """GB/s (read + written) of a 16-byte-per-lane grid-stride copy kernel on the device: the box's streaming ceiling."""
    out = C.c_double(0.0)
    rc = load_library().rc_selftest_copy_rate(int(device), int(mib) << 20, int(reps), C.byref(out))
    if rc != 0:
        raise RcError("rc_selftest_copy_rate failed (%d)" % rc)
    return out.value


def srgb8_encode_host(values, form=1):
    """The sRGB8 store of the pass kernels, evaluated on the host through the same per-run table (numpy float32 in);
    form 2: the table the strip kernels use (linear segment included)."""
    import numpy as np
    v = np.ascontiguousarray(values, dtype=np.float32)
    out = np.empty(v.size, dtype=np.uint8)
    rc = load_library().rc_selftest_srgb8_host_form(v.ctypes.data, out.ctypes.data, v.size, int(form))
    if rc != 0:
        raise RcError("rc_selftest_srgb8_host_form failed (%d)" % rc)
    return out.reshape(v.shape)


def srgb8_encode_device(d_values, d_out, n, device=0, stream=0, form=1):
    """The same on the device: d_values (float32) / d_out (uint8) are device pointers or torch tensors."""
    rc = load_library().rc_selftest_srgb8_device_form(int(device), _ptr(d_values), _ptr(d_out), int(n), C.c_void_p(int(stream)), int(form))
    if rc != 0:
        raise RcError("rc_selftest_srgb8_device_form failed (%d)" % rc)


def royale_scan_tables(off):
    """Host-built expansion tables of crt-royale's scanline pass: (A [9, nodes, 4] float32, bound [9, nodes] float32,
    node colour [9, nodes] float32)."""
    import numpy as np
    lib = load_library()
    n = lib.rc_selftest_royale_scan_tables(C.c_float(off), None, None, 0, 0)
    A = np.zeros((9, n, 4), np.float32)
    B = np.zeros((9, n, 2), np.uint32)
    if lib.rc_selftest_royale_scan_tables(C.c_float(off), A.ctypes.data, B.ctypes.data, A.size, B.size) != n:
        raise RcError("rc_selftest_royale_scan_tables failed")
    return A, B[..., 0].copy().view(np.float32), B[..., 1].copy().view(np.float32)


def royale_scan_bounds(off, dists, device=0):
    """The complete tables as the device builds them for the given row distances (needs a GPU): (A [9, nodes, 4] float32 with the
    node values, bound [9, nodes] float32 - the exhaustively measured bound of the expansion at every node)."""
    import numpy as np
    lib = load_library()
    n = lib.rc_selftest_royale_scan_tables(C.c_float(off), None, None, 0, 0)
    d = np.ascontiguousarray(dists, np.float32)
    A = np.zeros((9, n, 4), np.float32)
    B = np.zeros((9, n), np.float32)
    if lib.rc_selftest_royale_scan_bounds(int(device), C.c_float(off), d.ctypes.data, int(d.size), A.ctypes.data, B.ctypes.data, A.size, B.size) != n:
        raise RcError("rc_selftest_royale_scan_bounds failed")
    return A, B


def crt_geom_vertex(params):
    """crt-geom's vertex-stage constants for 17 parameter values: (sinangle.xy, cosangle.xy, stretch.xyz) float32."""
    import numpy as np
    p = np.ascontiguousarray(params, np.float32)
    if p.shape != (17,):
        raise ValueError("17 parameters expected")
    out = np.zeros(7, np.float32)
    if load_library().rc_selftest_crt_geom_vertex(p.ctypes.data, out.ctypes.data) != 0:
        raise RcError("rc_selftest_crt_geom_vertex failed")
    return out


def preset_save_as(preset_path, out_path, custom=None):
    """ShaderPreset::saveAs without an engine; custom: {name: value}.  True on success."""
    custom = custom or {}
    names = (C.c_char_p * max(1, len(custom)))(*[k.encode() for k in custom])
    values = (C.c_float * max(1, len(custom)))(*[float(v) for v in custom.values()])
    return load_library().rc_preset_save_as(str(preset_path).encode(), str(out_path).encode(), names, values, len(custom)) == 0


def kernel_list():
    lib = load_library()
    buf = C.create_string_buffer(1 << 14)
    lib.rc_kernel_list(buf, len(buf))
    return [l for l in buf.value.decode().split("\n") if l]


class ShaderParameter:
    __slots__ = ("name", "value", "defaultValue", "min", "max", "step", "description")

    def __init__(self, p):
        self.name = p.name.decode()
        self.description = p.description.decode()
        self.value, self.defaultValue, self.min, self.max, self.step = p.value, p.default_value, p.min, p.max, p.step

    def __repr__(self):
        return "ShaderParameter(%s=%g [%g..%g] default %g)" % (self.name, self.value, self.min, self.max, self.defaultValue)


class ShaderEngine:
    """Same surface as the reference's ShaderEngine; `init()` needs a HIP device."""

    def __init__(self):
        self._lib = load_library()
        self._h = None

    # -- lifecycle ---------------------------------------------------------------------------
    def init(self, device=-1, stream=None):
        if self._h:
            return True
        self._h = self._lib.rc_engine_create(int(device), C.c_void_p(stream) if stream else None)
        return bool(self._h)

    def shutdown(self):
        if self._h:
            self._lib.rc_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass

    def _need(self):
        if not self._h:
            raise RcError("ShaderEngine not initialized")
        return self._h

    # -- reference API -----------------------------------------------------------------------
    def loadPreset(self, presetPath):
        return self._lib.rc_engine_load_preset(self._need(), str(presetPath).encode()) >= 0

    def loadPresetStatus(self, presetPath):
        """0 ok, 1 loaded with skipped passes, <0 failure (C ABI status)."""
        return self._lib.rc_engine_load_preset(self._need(), str(presetPath).encode())

    def loadShader(self, shaderPath):
        return self._lib.rc_engine_load_shader(self._need(), str(shaderPath).encode()) >= 0

    def getPresetPath(self):
        buf = C.create_string_buffer(4096)
        self._lib.rc_engine_preset_path(self._need(), buf, len(buf))
        return buf.value.decode()

    def setViewport(self, width, height):
        self._lib.rc_engine_set_viewport(self._need(), int(width), int(height))

    def disableShader(self):
        self._lib.rc_engine_disable(self._need())

    def isShaderActive(self):
        return bool(self._lib.rc_engine_is_active(self._need()))

    def getOutputWidth(self):
        return self._lib.rc_engine_output_width(self._need())

    def getOutputHeight(self):
        return self._lib.rc_engine_output_height(self._need())

    def setMaxShaderResolution(self, maxWidth, maxHeight):
        self._lib.rc_engine_set_max_resolution(self._need(), int(maxWidth), int(maxHeight))

    def applyShader(self, inputFrame, width, height):
        """Returns (device_pointer, out_width, out_height).  The pointer equals the input's when
        the engine is inactive / has no usable pass, as in the reference."""
        out, ow, oh = C.c_void_p(), C.c_uint32(), C.c_uint32()
        rc = self._lib.rc_engine_apply(self._need(), _ptr(inputFrame), int(width), int(height), C.byref(out),
                                       C.byref(ow), C.byref(oh))
        if rc < 0:
            raise RcError("applyShader failed (%d): %s" % (rc, self._lib.rc_last_error().decode()))
        return out.value, ow.value, oh.value

    def applyShaderBatch(self, inputFrames, nFrames, width, height, frameStride=0):
        out, ow, oh = C.c_void_p(), C.c_uint32(), C.c_uint32()
        rc = self._lib.rc_engine_apply_batch(self._need(), _ptr(inputFrames), int(nFrames), int(width), int(height),
                                             int(frameStride), C.byref(out), C.byref(ow), C.byref(oh))
        if rc < 0:
            raise RcError("applyShaderBatch failed (%d): %s" % (rc, self._lib.rc_last_error().decode()))
        return out.value, ow.value, oh.value

    def getShaderParameters(self):
        n = self._lib.rc_engine_param_count(self._need())
        out = []
        for i in range(n):
            p = _RcParam()
            if self._lib.rc_engine_param_get(self._h, i, C.byref(p)) == 0:
                out.append(ShaderParameter(p))
        return out

    def setShaderParameter(self, name, value):
        return bool(self._lib.rc_engine_param_set(self._need(), name.encode(), float(value)))

    def setUniform(self, name, *v):
        if len(v) == 1:
            self._lib.rc_engine_set_uniform1(self._need(), name.encode(), *map(float, v))
        elif len(v) == 2:
            self._lib.rc_engine_set_uniform2(self._need(), name.encode(), *map(float, v))
        elif len(v) == 4:
            self._lib.rc_engine_set_uniform4(self._need(), name.encode(), *map(float, v))
        else:
            raise TypeError("setUniform takes 1, 2 or 4 floats")

    def savePreset(self, path):
        return self._lib.rc_engine_save_preset(self._need(), str(path).encode()) == 0

    # -- additions ---------------------------------------------------------------------------
    def sync(self):
        if self._lib.rc_engine_sync(self._need()) != 0:
            raise RcError("device error: " + self._lib.rc_last_error().decode())

    def setChunkFrames(self, n):
        self._lib.rc_engine_set_chunk_frames(self._need(), int(n))

    def setLanes(self, n):
        """2: the second half of every batch runs on a second HIP stream (rc_engine_set_lanes); default 1."""
        self._lib.rc_engine_set_lanes(self._need(), int(n))

    def setAllowMissingSources(self, allow):
        self._lib.rc_engine_set_allow_missing_sources(self._need(), int(bool(allow)))

    def setProfiling(self, on):
        self._lib.rc_engine_set_profiling(self._need(), int(bool(on)))

    def passProfile(self, i):
        p = _RcPassProfile()
        if self._lib.rc_engine_pass_profile(self._need(), int(i), C.byref(p)) != 0:
            raise RcError("passProfile failed: " + self._lib.rc_last_error().decode())
        return {"total_ms": p.total_ms, "launches": p.launches, "frames": p.frames,
                "read_bytes_per_frame": p.read_bytes_per_frame, "write_bytes_per_frame": p.write_bytes_per_frame,
                "folded": bool(p.folded)}

    def setUndefinedVaryingZero(self, zero):
        self._lib.rc_engine_set_undefined_varying_zero(self._need(), int(bool(zero)))

    def historyCount(self):
        return self._lib.rc_engine_history_count(self._need())

    def readHistory(self, k):
        """Host copy of entry k of the frame-history ring (0 = newest), (h, w, 4) uint8."""
        import numpy as np
        w, h = C.c_uint32(), C.c_uint32()
        if self._lib.rc_engine_read_history(self._need(), int(k), C.byref(w), C.byref(h), None, 0) != 0:
            raise RcError("readHistory(%d): no such entry" % k)
        arr = np.empty((h.value, w.value, 4), np.uint8)
        if self._lib.rc_engine_read_history(self._need(), int(k), C.byref(w), C.byref(h), arr.ctypes.data, arr.nbytes) != 0:
            raise RcError("readHistory(%d) failed" % k)
        return arr

    def setFloatTargetFp16(self, on):
        """float_framebuffer targets stored as binary16 (opt-in; default off = RGBA32F, bit-exact)."""
        self._lib.rc_engine_set_float_target_fp16(self._need(), int(bool(on)))

    def setAsyncTableBuilds(self, on):
        """Slow per-geometry table builds on a worker thread (default on); off: the first frame of a new geometry waits."""
        self._lib.rc_engine_set_async_table_builds(self._need(), int(bool(on)))

    def setFoldPasses(self, on):
        """Byte-map passes folded into their consumers (default on); off renders every pass."""
        self._lib.rc_engine_set_fold_passes(self._need(), int(bool(on)))

    def setGeneralKernelsOnly(self, on):
        self._lib.rc_engine_set_general_kernels_only(self._need(), int(bool(on)))

    def passCount(self):
        return self._lib.rc_engine_pass_count(self._need())

    def passInfo(self, i):
        info = _RcPassInfo()
        if self._lib.rc_engine_pass_info(self._need(), int(i), C.byref(info)) != 0:
            raise IndexError(i)
        return {"width": info.width, "height": info.height, "format": {0: "rgba8", 1: "srgb8", 3: "f32", 4: "f16"}[info.format],
                "has_kernel": bool(info.has_kernel), "filter_linear": bool(info.filter_linear),
                "wrap": ["clamp_to_edge", "clamp_to_border", "repeat", "mirrored_repeat"][info.wrap],
                "kernel": info.kernel.decode(), "alias": info.alias.decode()}

    def readPass(self, i, frame=0):
        """Host copy (numpy) of pass i's render target for `frame` of the last batch."""
        import numpy as np
        info = self.passInfo(i)
        dt = {"f32": np.float32, "f16": np.float16}.get(info["format"], np.uint8)
        arr = np.empty((info["height"], info["width"], 4), dt)
        rc = self._lib.rc_engine_read_pass(self._h, int(i), int(frame), arr.ctypes.data, arr.nbytes)
        if rc != 0:
            raise RcError("readPass(%d,%d) failed: %s" % (i, frame, self._lib.rc_last_error().decode()))
        return arr
