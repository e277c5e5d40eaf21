"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/rc_shaderchain.h declares; host-only entry points work; GPU entry points fail loudly."""
import ctypes as C
import os
import re

from conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rc_shaderchain.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(rc_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_symbols_are_exported(rc_lib):
    from retrocapture_amd import engine
    names = declared_functions()
    assert len(names) >= 30
    bound = {n for n, _, _ in engine.SYMBOLS}
    for n in names:
        assert hasattr(rc_lib, n), "missing export " + n
        assert n in bound, "python mirror does not bind " + n
    assert bound <= set(names), "python binds symbols the header does not declare: %r" % (bound - set(names))


def test_host_only_entry_points(rc_lib, tmp_path):
    from retrocapture_amd import engine
    assert b"gfx950" in rc_lib.rc_version()
    ks = engine.kernel_list()
    assert "crt/shaders/crt-pi.glsl" in ks and "blurs/blur9fast-vertical.glsl" in ks and len(ks) >= 15
    d = engine.preset_dump(str(tmp_path / "missing.glslp"))
    assert d["ok"] is False


def test_no_cpu_fallback(rc_lib):
    """Without a HIP device the engine refuses to initialise (no silent CPU path)."""
    import torch
    if torch.cuda.is_available():
        return
    h = rc_lib.rc_engine_create(-1, None)
    assert not h
    assert b"no HIP device" in rc_lib.rc_last_error()


def test_png_decoder_matches_fixture(rc_lib):
    """LUT PNG decode (zlib + unfilter) against the committed RGBA8 array of the same file."""
    import numpy as np
    png = os.path.join(ROOT, "tests", "golden", "lut_mask_slot_small_64.png")
    want = np.load(os.path.join(ROOT, "tests", "golden", "lut_mask_slot_small_64.npy"))
    buf = np.zeros(64 * 64 * 4, np.uint8)
    w, h = C.c_int(), C.c_int()
    rc_lib.rc_png_decode_rgba8.restype = C.c_int
    rc_lib.rc_png_decode_rgba8.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    assert rc_lib.rc_png_decode_rgba8(png.encode(), buf.ctypes.data, buf.nbytes, C.byref(w), C.byref(h)) == 0
    assert (w.value, h.value) == (64, 64)
    assert np.array_equal(buf.reshape(64, 64, 4), want)
