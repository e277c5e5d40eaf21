"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/rc_shaderchain.h declares; host-only entry points work; GPU entry points fail loudly."""
import ctypes as C
import os
import re

import pytest

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


def _decode_png(rc_lib, path, w, h):
    import numpy as np
    buf = np.zeros(w * h * 4, np.uint8)
    ww, hh = C.c_int(), C.c_int()
    rc_lib.rc_png_decode_rgba8.restype = C.c_int
    rc_lib.rc_png_decode_rgba8.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    assert rc_lib.rc_png_decode_rgba8(str(path).encode(), buf.ctypes.data, buf.nbytes, C.byref(ww), C.byref(hh)) == 0
    assert (ww.value, hh.value) == (w, h)
    return buf.reshape(h, w, 4)


def test_png_decoder_palette_fixture(rc_lib):
    """A 4-bit palette PNG with a tRNS chunk (png_set_palette_to_rgb + png_set_tRNS_to_alpha in the reference,
    ShaderEngine.cpp:2612-2627) against the RGBA array an independent decoder produced for the same file."""
    import numpy as np
    want = np.load(os.path.join(ROOT, "tests", "golden", "png_palette4_trns_9x7.npy"))
    got = _decode_png(rc_lib, os.path.join(ROOT, "tests", "golden", "png_palette4_trns_9x7.png"), 9, 7)
    assert np.array_equal(got, want)


def test_png_decoder_colour_types(rc_lib, tmp_path):
    """Every 8-bit-or-less colour type the reference's libpng transforms accept, written by PIL and decoded by PIL's own
    reader as the expected value: palette 1/2/4/8 bit (with and without tRNS), grey 1/2/4/8, grey+alpha, RGB, RGBA."""
    import numpy as np
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(9)
    w, h = 13, 6   # odd width: sub-byte rows end mid-byte
    cases = []
    for bits in (1, 2, 4, 8):
        n = 1 << bits
        idx = rng.integers(0, n, (h, w), dtype=np.uint8)
        pal = rng.integers(0, 256, (n, 3), dtype=np.uint8)
        for trns in (False, True):
            im = Image.fromarray(idx, "P")
            im.putpalette(pal.tobytes())
            kw = {"bits": bits}
            if trns:
                kw["transparency"] = bytes(rng.integers(0, 256, n, dtype=np.uint8).tolist())
            cases.append(("pal%d%s" % (bits, "t" if trns else ""), im, kw))
    cases.append(("grey8", Image.fromarray(rng.integers(0, 256, (h, w), dtype=np.uint8), "L"), {}))
    cases.append(("grey1", Image.fromarray(rng.integers(0, 2, (h, w), dtype=np.uint8) * 255, "L").convert("1"), {}))
    cases.append(("la", Image.fromarray(rng.integers(0, 256, (h, w, 2), dtype=np.uint8), "LA"), {}))
    cases.append(("rgb", Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB"), {}))
    cases.append(("rgba", Image.fromarray(rng.integers(0, 256, (h, w, 4), dtype=np.uint8), "RGBA"), {}))
    for name, im, kw in cases:
        path = tmp_path / (name + ".png")
        im.save(path, **kw)
        want = np.array(Image.open(path).convert("RGBA"))
        got = _decode_png(rc_lib, path, w, h)
        assert np.array_equal(got, want), name
