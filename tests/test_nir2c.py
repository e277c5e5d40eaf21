"""oracle/glrun/nir2c.py, the generator behind the instruction-list passes (crt-royale's curved last pass, lcd-grid, lcd-grid-v2): a small
hand-written listing in Mesa's print format - registers, an if / else, tex, texelFetch with an offset, UBO loads, a select - is turned
into C, compiled, and must compute what the listing says.  Pure CPU."""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = r"""
#include <math.h>
#include <string.h>
#define RCN_FN
static float RCN_BITS(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
#define RCN_FLOOR(x) floorf(x)
#define RCN_F2I(x) ((int)(x))
#define RCN_MAX(a, b) ((a) > (b) ? (a) : (b))
/* tex: (u, v, u + v, 1); texelFetch: (x, y, x * 10 + y, 1) */
#define RCN_TEX(ctx, unit, u, v, dst) do { dst[0] = (u); dst[1] = (v); dst[2] = (u) + (v); dst[3] = 1.0f; } while (0)
#define RCN_TXF(ctx, unit, x, y, dst) do { dst[0] = (float)(x); dst[1] = (float)(y); dst[2] = (float)((x) * 10 + (y)); dst[3] = 1.0f; } while (0)
#include "sample.inc"
"""


def test_generated_c_computes_what_the_listing_says(tmp_path):
    listing = os.path.join(ROOT, "tests", "data", "nir_listing_sample.txt")
    inc = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "glrun", "nir2c.py"), listing, "--stage", "fragment", "--name", "sample_fs"],
                         check=True, capture_output=True, text=True).stdout
    assert "sample_fs_uniforms" in inc and '{"gain", 2, 1, 0}' in inc and '{"TEX0#0", 1, 1, 0}' in inc
    (tmp_path / "sample.inc").write_text(inc)
    (tmp_path / "h.c").write_text(HARNESS)
    so = str(tmp_path / "sample.so")
    subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-ffp-contract=off", "-o", so, str(tmp_path / "h.c"), "-lm"], check=True, cwd=tmp_path)
    fn = ctypes.CDLL(so).sample_fs
    fn.argtypes = [ctypes.c_void_p] * 4
    for u, v, gain in ((0.75, 0.25, 2.0), (0.3, 0.6, 2.0), (0.9, 0.1, -1.0)):
        U = np.array([8.0, 4.0, gain], np.float32)
        IN = np.array([u, v], np.float32)
        OUT = np.zeros(4, np.float32)
        fn(U.ctypes.data, IN.ctypes.data, OUT.ctypes.data, None)
        taken = np.float32(0.5) < np.float32(u)
        reg = np.float32(max(np.float32(u) * np.float32(gain), np.float32(0))) if taken else np.float32(v)
        second = np.float32(0.5) + -reg
        i = int(np.floor(np.float32(u) * np.float32(8.0)))
        third = np.float32((i + 1) * 10 + i) if taken else second
        assert np.array_equal(OUT, np.array([reg, second, third, 0.5], np.float32)), (u, v, gain, OUT)
