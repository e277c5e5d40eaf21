"""crt-geom's vertex stage (sinangle / cosangle / stretch) as the engine's host setup evaluates it, against the oracle
(bit-identical to llvmpipe, tests/golden/f32_crt_geom_*).  Host only: no GPU."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib

DEFAULTS = [2.4, 2.2, 1.6, 1.0, 2.0, 0.03, 1000.0, 0.0, 0.0, 100.0, 100.0, 0.3, 1.0, 0.3, 0.0, 1.0, 1.0]


@pytest.mark.parametrize("seed", range(4))
def test_vertex_constants_match_oracle(seed, rc_lib):
    from retrocapture_amd import engine as eng
    L = oracle_lib.lib()
    rng = np.random.default_rng(seed)
    for k in range(200):
        p = np.array(DEFAULTS, np.float32)
        if seed or k:
            p[2] = np.float32(rng.uniform(0.1, 3.0))      # d
            p[4] = np.float32(rng.uniform(0.1, 10.0))     # R
            p[7] = np.float32(rng.uniform(-0.5, 0.5))     # x_tilt
            p[8] = np.float32(rng.uniform(-0.5, 0.5))     # y_tilt
        want = np.zeros(7, np.float32)
        L.o_crt_geom_vertex(p.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p))
        got = eng.crt_geom_vertex(p)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (p, got, want)
