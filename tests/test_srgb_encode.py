"""The sRGB8 store (GL_FRAMEBUFFER_SRGB, reference ShaderEngine.cpp:944-952): llvmpipe's conversion restated.

oracle/rc_sampler.c o_store_srgb8 is pinned against the GL for EVERY float in [0,1] by
oracle/probes/srgb_encode_sweep.py (build container only; 0 mismatches).  Here: the product's per-run table
(csrc/srgb_encode.cpp) and the device function of the pass kernels against that oracle, on every float around
every place the byte changes, every run boundary, the segment boundary, and a few million random floats."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib


def oracle_encode(v):
    lib = oracle_lib.lib()
    lib.o_store_srgb8_array.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    v = np.ascontiguousarray(v, dtype=np.float32)
    out = np.empty(v.size, dtype=np.uint8)
    lib.o_store_srgb8_array(v.ctypes.data, out.ctypes.data, v.size)
    return out


def probe_values():
    rng = np.random.default_rng(11)
    one = 0x3F800000
    lin = int(np.float32(0.0031308).view(np.uint32))
    parts = [rng.integers(0, one + 1, 1 << 22, dtype=np.uint32),                 # uniform over the float bits of [0,1]
             (rng.random(1 << 22, dtype=np.float32)).view(np.uint32),             # uniform over the values
             np.arange(lin - 4096, lin + 4096, dtype=np.uint32),                  # linear / power segment boundary
             np.arange(one - 70000, one + 64, dtype=np.uint32),                   # up to and past 1.0
             np.array([0, 1, 0x007FFFFF, 0x00800000, 0x7F800000, 0xFF800000, 0x7FC00000, 0x80000000, 0xBF800000,
                       0x40000000], dtype=np.uint32)]
    # every RSQRTPS run boundary of the argument between the segment boundary and 1.0, +-2 floats
    runs = np.arange((lin >> 13), (one >> 13) + 1, dtype=np.uint32) << 13
    parts.append((runs[:, None] + np.arange(-2, 3, dtype=np.int64)[None, :]).astype(np.uint32).reshape(-1))
    # around every float where the oracle's byte changes: found by bisection on a coarse grid, then +-48 floats
    grid = np.arange(lin, one + 1, 64, dtype=np.uint32)
    eb = oracle_encode(grid.view(np.float32))
    change = np.nonzero(eb[1:] != eb[:-1])[0]
    around = (grid[change][:, None].astype(np.int64) + np.arange(-48, 112, dtype=np.int64)[None, :]).astype(np.uint32)
    parts.append(around.reshape(-1))
    return np.concatenate(parts).view(np.float32)


def probe_values_linear_segment():
    """The second form of the table also covers the linear segment (byte = rint(x * 12.92 * 255) up to 0.0031308): every float
    within +-40 of each of its ten crossings, every run boundary +-2, the clamp's lower end, and random floats below the segment boundary."""
    rng = np.random.default_rng(12)
    lin = int(np.float32(0.0031308).view(np.uint32))
    lo = 0x391D4000
    parts = [rng.integers(0, lin + 1, 1 << 21, dtype=np.uint32), np.arange(lo - 20000, lo + 20000, dtype=np.uint32)]
    runs = np.arange(lo >> 13, (lin >> 13) + 2, dtype=np.uint32) << 13
    parts.append((runs[:, None] + np.arange(-2, 3, dtype=np.int64)[None, :]).astype(np.uint32).reshape(-1))
    for k in range(11):
        x = np.float32((k + 0.5) / (12.92 * 255.0))
        b = int(x.view(np.uint32))
        parts.append(np.arange(b - 40, b + 41, dtype=np.uint32))
    return np.concatenate(parts).view(np.float32)


def test_oracle_known_answers():
    v = np.array([0.0, 1.0, 2.0, -1.0, np.nan, 0.0031308, 0.5, 0.2], dtype=np.float32)
    got = oracle_encode(v)
    assert list(got[:5]) == [0, 255, 255, 0, 0]
    # the GL's approximation stays within one step of the textbook encode (188, 124)
    assert got[5] == 10 and abs(int(got[6]) - 188) <= 1 and abs(int(got[7]) - 124) <= 1


def test_run_table_matches_oracle(rc_lib):
    from retrocapture_amd import engine
    v = probe_values()
    assert np.array_equal(engine.srgb8_encode_host(v), oracle_encode(v))


def test_second_form_run_table_matches_oracle(rc_lib):
    from retrocapture_amd import engine
    v = np.concatenate([probe_values(), probe_values_linear_segment()])
    assert np.array_equal(engine.srgb8_encode_host(v, form=2), oracle_encode(v))


def test_second_form_linear_segment_exhaustive(rc_lib):
    """Every float of every run of the linear segment (36 M values) through the second form of the table."""
    from retrocapture_amd import engine
    lin = int(np.float32(0.0031308).view(np.uint32))
    lo = 0x391D4000 - 8192
    for start in range(lo, lin + 8192, 1 << 23):
        v = np.arange(start, min(start + (1 << 23), lin + 8192), dtype=np.uint32).view(np.float32)
        assert np.array_equal(engine.srgb8_encode_host(v, form=2), oracle_encode(v))


@pytest.mark.gpu
def test_device_second_form_matches_oracle(rc_lib):
    import torch
    from retrocapture_amd import engine
    v = np.concatenate([probe_values(), probe_values_linear_segment()])
    d = torch.from_numpy(v.copy()).cuda()
    o = torch.empty(v.size, dtype=torch.uint8, device="cuda")
    engine.srgb8_encode_device(d, o, v.size, stream=torch.cuda.current_stream().cuda_stream, form=2)
    torch.cuda.synchronize()
    assert np.array_equal(o.cpu().numpy(), oracle_encode(v))


@pytest.mark.gpu
def test_device_encode_matches_oracle(rc_lib):
    import torch
    from retrocapture_amd import engine
    v = probe_values()
    d = torch.from_numpy(v.copy()).cuda()
    o = torch.empty(v.size, dtype=torch.uint8, device="cuda")
    engine.srgb8_encode_device(d, o, v.size, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(o.cpu().numpy(), oracle_encode(v))


@pytest.mark.gpu
def test_device_encode_through_the_lds_copy_matches_oracle(rc_lib):
    """Form 3: the first form of the table copied into every workgroup's LDS, which is how every large sRGB8 target encodes
    (form 1 on the device reads the table in device memory, as the 320 x 240 passes do)."""
    import torch
    from retrocapture_amd import engine
    v = np.concatenate([probe_values(), probe_values_linear_segment()])
    d = torch.from_numpy(v.copy()).cuda()
    o = torch.empty(v.size, dtype=torch.uint8, device="cuda")
    engine.srgb8_encode_device(d, o, v.size, stream=torch.cuda.current_stream().cuda_stream, form=3)
    torch.cuda.synchronize()
    assert np.array_equal(o.cpu().numpy(), oracle_encode(v))
