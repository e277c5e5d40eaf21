"""Frame ingest / egress kernels (rc_ingest, rc_egress_rgb24) against the oracle's plain-C loops."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib

FMT = {"rgb24": (0, 3), "bgra": (1, 4), "rgba": (2, 4), "yuyv422": (3, 2)}


def o_ingest(src, fmt, n_px):
    L = oracle_lib.lib()
    dst = np.zeros(n_px * 4, np.uint8)
    L.o_ingest.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    L.o_ingest.restype = None
    L.o_ingest(src.ctypes.data, FMT[fmt][0], n_px, dst.ctypes.data)
    return dst


def o_egress(src, w, h, n, flip):
    L = oracle_lib.lib()
    dst = np.zeros(n * h * w * 3, np.uint8)
    L.o_egress_rgb24.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.o_egress_rgb24.restype = None
    L.o_egress_rgb24(src.ctypes.data, w, h, n, int(flip), dst.ctypes.data)
    return dst


def test_oracle_yuyv_known_answers():
    """BT.601 limited range: black, white, and the 75 % colour-bar primaries' published code values."""
    cases = {(16, 128, 128): (0, 0, 0), (235, 128, 128): (255, 255, 255), (81, 90, 240): (255, 0, 0),
             (145, 54, 34): (0, 255, 0), (41, 240, 110): (0, 0, 255), (0, 128, 128): (0, 0, 0), (255, 128, 128): (255, 255, 255)}
    for (y, u, v), want in cases.items():
        got = o_ingest(np.array([y, u, y, v], np.uint8), "yuyv422", 2).reshape(2, 4)
        assert np.abs(got[0, :3].astype(int) - np.array(want)).max() <= 1, ((y, u, v), got[0])
        assert (got[0] == got[1]).all() and got[0, 3] == 255


def test_oracle_rgb_roundtrip():
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, 7 * 5 * 3, dtype=np.uint8)
    rgba = o_ingest(rgb, "rgb24", 35)
    assert (rgba.reshape(-1, 4)[:, 3] == 255).all()
    assert np.array_equal(o_egress(rgba, 7, 5, 1, False), rgb)
    flipped = o_egress(rgba, 7, 5, 1, True).reshape(5, 7, 3)
    assert np.array_equal(flipped[::-1].ravel(), rgb)
    bgra = rng.integers(0, 256, 35 * 4, dtype=np.uint8)
    out = o_ingest(bgra, "bgra", 35).reshape(-1, 4)
    assert np.array_equal(out[:, :3], bgra.reshape(-1, 4)[:, [2, 1, 0]])


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", sorted(FMT))
@pytest.mark.parametrize("w,h,n", [(1920, 1080, 2), (322, 17, 3), (2, 1, 1), (6, 3, 1), (254, 255, 1)])
def test_ingest_matches_oracle(fmt, w, h, n, rc_lib):
    import torch
    from retrocapture_amd import engine
    rng = np.random.default_rng(w * 7 + h)
    src = rng.integers(0, 256, n * h * w * FMT[fmt][1], dtype=np.uint8)
    d_src = torch.from_numpy(src).cuda()
    d_dst = torch.zeros(n * h * w * 4, dtype=torch.uint8, device="cuda")
    engine.ingest(d_src, fmt, w, h, n, d_dst)
    torch.cuda.synchronize()
    assert np.array_equal(d_dst.cpu().numpy(), o_ingest(src, fmt, n * h * w))


@pytest.mark.gpu
@pytest.mark.parametrize("flip", [False, True])
@pytest.mark.parametrize("w,h,n", [(1920, 1080, 2), (322, 17, 3), (3, 5, 2), (1, 1, 1), (256, 224, 1)])
def test_egress_matches_oracle(flip, w, h, n, rc_lib):
    import torch
    from retrocapture_amd import engine
    rng = np.random.default_rng(w + h * 3)
    src = rng.integers(0, 256, n * h * w * 4, dtype=np.uint8)
    d_src = torch.from_numpy(src).cuda()
    d_dst = torch.zeros(n * h * w * 3, dtype=torch.uint8, device="cuda")
    engine.egress_rgb24(d_src, w, h, n, d_dst, flip_y=flip)
    torch.cuda.synchronize()
    assert np.array_equal(d_dst.cpu().numpy(), o_egress(src, w, h, n, flip))


@pytest.mark.gpu
def test_ingest_chain_egress_roundtrip(preset_tree, rc_lib):
    """RGB24 -> ingest -> stock chain at 1:1 nearest -> egress == the input bytes (the whole frame path)."""
    import torch
    from gpu_util import make_engine
    from retrocapture_amd import engine
    w, h = 96, 40
    rgb = np.random.default_rng(3).integers(0, 256, h * w * 3, dtype=np.uint8)
    d_rgb = torch.from_numpy(rgb).cuda()
    d_rgba = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
    engine.ingest(d_rgb, "rgb24", w, h, 1, d_rgba)
    e = make_engine(preset_tree["stock"], w, h)
    ptr, ow, oh = e.applyShader(d_rgba, w, h)
    assert (ow, oh) == (w, h)
    d_out = torch.zeros(h * w * 3, dtype=torch.uint8, device="cuda")
    engine.egress_rgb24(ptr, w, h, 1, d_out)
    e.sync()
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), rgb)
    e.shutdown()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["rgb24", "bgra", "yuyv422"])
def test_frame_pipeline_equals_direct_path(fmt, preset_tree, rc_lib):
    """The pipelined host-to-host path (3 slots, 3 streams) returns, in order, exactly what ingest ->
    chain -> egress gives frame by frame; crt-pi upscaling 2x, 7 frames through 3 slots."""
    import torch
    from gpu_util import make_engine
    from retrocapture_amd import engine
    w, h, vw, vh, n = 64, 48, 128, 96, 7
    rng = np.random.default_rng(17)
    frames = [rng.integers(0, 256, h * w * FMT[fmt][1], dtype=np.uint8) for _ in range(n)]
    e = make_engine(preset_tree["crt-pi"], vw, vh)
    want = []
    d_rgba = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
    d_out = torch.zeros(vh * vw * 3, dtype=torch.uint8, device="cuda")
    for f in frames:
        engine.ingest(torch.from_numpy(f).cuda(), fmt, w, h, 1, d_rgba)
        ptr, ow, oh = e.applyShader(d_rgba, w, h)
        engine.egress_rgb24(ptr, ow, oh, 1, d_out)
        e.sync()
        torch.cuda.synchronize()
        want.append(d_out.cpu().numpy().reshape(oh, ow, 3).copy())
    e2 = make_engine(preset_tree["crt-pi"], vw, vh)
    pipe = engine.FramePipeline(e2, slots=3)
    got = []
    for f in frames:
        while not pipe.submit(f, fmt, w, h):
            got.append(pipe.receive(wait=True).copy())
    while pipe.inFlight():
        got.append(pipe.receive(wait=True).copy())
    assert len(got) == n
    for k in range(n):
        assert np.array_equal(got[k], want[k]), "frame %d" % k
    assert pipe.receive(wait=False) is None
    pipe.close()
    e.shutdown()
    e2.shutdown()


@pytest.mark.gpu
def test_frame_pipeline_ownership_and_engine_lifetime(preset_tree, rc_lib):
    """A received frame stays intact while further frames are submitted (its slot is not reused before the next
    receive), and shutting the engine down before the pipeline is closed is safe: the pipeline lets go of it."""
    from gpu_util import make_engine
    from retrocapture_amd import engine
    w, h = 64, 48
    rng = np.random.default_rng(3)
    frames = [rng.integers(0, 256, h * w * 3, dtype=np.uint8) for _ in range(6)]
    e = make_engine(preset_tree["stock"], w, h)
    pipe = engine.FramePipeline(e, slots=2)
    assert pipe.submit(frames[0], "rgb24", w, h)
    first = pipe.receive(wait=True)           # a view of the pipeline's pinned memory: ours until the next receive
    keep = first.copy()
    assert pipe.submit(frames[1], "rgb24", w, h)
    assert not pipe.submit(frames[2], "rgb24", w, h)      # 2 slots: one held by us, one in flight
    assert np.array_equal(first, keep)                    # ... and the held frame was not overwritten
    second = pipe.receive(wait=True).copy()
    assert not np.array_equal(second, keep)
    assert pipe.submit(frames[2], "rgb24", w, h)
    e.shutdown()                                          # engine first: the pipeline drains and detaches
    assert not pipe.submit(frames[3], "rgb24", w, h)
    assert pipe.receive(wait=True) is None
    pipe.close()
