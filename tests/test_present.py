"""rc_present - OpenGLRenderer::renderTexture off-screen (reference OpenGLRenderer.cpp:378-470), i.e. the
source pre-pass, output-resolution resize and brightness / contrast bake of FrameCapturePipeline - against
goldens rendered on llvmpipe (tests/golden/present_*.npz, oracle/glrun/glpresent.cpp) and the oracle."""
import glob
import os

import numpy as np
import pytest

import oracle_lib
from oracle_lib import Tex

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "present_*.npz")))


def rgba_of(src):
    if src.shape[2] == 4:
        return src, False
    return np.concatenate([src, np.full(src.shape[:2] + (1,), 255, np.uint8)], -1), True


def oracle_case(g):
    src, is_rgb = rgba_of(g["src"])
    linear = str(g["filt"]) == "linear"
    dh, dw = g["out"].shape[:2]
    t = Tex(src, "rgbx8" if is_rgb else "rgba8", linear, "clamp_to_edge")
    o = oracle_lib.present(t, dw, dh, str(g["dst"]), tuple(int(v) for v in g["vp"]), bool(int(g["flip"])), float(g["b"]), float(g["c"]))
    if "bake" in g.files:
        bb, bc = (float(v) for v in g["bake"])
        o = oracle_lib.present(Tex(o, "rgba8", True, "clamp_to_edge"), dw, dh, "rgba8", None, False, bb, bc, clear=(0, 0, 0, 1))
    return o


def test_goldens_present():
    assert len(CASES) >= 9


@pytest.mark.parametrize("case", CASES)
def test_oracle_present_matches_llvmpipe(case):
    g = np.load(os.path.join(GOLD, case + ".npz"))
    o = oracle_case(g)
    assert o.dtype == g["out"].dtype
    assert np.array_equal(o, g["out"]), case      # bit-exact, float target included
    if "mid" in g.files:   # the resize draw of the resize -> bake chain
        src, _ = rgba_of(g["src"])
        dh, dw = g["mid"].shape[:2]
        assert np.array_equal(oracle_lib.present(Tex(src, "rgba8", True, "clamp_to_edge"), dw, dh, "rgba8"), g["mid"])


def test_overscan_viewport_matches_goldens(rc_lib):
    """FrameCapturePipeline.cpp:205-216: the product's and the oracle's viewport against the one the golden
    was rendered with, plus the clamp at 45 % and the no-overscan identity."""
    from retrocapture_amd import engine
    for case in CASES:
        g = np.load(os.path.join(GOLD, case + ".npz"))
        if "overscan" not in g.files:
            continue
        dh, dw = g["out"].shape[:2]
        px, py = (float(v) for v in g["overscan"])
        want = tuple(int(v) for v in g["vp"])
        assert engine.overscan_viewport(dw, dh, px, py) == want
        assert oracle_lib.overscan_viewport(dw, dh, px, py) == want
    assert engine.overscan_viewport(640, 480, 0.0, 0.0) == (0, 0, 640, 480)
    assert engine.overscan_viewport(640, 480, 80.0, 45.0) == engine.overscan_viewport(640, 480, 45.0, 45.0)
    for args in ((1920, 1080, 7.5, 2.5), (320, 240, 0.5, 12.3), (1280, 720, 33.3, 44.9)):
        assert engine.overscan_viewport(*args) == oracle_lib.overscan_viewport(*args)


def dev_present(src4, is_rgb, linear, dw, dh, kind="rgba8", n=1, **kw):
    import torch
    from retrocapture_amd import engine
    sh, sw = src4.shape[-3:-1]
    d_src = torch.from_numpy(np.ascontiguousarray(src4)).cuda()
    bpp = 3 if kind == "rgb24" else 4
    d_dst = torch.full((n, dh, dw, bpp), 0xCD, dtype=torch.uint8, device="cuda")
    engine.present(d_src, sw, sh, d_dst, dw, dh, n_frames=n, src_rgb=is_rgb, src_linear=linear, dst_kind=kind,
                   stream=torch.cuda.current_stream().cuda_stream, **kw)
    torch.cuda.synchronize()
    return d_dst.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in CASES if "f32" not in c])
def test_present_matches_llvmpipe_golden(case, rc_lib):
    g = np.load(os.path.join(GOLD, case + ".npz"))
    src, is_rgb = rgba_of(g["src"])
    dh, dw = g["out"].shape[:2]
    bake = tuple(float(v) for v in g["bake"]) if "bake" in g.files else None
    out = dev_present(src, is_rgb, str(g["filt"]) == "linear", dw, dh, str(g["dst"]), viewport=tuple(int(v) for v in g["vp"]),
                      flip_y=bool(int(g["flip"])), brightness=float(g["b"]), contrast=float(g["c"]), bake=bake)[0]
    assert np.array_equal(out, g["out"]), case


@pytest.mark.gpu
def test_present_fused_rgb24_flip_and_batch(rc_lib):
    """The fused form (resize + bake + alpha strip + row flip, several frames per launch) against the oracle's
    three separate steps, at a row length that takes the 4-pixel path and one that does not."""
    rng = np.random.default_rng(5)
    for (sw, sh, dw, dh) in ((96, 64, 200, 150), (96, 64, 151, 77)):
        n = 3
        src = rng.integers(0, 256, (n, sh, sw, 4), dtype=np.uint8)
        out = dev_present(src, False, True, dw, dh, "rgb24", n=n, bake=(1.15, 0.9), out_flip_rows=True)
        for z in range(n):
            a = oracle_lib.present(Tex(src[z], "rgba8", True, "clamp_to_edge"), dw, dh, "rgba8")
            b = oracle_lib.present(Tex(a, "rgba8", True, "clamp_to_edge"), dw, dh, "rgba8", None, False, 1.15, 0.9)
            assert np.array_equal(out[z], b[::-1, :, :3]), (sw, sh, dw, dh, z)


@pytest.mark.gpu
def test_present_full_size_properties(rc_lib):
    """1080p: identity (same size, NEAREST and LINEAR, brightness = contrast = 1) returns the source; a 2x
    NEAREST downscale of a frame made of 2x2 blocks returns the blocks; brightness 0 with contrast 1 is black;
    contrast 0 is mid grey 128 (0.5 * 255 = 127.5 rounds to even: 128)."""
    rng = np.random.default_rng(6)
    src = rng.integers(0, 256, (1080, 1920, 4), dtype=np.uint8)
    for linear in (False, True):
        assert np.array_equal(dev_present(src, False, linear, 1920, 1080)[0], src)
    small = rng.integers(0, 256, (540, 960, 4), dtype=np.uint8)
    blocks = np.repeat(np.repeat(small, 2, 0), 2, 1)
    blocks[..., 3] = 255
    small[..., 3] = 255
    assert np.array_equal(dev_present(blocks, True, False, 960, 540, "rgbx8")[0], small)
    black = dev_present(src, False, True, 1920, 1080, brightness=0.0)[0]
    assert (black[..., :3] == 0).all() and np.array_equal(black[..., 3], src[..., 3])
    grey = dev_present(src, False, True, 1920, 1080, contrast=0.0)[0]
    assert (grey[..., :3] == 128).all()


@pytest.mark.gpu
def test_present_rejects_bad_arguments(rc_lib):
    import torch
    from retrocapture_amd import engine
    a = torch.zeros(16 * 16 * 4, dtype=torch.uint8, device="cuda")
    with pytest.raises(engine.RcError):
        engine.present(a, 0, 16, a, 16, 16)
    with pytest.raises(engine.RcError):
        engine.present(a, 16, 16, a, 16, 16, viewport=(0, 0, -4, 16))


@pytest.mark.gpu
def test_pipeline_prepass_resize_adjust(tmp_path, rc_lib):
    """FramePipeline with the reference's optional stages switched on: logical capture size + overscan before
    the chain, output resolution and image adjustments after it - against the same steps through the oracle."""
    import chain_specs
    from oracle_chain import run_chain
    from retrocapture_amd import engine, ShaderEngine
    tree = chain_specs.write_tree(str(tmp_path))
    e = ShaderEngine()
    assert e.init(0)
    e.setAllowMissingSources(True)
    assert e.loadPresetStatus(tree["crt-pi"]) == 0
    e.setViewport(300, 200)
    rng = np.random.default_rng(9)
    frames = [rng.integers(0, 256, (120, 160, 3), dtype=np.uint8) for _ in range(3)]
    p = engine.FramePipeline(e, 2)
    p.setSourcePrepass(128, 96, 4.0, 2.0)
    p.setOutputResolution(231, 150)
    p.setImageAdjust(1.1, 0.95)
    p.setFlipY(True)
    passes = engine.preset_dump(tree["crt-pi"])["passes"]
    for k, f in enumerate(frames):
        assert p.submit(f, "rgb24", 160, 120)
        got = p.receive(True).copy()
        src4, _ = rgba_of(f)
        vp = oracle_lib.overscan_viewport(128, 96, 4.0, 2.0)
        pre = oracle_lib.present(Tex(src4, "rgbx8", False, "clamp_to_edge"), 128, 96, "rgbx8", vp)
        shaded = run_chain(passes, pre[..., :3], 300, 200, frame_count=k + 1)[-1]
        rs = oracle_lib.present(Tex(shaded, "rgba8", True, "clamp_to_edge"), 231, 150, "rgba8")
        fin = oracle_lib.present(Tex(rs, "rgba8", True, "clamp_to_edge"), 231, 150, "rgba8", None, False, 1.1, 0.95)
        assert got.shape == (150, 231, 3)
        assert np.array_equal(got, fin[::-1, :, :3]), k
    p.close()
    e.shutdown()
