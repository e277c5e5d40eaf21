"""The headline configuration as users run it: crt/crt-royale.glslp, 1920x1080 source, 1920x1080 viewport
(BASELINE config 4), in both behaviours of pass 6's unwritten varying (llvmpipe's: mask discarded; and the
GPU drivers': mask rendered).  Every pass of the engine is compared, over the whole frame, with the oracle
fed the engine's own output of the passes before it; the specialised kernel forms (pass 0's byte map, pass 1's
expansion-table form with its exact fallback, ...) are compared with the general per-pixel forms; frames are
uniform noise (every texel pair differs: the worst case for the table form), a smooth natural-like frame and
the reference's test-pattern bars."""
import os

import numpy as np
import pytest

from oracle_chain import run_chain
import oracle_lib

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
W, H = 1920, 1080


def royale_luts():
    lut = np.load(os.path.join(GOLD, "lut_mask_slot_small_64.npy"))
    return {"mask_slot_texture_small": (lut, True, "repeat")}


def bars(w, h, f):
    """The reference's synthetic source (VideoCaptureTestPattern.cpp:65-101): eight bars and a moving marker."""
    cols = np.array([[255, 255, 255], [255, 255, 0], [0, 255, 255], [0, 255, 0], [255, 0, 255], [255, 0, 0], [0, 0, 255],
                     [16, 16, 16]], np.uint8)
    img = cols[np.minimum(np.arange(w) // (w // 8), 7)][None].repeat(h, 0).copy()
    x0 = f % w
    img[: h // 8, x0:x0 + 8] = 0
    return img


def smooth(w, h, seed):
    rng = np.random.default_rng(seed)
    low = rng.integers(0, 256, (h // 24 + 2, w // 24 + 2, 3)).astype(np.float32)
    yy, xx = np.arange(h)[:, None] / 24.0, np.arange(w)[None, :] / 24.0
    y0, x0 = yy.astype(int), xx.astype(int)
    fy, fx = (yy - y0)[..., None], (xx - x0)[..., None]
    img = (low[y0, x0] * (1 - fy) * (1 - fx) + low[y0, x0 + 1] * (1 - fy) * fx + low[y0 + 1, x0] * fy * (1 - fx)
           + low[y0 + 1, x0 + 1] * fy * fx)
    return np.clip(img + rng.normal(0, 1.5, img.shape), 0, 255).astype(np.uint8)


def frames3():
    noise = np.random.default_rng(4).integers(0, 256, (H, W, 3), dtype=np.uint8)
    return np.stack([noise, smooth(W, H, 9), bars(W, H, 1234)])


@pytest.mark.parametrize("mask_rendered", [False, True])
def test_crt_royale_1080p_every_pass_against_the_oracle(mask_rendered, preset_tree, rc_lib):
    from gpu_util import make_engine, run_engine
    from retrocapture_amd import engine as eng
    oracle_lib.set_threads(min(16, os.cpu_count() or 1))
    passes = eng.preset_dump(preset_tree["crt-royale"])["passes"]
    frames = frames3()
    e = make_engine(preset_tree["crt-royale"], W, H)
    e.setUndefinedVaryingZero(mask_rendered)
    final = run_engine(e, frames)
    assert final.shape == (3, H, W, 4)
    assert e.passInfo(1)["kernel"] == "royale-scanlines-v"
    mine = [[e.readPass(i, k) for i in range(12)] for k in range(3)]
    # the general per-pixel forms give the same bytes on every pass
    e.setGeneralKernelsOnly(True)
    final_g = run_engine(e, frames)
    for k in range(3):
        for i in range(12):
            assert np.array_equal(e.readPass(i, k), mine[k][i]), "frame %d pass %d: specialised vs general form" % (k, i)
    assert np.array_equal(final, final_g)
    e.shutdown()
    # every pass against the oracle, fed the engine's own previous passes (FrameCount of frame k = k + 1)
    for k in range(3):
        want = run_chain(passes, frames[k], W, H, frame_count=k + 1, luts=royale_luts(), flags=1 if mask_rendered else 0,
                         given=mine[k])
        for i in range(12):
            assert want[i].shape == mine[k][i].shape
            bad = int((want[i] != mine[k][i]).sum())
            assert bad == 0, "frame %d pass %d: %d bytes differ from the oracle" % (k, i, bad)
        assert np.array_equal(final[k], want[11])
    if mask_rendered:
        assert final[0][..., :3].std() > 10      # the mask path carries signal


@pytest.mark.parametrize("params", [{"geom_mode_runtime": 2.0, "geom_tilt_angle_x": 0.1, "geom_tilt_angle_y": -0.05},
                                    {"geom_overscan_x": 1.05, "geom_overscan_y": 0.95}])
def test_crt_royale_1080p_curved_last_pass_against_the_oracle(params, preset_tree, rc_lib):
    """The headline size with the last pass in its tex2Daa / ray-cast form (curved geometry, overscan): every pass against the
    oracle fed the engine's own previous passes; the parameters only reach pass 11."""
    from gpu_util import make_engine, run_engine
    from retrocapture_amd import engine as eng
    oracle_lib.set_threads(min(16, os.cpu_count() or 1))
    passes = eng.preset_dump(preset_tree["crt-royale"])["passes"]
    frame = smooth(W, H, 21)
    e = make_engine(preset_tree["crt-royale"], W, H)
    e.setUndefinedVaryingZero(True)
    flat = run_engine(e, frame)[0]
    for k, v in params.items():
        assert e.setShaderParameter(k, v)
    final = run_engine(e, frame)[0]
    mine = [e.readPass(i, 0) for i in range(12)]
    e.shutdown()
    want = run_chain(passes, frame, W, H, luts=royale_luts(), flags=1, given=mine, custom=params)
    for i in range(12):
        bad = int((want[i] != mine[i]).sum())
        assert bad == 0, "pass %d: %d bytes differ from the oracle" % (i, bad)
    assert np.array_equal(final, want[11]) and not np.array_equal(final, flat) and final[..., :3].std() > 10


@pytest.mark.parametrize("params", [{"geom_mode_runtime": 1.0}, {"geom_mode_runtime": 3.0, "geom_radius": 1.4, "lcd_gamma": 1.9},
                                    {"geom_overscan_x": 0.93, "geom_overscan_y": 1.04, "aa_cubic_c": 0.8}])
def test_crt_royale_1080p_general_last_pass_gamma_table_equals_exact(params, preset_tree, rc_lib):
    """The general (curved / overscanned) last pass takes its three output-gamma pows from the certified table where the colour lies
    in [0, 1] and the byte is certain (pass_royale_last_general.hip): every byte must equal the all-exact form's
    (rc_engine_set_general_kernels_only) on noise, smooth and bar frames - the tex2Daa weights overshoot on the bars' edges, which
    sends colours outside [0, 1] to the exact pow."""
    from gpu_util import make_engine, run_engine
    fr = frames3()
    e = make_engine(preset_tree["crt-royale"], W, H)
    e.setUndefinedVaryingZero(True)
    for k, v in params.items():
        assert e.setShaderParameter(k, v)
    fast = run_engine(e, fr).copy()
    e.setGeneralKernelsOnly(True)
    exact = run_engine(e, fr)
    e.shutdown()
    assert np.array_equal(fast, exact), "%d differing bytes" % int((fast != exact).sum())
    assert fast[..., :3].std() > 10


def test_scanline_table_form_fallback_share(preset_tree, rc_lib):
    """Geometry that is not the regular 1:1 one (here 1080 -> 1000 lines) must take the general form: same bytes
    as the oracle on a band."""
    from gpu_util import make_engine, run_engine
    from retrocapture_amd import engine as eng
    from oracle_lib import Tex, run_pass_rows
    passes = eng.preset_dump(preset_tree["crt-royale"])["passes"]
    frame = np.random.default_rng(5).integers(0, 256, (H, W, 3), dtype=np.uint8)
    e = make_engine(preset_tree["crt-royale"], W, 1000)
    run_engine(e, frame)
    p0, p1 = e.readPass(0, 0), e.readPass(1, 0)
    e.shutdown()
    assert p1.shape == (1000, W, 4)
    from oracle_chain import pass_sizes
    sizes = pass_sizes(passes, W, H, W, 1000)
    rows = run_pass_rows("royale_scan_v", Tex(p0, "srgb8", True, "clamp_to_edge"), W, 1000, 470, 534, out_fmt="srgb8",
                         src_w=W, src_h=H, chain=sizes, pass_index=1, vp=(W, 1000))
    assert np.array_equal(rows, p1[470:534])


@pytest.mark.parametrize("mask_rendered", [False, True])
def test_crt_royale_1080p_default_launch_shape_equals_small_launches(mask_rendered, preset_tree, rc_lib):
    """The engine's default for 1080p chains is 128 frames per kernel launch (shader_engine.cpp: applyShaderBatch); a batch of
    136 frames - one full launch and a short one - must come out byte for byte as it does 8 frames per launch: frame indices up
    to 127 inside a launch reach every kernel's frame / strip / run arithmetic (the per-wave runs of the bloom pass, the scanline
    pass's fix list, the strip counts)."""
    import torch
    from gpu_util import make_engine
    n = 136
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    frames = torch.randint(0, 256, (n, H, W, 4), dtype=torch.uint8, device="cuda", generator=g)
    frames[..., 3] = 255
    frames[3, :, :, :3] = torch.from_numpy(bars(W, H, 3)).cuda()       # not only noise
    frames[131, :, :, :3] = torch.from_numpy(smooth(W, H, 9)).cuda()
    big, small = make_engine(preset_tree["crt-royale"], W, H), make_engine(preset_tree["crt-royale"], W, H, chunk=8)
    for e in (big, small):
        e.setUndefinedVaryingZero(mask_rendered)
        e.applyShaderBatch(frames, n, W, H)
        e.sync()
    last = big.passCount() - 1
    for k in (0, 3, 7, 8, 63, 64, 126, 127, 128, 131, 135):
        a, b = big.readPass(last, k), small.readPass(last, k)
        assert np.array_equal(a, b), "frame %d: %d differing bytes" % (k, int((a != b).sum()))
    assert a[..., :3].std() > 5
    big.shutdown()
    small.shutdown()


@pytest.mark.parametrize("mask_rendered", [False, True])
def test_crt_royale_1080p_two_lanes_equal_one_lane(mask_rendered, preset_tree, rc_lib):
    """rc_engine_set_lanes(2): the second half of a batch is rendered by the helper instance on its own HIP stream into the same
    batch output.  Every frame must equal the one-lane engine's - the frame counter (crt-royale's field parity) continues across
    the two halves, parameters set on the engine reach the helper, and a second batch (odd size: halves of 4 and 3) continues
    the count."""
    import torch
    from gpu_util import make_engine
    n = 10
    g = torch.Generator(device="cuda")
    g.manual_seed(78)
    frames = torch.randint(0, 256, (n, H, W, 4), dtype=torch.uint8, device="cuda", generator=g)
    frames[..., 3] = 255
    frames[6, :, :, :3] = torch.from_numpy(bars(W, H, 6)).cuda()
    one, two = make_engine(preset_tree["crt-royale"], W, H), make_engine(preset_tree["crt-royale"], W, H)
    two.setLanes(2)
    last = one.passCount() - 1
    for e in (one, two):
        e.setUndefinedVaryingZero(mask_rendered)
        assert e.setShaderParameter("crt_gamma", 2.4)
    for count in (n, 7, 1):   # 1: a single frame stays on one lane
        outs = []
        for e in (one, two):
            e.applyShaderBatch(frames, count, W, H)
            e.sync()
            outs.append([e.readPass(last, k) for k in range(count)])
        for k in range(count):
            assert np.array_equal(outs[0][k], outs[1][k]), "batch of %d, frame %d: %d differing bytes" % (
                count, k, int((outs[0][k] != outs[1][k]).sum()))
    assert outs[0][0][..., :3].std() > 5
    one.shutdown()
    two.shutdown()


def test_crt_royale_1080p_folded_first_pass_equals_rendered(preset_tree, rc_lib):
    """Pass 0 at 1:1 is a byte map of the source (gamma 2.5 into an sRGB8 target): by default the engine does not render it -
    passes 1 and 2 read the source frame through the composed decode table (rc_engine_set_fold_passes).  Every other pass must
    hold the bytes it holds with pass 0 rendered, whatever the source's alpha bytes are (the reference's source texture is
    GL_RGB, its pass 0 writes alpha 1); the folded pass's own bytes, rendered on demand by rc_engine_read_pass, are pass 0's;
    and an interlaced-height source (pass 0 then blends fields: not a byte map) is not folded."""
    import torch
    from gpu_util import make_engine
    n = 5
    g = torch.Generator(device="cuda")
    g.manual_seed(79)
    frames = torch.randint(0, 256, (n, H, W, 4), dtype=torch.uint8, device="cuda", generator=g)   # alpha: noise too
    frames[1, :, :, :3] = torch.from_numpy(bars(W, H, 6)).cuda()
    frames[2, :, :, :3] = torch.from_numpy(smooth(W, H, 5)).cuda()
    folded, plain = make_engine(preset_tree["crt-royale"], W, H), make_engine(preset_tree["crt-royale"], W, H)
    plain.setFoldPasses(False)
    for e in (folded, plain):
        e.setProfiling(True)
        e.applyShaderBatch(frames, n, W, H)
        e.sync()
    assert folded.passProfile(0)["folded"] and folded.passProfile(0)["launches"] == 0
    assert not plain.passProfile(0)["folded"] and plain.passProfile(0)["launches"] == 1
    assert not any(folded.passProfile(i)["folded"] for i in range(1, 12))
    # the algorithmic byte counts do not change with folding
    for i in range(12):
        a, b = folded.passProfile(i), plain.passProfile(i)
        assert (a["read_bytes_per_frame"], a["write_bytes_per_frame"]) == (b["read_bytes_per_frame"], b["write_bytes_per_frame"]), i
    for k in range(n):
        for i in range(11, -1, -1):   # pass 0 last: reading it renders it
            a, b = folded.readPass(i, k), plain.readPass(i, k)
            assert np.array_equal(a, b), "frame %d pass %d: %d differing bytes" % (k, i, int((a != b).sum()))
    assert plain.readPass(0, 0)[..., 3].min() == 255 and plain.readPass(2, 0)[..., 3].min() == 255
    folded.shutdown()
    plain.shutdown()
    # 480 source rows: is_interlaced, pass 0 bobs fields - rendered, not folded
    e = make_engine(preset_tree["crt-royale"], 640, 480)
    e.setProfiling(True)
    small = torch.randint(0, 256, (2, 480, 640, 4), dtype=torch.uint8, device="cuda", generator=g)
    e.applyShaderBatch(small, 2, 640, 480)
    e.sync()
    assert not e.passProfile(0)["folded"] and e.passProfile(0)["launches"] == 1
    e.shutdown()


def test_async_table_builds_do_not_stall_the_frame_path(preset_tree, rc_lib):
    """rc_engine_set_async_table_builds (the engine's default; the test helper switches it off): the scanline pass's tables - error
    bounds proven by exhaustion, 140 ms at 1080p - are built on a worker thread while the exact per-pixel form serves the frames,
    so no apply call waits for them (the reference's callers resize the window and move parameters from UI and HTTP threads);
    the bytes are the table form's, before and after the switch.  A viewport no other test uses, so that the build is this
    test's own."""
    import time
    import torch
    from gpu_util import make_engine
    vw, vh = 1888, 1062
    g = torch.Generator(device="cuda")
    g.manual_seed(80)
    frame = torch.randint(0, 256, (1, vh, vw, 4), dtype=torch.uint8, device="cuda", generator=g)
    e = make_engine(preset_tree["crt-royale"], vw, vh)
    e.setAsyncTableBuilds(True)
    last = e.passCount() - 1
    longest, first = 0.0, None
    t_end = time.perf_counter() + 3.0
    n = 0
    while time.perf_counter() < t_end and n < 400:
        t0 = time.perf_counter()
        e.applyShaderBatch(frame, 1, vw, vh)
        e.sync()
        dt = time.perf_counter() - t0
        if n > 0: longest = max(longest, dt)      # (the first call allocates every target)
        if first is None: first = [e.readPass(i, 0) for i in (1, last)]
        n += 1
    e.setProfiling(True)
    e.applyShaderBatch(frame, 1, vw, vh)
    e.sync()
    late = [e.readPass(i, 0) for i in (1, last)]
    assert np.array_equal(first[0], late[0]) and np.array_equal(first[1], late[1])
    print("longest apply call while the tables were being built: %.2f ms over %d calls" % (longest * 1e3, n))
    assert longest < 0.010, "an apply call took %.1f ms while the tables were being built" % (longest * 1e3)
    e.shutdown()
    # and the same bytes from an engine that waited for its tables
    s = make_engine(preset_tree["crt-royale"], vw, vh)
    s.applyShaderBatch(frame, 1, vw, vh)
    s.sync()
    assert np.array_equal(s.readPass(1, 0), late[0]) and np.array_equal(s.readPass(last, 0), late[1])
    s.shutdown()


def test_two_lanes_first_on_a_cold_geometry(preset_tree, rc_lib):
    """The second lane's stream must never see a per-geometry table before its build has finished (every build is synchronised
    before the tables are cached, royale_strip.h geo_tables): on a viewport no other test uses, the two-lane engine runs FIRST -
    every table is built during its call, with both streams live - and is compared with a one-lane engine afterwards; then a
    viewport and a parameter change between batches (the helper instance reloads its configuration)."""
    import torch
    from gpu_util import make_engine
    n = 6
    g = torch.Generator(device="cuda")
    g.manual_seed(81)
    for (vw, vh), param in (((1872, 1053), None), ((1856, 1044), ("crt_gamma", 2.3))):
        frames = torch.randint(0, 256, (n, vh, vw, 4), dtype=torch.uint8, device="cuda", generator=g)
        frames[..., 3] = 255
        if param is None:
            two = make_engine(preset_tree["crt-royale"], vw, vh)
            two.setLanes(2)
            two.setUndefinedVaryingZero(True)
        else:
            two.setViewport(vw, vh)
            assert two.setShaderParameter(*param)
        two.applyShaderBatch(frames, n, vw, vh)
        two.sync()
        last = two.passCount() - 1
        got = [two.readPass(last, k) for k in range(n)]
        one = make_engine(preset_tree["crt-royale"], vw, vh)
        one.setUndefinedVaryingZero(True)
        if param is not None:
            assert one.setShaderParameter(*param)
            one.applyShaderBatch(frames, n, vw, vh)   # (the frame counter: the two-lane engine has rendered one batch before)
        one.applyShaderBatch(frames, n, vw, vh)
        one.sync()
        for k in range(n):
            want = one.readPass(last, k)
            assert np.array_equal(got[k], want), "%dx%d frame %d: %d differing bytes" % (vw, vh, k, int((got[k] != want).sum()))
        one.shutdown()
    two.shutdown()


@pytest.mark.parametrize("size", [(1872, 1053), (1840, 1035)])
def test_crt_royale_widths_that_leave_a_partial_wave(size, preset_tree, rc_lib):
    """Widths that are not a multiple of 64 (16 / 48 columns in the last wave of a row): the strip and table forms against the
    general per-pixel forms, every pass.  (Round 4 found the scanline pass's per-wave list of uncertain pixels losing entries
    after a wave had rendered a strip with lanes beyond the right edge switched off.)"""
    import torch
    from gpu_util import make_engine
    vw, vh = size
    g = torch.Generator(device="cuda")
    g.manual_seed(82)
    frames = torch.randint(0, 256, (3, vh, vw, 4), dtype=torch.uint8, device="cuda", generator=g)
    e = make_engine(preset_tree["crt-royale"], vw, vh)
    e.setUndefinedVaryingZero(True)
    e.applyShaderBatch(frames, 3, vw, vh)
    e.sync()
    mine = [[e.readPass(i, k) for i in range(12)] for k in range(3)]
    e.setGeneralKernelsOnly(True)
    e.applyShaderBatch(frames, 3, vw, vh)
    e.sync()
    for k in range(3):
        for i in range(12):
            want = e.readPass(i, k)
            assert np.array_equal(mine[k][i], want), "frame %d pass %d: %d differing bytes" % (k, i, int((mine[k][i] != want).sum()))
    e.shutdown()


def test_crt_royale_1080p_letterboxed_frames_forms_agree(preset_tree, rc_lib):
    """Black bars with the mask rendered: passes 7 - 10 see black regions NEXT to coloured ones (the bloom pass's black-window
    shortcut must switch on and off inside a wave's run; in llvmpipe's default mode the whole frame is black for these passes
    and no transition occurs).  Letterbox, pillarbox, a black frame, and bars that start / end off the 4-row step grid."""
    import torch
    from gpu_util import make_engine
    g = torch.Generator(device="cuda")
    g.manual_seed(83)
    frames = torch.randint(0, 256, (4, H, W, 4), dtype=torch.uint8, device="cuda", generator=g)
    frames[0, :137] = 0
    frames[0, H - 141:] = 0            # letterbox
    frames[1, :, :241] = 0
    frames[1, :, W - 239:] = 0         # pillarbox
    frames[2] = 0                      # a black frame
    frames[3, 301:613, 500:1400] = 0   # a black window inside the picture
    e = make_engine(preset_tree["crt-royale"], W, H)
    e.setUndefinedVaryingZero(True)
    e.applyShaderBatch(frames, 4, W, H)
    e.sync()
    mine = [[e.readPass(i, k) for i in range(12)] for k in range(4)]
    assert mine[0][9][300:700, :, :3].any() and not mine[2][9][..., :3].any()
    e.setGeneralKernelsOnly(True)
    e.applyShaderBatch(frames, 4, W, H)
    e.sync()
    for k in range(4):
        for i in range(12):
            want = e.readPass(i, k)
            assert np.array_equal(mine[k][i], want), "frame %d pass %d: %d differing bytes" % (k, i, int((mine[k][i] != want).sum()))
    e.shutdown()


@pytest.mark.parametrize("size", [(1280, 720), (1366, 768), (2560, 1440), (854, 480), (720, 576), (1920, 1200)])
@pytest.mark.parametrize("mask_rendered", [False, True])
def test_crt_royale_other_frame_sizes_forms_agree(size, mask_rendered, preset_tree, rc_lib):
    """The specialised forms (strip / table / tile / quad kernels, folded pass 0, two lanes) against the general per-pixel forms on
    one lane at frame sizes other than 1080p - heights that are not multiples of the strips' 8 / 16 / 32 rows, widths that are not
    multiples of 64 / 128, a 4:3 and a 16:10 frame: every pass, every byte."""
    import torch
    from gpu_util import make_engine
    vw, vh = size
    g = torch.Generator(device="cuda")
    g.manual_seed(84 + vw)
    n = 5
    frames = torch.randint(0, 256, (n, vh, vw, 4), dtype=torch.uint8, device="cuda", generator=g)
    frames[1, : vh // 7] = 0   # (a black bar: the bloom pass's black-window shortcut next to coloured rows)
    e = make_engine(preset_tree["crt-royale"], vw, vh)
    e.setUndefinedVaryingZero(mask_rendered)
    e.applyShaderBatch(frames, n, vw, vh)
    e.sync()
    mine = [[e.readPass(i, k) for i in range(12)] for k in range(n)]
    e.shutdown()
    # a fresh engine for the general forms: FrameCount starts over (480- and 576-line sources are interlaced to crt-royale, and
    # its field handling reads the frame counter's parity - a second call on the same engine renders the other fields)
    e = make_engine(preset_tree["crt-royale"], vw, vh)
    e.setUndefinedVaryingZero(mask_rendered)
    e.setGeneralKernelsOnly(True)
    e.setLanes(1)
    e.applyShaderBatch(frames, n, vw, vh)
    e.sync()
    for k in range(n):
        for i in range(12):
            want = e.readPass(i, k)
            assert np.array_equal(mine[k][i], want), "%dx%d frame %d pass %d: %d differing bytes" % (vw, vh, k, i, int((mine[k][i] != want).sum()))
    e.shutdown()
