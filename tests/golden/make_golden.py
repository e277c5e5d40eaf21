#!/usr/bin/env python3
"""Generates tests/golden/*.npz: golden vectors for the shader chain.

Runs ONLY in the build container (needs /root/reference and oracle/_ref, built by
`make -C oracle ref`): each case executes the reference's own GLSL shader files on Mesa
llvmpipe through oracle/_ref/glchain (pass plumbing restated from ShaderEngine.cpp, preset
parsed by the reference's own ShaderPreset.cpp) and stores the input frame plus every
pass's stored texels.  What is committed is data (inputs and outputs), never shader text.

    python tests/golden/make_golden.py [case ...]
"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
GLSL = REF + "/shaders/shaders_glsl"
GLCHAIN = os.path.join(ROOT, "oracle", "_ref", "glchain")


def bars(w, h, f=0):
    """The reference's synthetic source (VideoCaptureTestPattern.cpp:65-101): 8 colour bars
    plus a black 8-px marker that moves with the frame number."""
    cols = np.array([[255, 255, 255], [255, 255, 0], [0, 255, 255], [0, 255, 0], [255, 0, 255], [255, 0, 0],
                     [0, 0, 255], [16, 16, 16]], np.uint8)
    bw = max(1, w // 8)
    idx = np.minimum(np.arange(w) // bw, 7)
    img = np.broadcast_to(cols[idx][None, :, :], (h, w, 3)).copy()
    x0 = f % w
    img[: h // 8, x0:min(w, x0 + 8)] = 0
    return img


def noise(w, h, seed):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


def mixed(w, h, seed):
    """Left half bars, right half noise, plus a smooth gradient band: exercises flat
    regions, hard edges and every byte value."""
    img = bars(w, h, 3)
    img[:, w // 2:] = noise(w - w // 2, h, seed)
    g = (np.arange(w) * 255 // max(1, w - 1)).astype(np.uint8)
    img[h // 3: h // 3 + max(2, h // 8)] = np.stack([g, g[::-1], g], -1)[None]
    return img


def write_preset(d, text):
    p = os.path.join(d, "case.glslp")
    open(p, "w").write(text)
    return p


def run_case(name, preset, rgb, vw, vh, frames=1, luts=(), params=(), f32=False):
    """rgb: (h, w, 3) applied `frames` times, or (frames, h, w, 3): a different source every frame."""
    if rgb.ndim == 4:
        frames = rgb.shape[0]
    h, w, _ = rgb.shape[-3:]
    with tempfile.TemporaryDirectory() as d:
        rgb.tofile(os.path.join(d, "in.rgb"))
        cmd = [GLCHAIN, "--preset", preset, "--input", os.path.join(d, "in.rgb"), "--w", str(w), "--h", str(h),
               "--vw", str(vw), "--vh", str(vh), "--frames", str(frames), "--out", d]
        for n, (path, lw, lh) in luts:
            cmd += ["--lut", "%s=%s:%d:%d" % (n, path, lw, lh)]
        for k, v in params:
            cmd += ["--param", "%s=%g" % (k, v)]
        env = dict(os.environ, RETROCAPTURE_LOG_LEVEL="error")
        if f32:
            # every target RGBA32F: the stored values are the shaders' floats, so the oracle's arithmetic
            # can be compared bit for bit (an 8-bit target hides last-bit differences)
            env["GLCHAIN_F32"] = "1"
        r = subprocess.run(cmd, cwd=REF, env=env, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr)
        out = {"input_rgb": rgb, "viewport": np.array([vw, vh]), "frames": np.array(frames)}
        if params:
            out["param_names"] = np.array([k for k, _ in params])
            out["param_values"] = np.array([v for _, v in params], np.float32)
        meta = open(os.path.join(d, "meta.txt")).read().splitlines()
        k = 0
        for line in meta:
            if not line.startswith("pass "):
                continue
            _, i, pw, ph, fmt = line.split()[:5]
            pw, ph = int(pw), int(ph)
            dt = np.float32 if fmt == "f32" else np.uint8
            out["pass%d" % k] = np.fromfile(os.path.join(d, "pass%s.bin" % i), dtype=dt).reshape(ph, pw, 4)
            out["pass%d_fmt" % k] = np.array(fmt)
            k += 1
        nh = 0
        for line in meta:
            if line.startswith("history "):
                _, i, pw, ph = line.split()[:4]
                out["history%d" % nh] = np.fromfile(os.path.join(d, "history%s.bin" % i), dtype=np.uint8).reshape(int(ph), int(pw), 4)
                nh += 1
        out["n_history"] = np.array(nh)
        out["n_passes"] = np.array(k)
        out["sha256_last"] = np.array(hashlib.sha256(out["pass%d" % (k - 1)].tobytes()).hexdigest())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {kk: getattr(v, "shape", None) for kk, v in out.items() if kk.startswith("pass") and not kk.endswith("fmt")})


def case_scanline():
    # BASELINE config 1: misc/scanline.glslp does not exist in the reference; the shader is
    # run from a synthesized one-pass preset (SURVEY.md, facts established by probing).
    with tempfile.TemporaryDirectory() as d:
        p = write_preset(d, 'shaders = 1\nshader0 = %s/scanlines/shaders/scanline.glsl\n' % GLSL)
        run_case("scanline_320x240", p, mixed(320, 240, 1), 320, 240)
        run_case("scanline_64x48_to_160x100", p, mixed(64, 48, 2), 160, 100)


def case_crt_pi():
    run_case("crt_pi_96x64_to_192x128", GLSL + "/crt/crt-pi.glslp", mixed(96, 64, 3), 192, 128)
    run_case("crt_pi_80x60_to_250x190", GLSL + "/crt/crt-pi.glslp", noise(80, 60, 4), 250, 190)


def royale_luts(d):
    """LUT PNGs decoded to RGBA8 exactly as the reference's loader does for 8-bit RGB files
    (opaque alpha added, ShaderEngine.cpp:2535-2706); glchain uploads the raw texels."""
    from PIL import Image
    names = {
        "mask_grille_texture_small": "TileableLinearApertureGrille15Wide8And5d5SpacingResizeTo64.png",
        "mask_grille_texture_large": "TileableLinearApertureGrille15Wide8And5d5Spacing.png",
        "mask_slot_texture_small": "TileableLinearSlotMaskTall15Wide9And4d5Horizontal9d14VerticalSpacingResizeTo64.png",
        "mask_slot_texture_large": "TileableLinearSlotMaskTall15Wide9And4d5Horizontal9d14VerticalSpacing.png",
        "mask_shadow_texture_small": "TileableLinearShadowMaskEDPResizeTo64.png",
        "mask_shadow_texture_large": "TileableLinearShadowMaskEDP.png",
    }
    out = []
    for n, f in names.items():
        im = Image.open(GLSL + "/crt/shaders/crt-royale/" + f).convert("RGBA")
        raw = os.path.join(d, n + ".rgba")
        np.asarray(im).tofile(raw)
        out.append((n, (raw, im.size[0], im.size[1])))
    return out


def case_crt_royale():
    with tempfile.TemporaryDirectory() as d:
        luts = royale_luts(d)
        run_case("crt_royale_160x120_to_320x240", GLSL + "/crt/crt-royale.glslp", mixed(160, 120, 5), 320, 240,
                 frames=2, luts=luts)
        run_case("crt_royale_128x96_to_400x300", GLSL + "/crt/crt-royale.glslp", noise(128, 96, 6), 400, 300, luts=luts)


ROYALE_GEOM = {
    # name suffix -> the last pass's geometry parameters (crt-royale-geometry-aa-last-pass.glsl 50-58)
    "sphere": [("geom_mode_runtime", 1.0)],
    "sphere_alt_tilt": [("geom_mode_runtime", 2.0), ("geom_tilt_angle_x", 0.2), ("geom_tilt_angle_y", -0.15), ("geom_radius", 1.5)],
    "cylinder": [("geom_mode_runtime", 3.0), ("geom_view_dist", 1.25), ("geom_overscan_x", 1.0625), ("aa_cubic_c", 0.25)],
    "flat_overscan": [("geom_overscan_x", 1.125), ("geom_overscan_y", 0.9375), ("border_size", 0.03)],
}


def case_crt_royale_geom():
    """crt-royale with curved geometry / overscan: the last pass's tex2Daa12x + ray-cast branch."""
    with tempfile.TemporaryDirectory() as d:
        luts = royale_luts(d)
        for k, params in ROYALE_GEOM.items():
            run_case("crt_royale_geom_%s_96x72_to_240x180" % k, GLSL + "/crt/crt-royale.glslp", mixed(96, 72, 31), 240, 180, luts=luts, params=params)
            run_case("f32_crt_royale_geom_%s_64x48_to_128x96" % k, GLSL + "/crt/crt-royale.glslp", mixed(64, 48, 5), 128, 96, luts=luts,
                     params=params, f32=True)
        run_case("crt_royale_geom_sphere_128x96_to_401x299", GLSL + "/crt/crt-royale.glslp", noise(128, 96, 33), 401, 299, luts=luts,
                 params=ROYALE_GEOM["sphere"])


def case_royale_fake_bloom_geom():
    """crt-royale-fake-bloom (mask pass active) with curved geometry / overscan: its last pass declares mipmap_input, so each
    of the twelve tex2Daa taps takes its LOD from the pixel quad; overscan 1.6 minifies enough to reach mip levels 1-2."""
    import shutil
    with tempfile.TemporaryDirectory() as d:
        luts = royale_luts(d)
        dst = os.path.join(d, "shaders_glsl")
        os.makedirs(os.path.join(dst, "crt", "shaders"))
        shutil.copytree(GLSL + "/crt/shaders/crt-royale", dst + "/crt/shaders/crt-royale")
        shutil.copytree(GLSL + "/blurs", dst + "/blurs")
        shutil.copy(GLSL + "/crt/crt-royale-fake-bloom.glslp", dst + "/crt/crt-royale-fake-bloom.glslp")
        f = dst + "/crt/shaders/crt-royale/src/crt-royale-mask-resize-horizontal.glsl"
        txt = open(f).read()
        needle = "max(tile_uv_wrap.x, tile_uv_wrap.y) <= mask_resize_num_tiles"
        assert txt.count(needle) == 1
        open(f, "w").write(txt.replace(needle, "0.0 <= mask_resize_num_tiles"))
        P = dst + "/crt/crt-royale-fake-bloom.glslp"
        cases = {"sphere": [("geom_mode_runtime", 1.0), ("geom_overscan_x", 1.6), ("geom_overscan_y", 1.6)],
                 "cylinder_tilt": [("geom_mode_runtime", 3.0), ("geom_tilt_angle_x", 0.25), ("geom_radius", 1.2)],
                 "flat_overscan": [("geom_overscan_x", 2.5), ("geom_overscan_y", 1.75)]}
        for k, params in cases.items():
            run_case("crt_royale_fake_bloom_geom_%s_maskon_96x72_to_240x180" % k, P, mixed(96, 72, 41), 240, 180, luts=luts, params=params)
            run_case("f32_crt_royale_fake_bloom_geom_%s_maskon_64x48_to_128x96" % k, P, mixed(64, 48, 5), 128, 96, luts=luts, params=params, f32=True)


def case_crt_royale_mask_active(f32=False):
    """crt-royale as a GL driver that returns 0 for an unwritten varying would render it.
    Pass 6's fragment shader tests `max(tile_uv_wrap.x, tile_uv_wrap.y) <= mask_resize_num_tiles`
    on a varying its vertex shader never writes (the VS shadows it with a local).  Mesa llvmpipe
    discards every fragment there; drivers that read 0 keep them all.  To produce vectors for
    that second behaviour the pass-6 file is copied to a temp dir with that one test replaced by
    the constant it would see (0.0); nothing of the copy is kept."""
    import shutil
    with tempfile.TemporaryDirectory() as d:
        dst = os.path.join(d, "shaders_glsl")
        os.makedirs(os.path.join(dst, "crt", "shaders"))
        shutil.copytree(GLSL + "/crt/shaders/crt-royale", dst + "/crt/shaders/crt-royale")
        shutil.copytree(GLSL + "/blurs", dst + "/blurs")
        shutil.copy(GLSL + "/crt/crt-royale.glslp", dst + "/crt/crt-royale.glslp")
        f = dst + "/crt/shaders/crt-royale/src/crt-royale-mask-resize-horizontal.glsl"
        txt = open(f).read()
        needle = "max(tile_uv_wrap.x, tile_uv_wrap.y) <= mask_resize_num_tiles"
        assert txt.count(needle) == 1
        open(f, "w").write(txt.replace(needle, "0.0 <= mask_resize_num_tiles"))
        luts = royale_luts(d)
        if f32:
            run_case("f32_crt_royale_maskon_64x48_to_128x96", dst + "/crt/crt-royale.glslp", mixed(64, 48, 5), 128, 96,
                     luts=luts, f32=True)
            return
        run_case("crt_royale_maskon_160x120_to_320x240", dst + "/crt/crt-royale.glslp", mixed(160, 120, 5), 320, 240,
                 luts=luts)
        run_case("crt_royale_maskon_96x128_to_512x384", dst + "/crt/crt-royale.glslp", noise(96, 128, 7), 512, 384,
                 luts=luts)


def case_ntsc():
    # BASELINE config 3: 2 passes, RGBA32F intermediate of absolute width 1024, frame_count_mod 2
    run_case("ntsc_svideo_96x64_to_256x192", GLSL + "/ntsc/ntsc-256px-svideo.glslp", mixed(96, 64, 8), 256, 192,
             frames=2)
    run_case("ntsc_svideo_120x50_to_301x117", GLSL + "/ntsc/ntsc-256px-svideo.glslp", noise(120, 50, 9), 301, 117,
             frames=3)


def case_ntsc_family():
    # the other members of the ntsc family: composite / 2-phase pass 1, 65-tap 2-phase pass 2, and the
    # -linear / plain pass-2 epilogues (through synthesized two-pass presets with the shipped geometry)
    run_case("ntsc_256px_composite_80x48_to_200x144", GLSL + "/ntsc/ntsc-256px.glslp", mixed(80, 48, 40), 200, 144, frames=2)
    run_case("ntsc_320px_composite_72x40_to_320x120", GLSL + "/ntsc/ntsc-320px.glslp", mixed(72, 40, 41), 320, 120, frames=3)
    run_case("ntsc_320px_svideo_64x36_to_161x77", GLSL + "/ntsc/ntsc-320px-svideo.glslp", noise(64, 36, 42), 161, 77)
    with tempfile.TemporaryDirectory() as d:
        for name, p1, p2, width in (("ntsc_3phase_linear", "svideo-3phase", "3phase-linear", 1024),
                                    ("ntsc_3phase_plain", "composite-3phase", "3phase", 1024),
                                    ("ntsc_2phase_linear", "composite-2phase", "2phase-linear", 1280),
                                    ("ntsc_2phase_plain", "svideo-2phase", "2phase", 1280)):
            p = write_preset(d, "shaders = 2\nshader0 = %s/ntsc/shaders/ntsc-pass1-%s.glsl\nshader1 = %s/ntsc/shaders/ntsc-pass2-%s.glsl\n"
                                "filter_linear0 = false\nfilter_linear1 = false\nscale_type_x0 = absolute\nscale_type_y0 = source\n"
                                "scale_x0 = %d\nscale_y0 = 1.0\nfloat_framebuffer0 = true\nscale_type1 = source\nscale_x1 = 0.5\nscale_y1 = 1.0\n"
                             % (GLSL, p1, GLSL, p2, width))
            run_case(name + "_56x30_to_140x66", p, mixed(56, 30, 43), 140, 66, frames=2)


def pixel_art(w, h, seed):
    """Few-colour blocky image with diagonals and single-pixel features: the edge patterns
    xBR's rules react to (noise alone almost never satisfies its equality tests)."""
    rng = np.random.default_rng(seed)
    pal = rng.integers(0, 256, (6, 3), dtype=np.uint8)
    pal[0] = 0
    pal[1] = 255
    yy, xx = np.mgrid[0:h, 0:w]
    idx = ((xx // 5 + yy // 3) % 3).astype(np.int64)
    idx[(xx + yy) % 11 < 3] = 3
    idx[(xx - 2 * yy) % 17 < 2] = 4
    idx[((xx - w // 2) ** 2 + (yy - h // 2) ** 2) < (min(w, h) // 3) ** 2] = 5
    idx[((xx - w // 2) ** 2 + (yy - h // 2) ** 2) < (min(w, h) // 5) ** 2] = 1
    sp = rng.random((h, w)) < 0.03
    idx[sp] = rng.integers(0, 6, sp.sum())
    img = pal[idx]
    img[:, : w // 6] = noise(w // 6, h, seed + 1)   # a noisy strip: near-equal colours
    return img


def case_xbr():
    # BASELINE config 5: one pass, nearest, viewport-sized RGBA8 target
    run_case("xbr_lv3_64x56_to_256x224", GLSL + "/xbr/xbr-lv3.glslp", pixel_art(64, 56, 10), 256, 224)
    run_case("xbr_lv3_48x40_to_331x217", GLSL + "/xbr/xbr-lv3.glslp", pixel_art(48, 40, 11), 331, 217)
    run_case("xbr_lv3_noise_40x36_to_240x216", GLSL + "/xbr/xbr-lv3.glslp", noise(40, 36, 14), 240, 216,
             params=[("XBR_Y_WEIGHT", 30.0)])
    # the other two corner rules and non-default thresholds (custom parameters, ShaderEngine.cpp:3264-3387)
    run_case("xbr_lv3_corner1_40x36_to_200x180", GLSL + "/xbr/xbr-lv3.glslp", pixel_art(40, 36, 12), 200, 180,
             params=[("corner_type", 1.0)])
    run_case("xbr_lv3_corner2_40x36_to_240x216", GLSL + "/xbr/xbr-lv3.glslp", pixel_art(40, 36, 13), 240, 216,
             params=[("corner_type", 2.0), ("XBR_EQ_THRESHOLD", 20.0), ("XBR_EQ_THRESHOLD2", 4.0),
                     ("XBR_LV2_COEFFICIENT", 3.0), ("XBR_Y_WEIGHT", 30.0)])


def moving(w, h, n, seed):
    """n frames: a noise background with a bright bar that moves 3 px per frame."""
    base = noise(w, h, seed)
    out = np.stack([base] * n)
    for f in range(n):
        x0 = (5 + 3 * f) % max(1, w - 6)
        out[f, h // 4: h // 2, x0:x0 + 6] = (250, 240, 10)
        out[f, :, :, 2] = np.roll(out[f, :, :, 2], f, axis=0)   # the whole blue plane scrolls
    return out


def case_mix_frames():
    # frame history (reference ShaderEngine.cpp:1095-1159 binds it, :1735-1865 pushes it): one pass that
    # samples PrevTexture.  3 frames exercise the first-frame rule and the recursion of the ring's
    # content; 9 frames wrap the 7-deep ring.
    run_case("mix_frames_72x40_to_72x40_f3", GLSL + "/motionblur/mix_frames.glslp", moving(72, 40, 3, 20), 72, 40)
    run_case("mix_frames_48x36_to_120x90_f9", GLSL + "/motionblur/mix_frames.glslp", moving(48, 36, 9, 21), 120, 90)


def flicker(w, h, n, seed):
    """n frames whose pixels partly alternate between two colours frame by frame (what mix_frames_smart looks for),
    partly move, partly stay."""
    rng = np.random.default_rng(seed)
    a, b = rng.integers(0, 256, (h, w, 3), dtype=np.uint8), rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    kind = rng.integers(0, 4, (h, w, 1))
    out = np.zeros((n, h, w, 3), np.uint8)
    for f in range(n):
        alt = a if f % 2 == 0 else b
        near = np.clip(alt.astype(np.int16) + rng.integers(-1, 2, (h, w, 3)), 0, 255).astype(np.uint8)
        rnd = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        out[f] = np.where(kind == 0, alt, np.where(kind == 1, near, np.where(kind == 2, a, rnd)))
    return out


def case_motionblur():
    """The other four motionblur/ presets: frame history down to Prev6Texture, first-frames rule included (2 and 9 frames)."""
    M = GLSL + "/motionblur/"
    run_case("motionblur_simple_48x36_to_120x90_f9", M + "motionblur-simple.glslp", moving(48, 36, 9, 31), 120, 90)
    run_case("motionblur_simple_40x30_to_40x30_f3", M + "motionblur-simple.glslp", moving(40, 30, 3, 32), 40, 30)
    run_case("braid_rewind_48x36_to_120x90_f8", M + "braid-rewind.glslp", moving(48, 36, 8, 33), 120, 90)
    run_case("response_time_48x36_to_120x90_f9", M + "response-time.glslp", moving(48, 36, 9, 34), 120, 90)
    run_case("response_time_params_40x30_to_100x75_f4", M + "response-time.glslp", moving(40, 30, 4, 35), 100, 75, params=[("response_time", 0.666)])
    run_case("mix_frames_smart_48x36_to_120x90_f8", M + "mix_frames_smart.glslp", flicker(48, 36, 8, 36), 120, 90)
    run_case("mix_frames_smart_params_40x30_to_40x30_f7", M + "mix_frames_smart.glslp", flicker(40, 30, 7, 37), 40, 30, params=[("DEFLICKER_EMPHASIS", 0.01)])
    run_case("f32_response_time_48x36_to_120x90_f9", M + "response-time.glslp", moving(48, 36, 9, 38), 120, 90, f32=True)
    run_case("f32_mix_frames_smart_48x36_to_120x90_f8", M + "mix_frames_smart.glslp", flicker(48, 36, 8, 39), 120, 90, f32=True)
    run_case("f32_motionblur_simple_48x36_to_120x90_f9", M + "motionblur-simple.glslp", moving(48, 36, 9, 40), 120, 90, f32=True)


def case_feedback():
    # PassFeedback (reference ShaderEngine.cpp:1285-1347, swap :1710-1718): no shader in the reference's
    # tree declares it, so the engine semantics are pinned with a fixture shader of this repository
    # (tests/fixtures/conformance/feedback-persist.glsl) as pass 1 behind the reference's stock.glsl.
    fixture = os.path.join(ROOT, "tests", "fixtures", "conformance", "feedback-persist.glsl")
    with tempfile.TemporaryDirectory() as d:
        p = write_preset(d, 'shaders = 2\nshader0 = %s/stock.glsl\nfilter_linear0 = false\nscale_type0 = source\n'
                            'shader1 = %s\nfilter_linear1 = true\n' % (GLSL, fixture))
        run_case("feedback_persist_64x40_to_64x40_f1", p, moving(64, 40, 1, 30), 64, 40)
        run_case("feedback_persist_64x40_to_64x40_f2", p, moving(64, 40, 2, 30), 64, 40)
        run_case("feedback_persist_64x40_to_150x90_f5", p, moving(64, 40, 5, 31), 150, 90, params=[("PERSIST", 0.6)])


def case_history_size():
    """The history re-draw through a pass 0 that reads its size uniforms (ShaderEngine.cpp:1805-1834: the program keeps what pass
    0's own draw set - stale whenever the final output's size is not pass 0's), pinned with this repository's fixture shader:
    pass 0 scaled 2x, stock behind it to the viewport; ring content after 4 and 9 frames (the recycling rule included)."""
    fixture = os.path.join(ROOT, "tests", "fixtures", "conformance", "history-size.glsl")
    with tempfile.TemporaryDirectory() as d:
        p = write_preset(d, 'shaders = 2\nshader0 = %s\nfilter_linear0 = true\nscale_type0 = source\nscale0 = 2.0\n'
                            'shader1 = %s/stock.glsl\nfilter_linear1 = true\n' % (fixture, GLSL))
        run_case("history_size_48x36_to_120x90_f4", p, moving(48, 36, 4, 61), 120, 90)
        run_case("history_size_params_40x30_to_131x77_f9", p, moving(40, 30, 9, 62), 131, 77, params=[("HS_MIX", 0.55)])
        q = write_preset(d, 'shaders = 1\nshader0 = %s\nfilter_linear0 = false\n' % fixture)
        run_case("history_size_single_48x36_to_100x75_f3", q, moving(48, 36, 3, 63), 100, 75)


def case_sampler_matrix():
    """Wrap modes and filters at chain level: crt-pi reading the source, stock reading crt-pi's (sRGB8 /
    RGBA8) target, both with the wrap mode under test, LINEAR.  Pins clamp_to_border / repeat /
    mirrored_repeat handling at the texture edges on the real GL."""
    with tempfile.TemporaryDirectory() as d:
        for wrap in ("clamp_to_edge", "clamp_to_border", "repeat", "mirrored_repeat"):
            for fb, tag in (("", "rgba8"), ("srgb_framebuffer0 = true\n", "srgb8")):
                p = write_preset(d, "shaders = 2\nshader0 = %s/crt/shaders/crt-pi.glsl\nfilter_linear0 = true\nwrap_mode0 = %s\n"
                                    "scale_type0 = source\nscale0 = 2.0\n%sshader1 = %s/stock.glsl\nfilter_linear1 = true\nwrap_mode1 = %s\n"
                                 % (GLSL, wrap, fb, GLSL, wrap))
                run_case("wrap_%s_%s_40x30_to_97x71" % (wrap, tag), p, noise(40, 30, 50), 97, 71)
        # llvmpipe's blit fast path: stock copying an RGBA8 target with NEAREST + clamp to edge, at a scale
        # (4.5x) where every other sample row / column lands exactly on a texel boundary
        p = write_preset(d, "shaders = 2\nshader0 = %s/crt/shaders/crt-pi.glsl\nfilter_linear0 = true\nscale_type0 = source\nscale0 = 2.0\n"
                            "shader1 = %s/stock.glsl\nfilter_linear1 = false\nwrap_mode1 = clamp_to_edge\n" % (GLSL, GLSL))
        run_case("blit_nearest_60x45_to_540x405", p, noise(60, 45, 51), 540, 405)
        # ... and its LINEAR filter: several 64x64 target tiles (edge tiles lerp vertically first, inner
        # tiles horizontally first), up- and down-scaled
        p = write_preset(d, "shaders = 2\nshader0 = %s/crt/shaders/crt-pi.glsl\nfilter_linear0 = true\nscale_type0 = source\nscale0 = 2.0\n"
                            "shader1 = %s/stock.glsl\nfilter_linear1 = true\nwrap_mode1 = clamp_to_edge\n" % (GLSL, GLSL))
        run_case("blit_linear_60x45_to_540x405", p, noise(60, 45, 51), 540, 405)
        run_case("blit_linear_160x120_to_233x150", p, noise(160, 120, 52), 233, 150)


GLPRESENT = os.path.join(ROOT, "oracle", "_ref", "glpresent")


def gl_present(src, filt, dw, dh, dfmt, vp=None, flip=0, b=1.0, c=1.0):
    """One off-screen renderTexture draw on llvmpipe (oracle/glrun/glpresent.cpp)."""
    sh, sw, ch = src.shape
    vp = vp or (0, 0, dw, dh)
    with tempfile.TemporaryDirectory() as d:
        src.tofile(os.path.join(d, "in.raw"))
        r = subprocess.run([GLPRESENT, os.path.join(d, "in.raw"), str(sw), str(sh), "rgb" if ch == 3 else "rgba", filt,
                            str(dw), str(dh), dfmt] + [str(v) for v in vp] + [str(flip), repr(float(b)), repr(float(c))],
                           capture_output=True)
        if r.returncode or r.stderr:
            raise RuntimeError(r.stderr.decode())
    return np.frombuffer(r.stdout, np.float32 if dfmt == "f32" else np.uint8).reshape(dh, dw, 4).copy()


def overscan_vp(fbo_w, fbo_h, pct_x, pct_y):
    # FrameCapturePipeline.cpp:205-216 in float32
    f = np.float32
    ox = max(f(0), min(f(0.45), f(pct_x) / f(100))); oy = max(f(0), min(f(0.45), f(pct_y) / f(100)))
    fx = f(1) - f(2) * ox; fy = f(1) - f(2) * oy
    w = f(fbo_w) / fx; h = f(fbo_h) / fy
    return (int((f(fbo_w) - w) / f(2)), int((f(fbo_h) - h) / f(2)), int(w), int(h))


def case_present():
    """OpenGLRenderer::renderTexture off-screen (OpenGLRenderer.cpp:378-470) as FrameCapturePipeline uses
    it: source pre-pass (FCP.cpp:160-250), output resize (:413-505), brightness/contrast bake (:739-804)."""
    def save(name, src, out, **meta):
        np.savez_compressed(os.path.join(HERE, name + ".npz"), src=src, out=out, **{k: np.array(v) for k, v in meta.items()})
        print("wrote", name, src.shape, out.shape)

    rgba = lambda w, h, seed: np.random.default_rng(seed).integers(0, 256, (h, w, 4), dtype=np.uint8)
    # pre-pass: GL_RGB source, NEAREST, GL_RGB target
    src = mixed(320, 240, 60)
    save("present_prepass_down_320x240_to_256x224", src, gl_present(src, "nearest", 256, 224, "rgb"), filt="nearest", dst="rgbx8",
         vp=(0, 0, 256, 224), overscan=(0.0, 0.0), b=1.0, c=1.0, flip=0)
    vp = overscan_vp(256, 224, 5.0, 3.0)
    save("present_prepass_overscan_down_320x240_to_256x224", src, gl_present(src, "nearest", 256, 224, "rgb", vp), filt="nearest",
         dst="rgbx8", vp=vp, overscan=(5.0, 3.0), b=1.0, c=1.0, flip=0)
    src = noise(161, 120, 61)
    vp = overscan_vp(161, 120, 12.5, 0.0)
    save("present_prepass_overscan_161x120", src, gl_present(src, "nearest", 161, 120, "rgb", vp), filt="nearest", dst="rgbx8",
         vp=vp, overscan=(12.5, 0.0), b=1.0, c=1.0, flip=0)
    # resize: RGBA8 source (a render target), LINEAR, GL_RGBA target
    src = rgba(64, 48, 62)
    save("present_resize_64x48_to_160x100", src, gl_present(src, "linear", 160, 100, "rgba"), filt="linear", dst="rgba8",
         vp=(0, 0, 160, 100), b=1.0, c=1.0, flip=0)
    src = rgba(160, 120, 63)
    small = gl_present(src, "linear", 97, 71, "rgba")
    save("present_resize_160x120_to_97x71", src, small, filt="linear", dst="rgba8", vp=(0, 0, 97, 71), b=1.0, c=1.0, flip=0)
    # bake at the same size, then the reference's resize -> bake chain (two draws)
    src = rgba(120, 90, 64)
    save("present_bake_120x90", src, gl_present(src, "linear", 120, 90, "rgba", b=1.2, c=0.9), filt="linear", dst="rgba8",
         vp=(0, 0, 120, 90), b=1.2, c=0.9, flip=0)
    src = rgba(160, 120, 65)
    mid = gl_present(src, "linear", 232, 150, "rgba")
    save("present_resize_bake_160x120_to_232x150", src, gl_present(mid, "linear", 232, 150, "rgba", b=0.85, c=1.25), mid=mid,
         filt="linear", dst="rgba8", vp=(0, 0, 232, 150), b=1.0, c=1.0, bake=(0.85, 1.25), flip=0)
    # flipY uniform and a viewport inside the target (letterbox; the rest keeps the clear colour)
    src = rgba(96, 54, 66)
    save("present_flip_letterbox_96x54_to_120x100", src, gl_present(src, "nearest", 120, 100, "rgba", (0, 16, 120, 67), 1, 1.1, 1.1),
         filt="nearest", dst="rgba8", vp=(0, 16, 120, 67), b=1.1, c=1.1, flip=1)
    # float target: the program's arithmetic before the UNORM8 store
    src = rgba(32, 32, 67)
    save("present_f32_32x32", src, gl_present(src, "linear", 32, 32, "f32", b=0.7, c=1.45), filt="linear", dst="f32",
         vp=(0, 0, 32, 32), b=0.7, c=1.45, flip=0)


def case_float():
    """The same shaders with every render target forced to RGBA32F (not the reference's formats): pins
    the arithmetic of every pass at float precision."""
    with tempfile.TemporaryDirectory() as d:
        p = write_preset(d, 'shaders = 1\nshader0 = %s/scanlines/shaders/scanline.glsl\n' % GLSL)
        run_case("f32_scanline_64x48_to_160x100", p, mixed(64, 48, 2), 160, 100, f32=True)
        run_case("f32_crt_pi_80x60_to_250x190", GLSL + "/crt/crt-pi.glslp", noise(80, 60, 4), 250, 190, f32=True)
        run_case("f32_ntsc_svideo_96x64_to_256x192", GLSL + "/ntsc/ntsc-256px-svideo.glslp", mixed(96, 64, 8), 256, 192, frames=2, f32=True)
        run_case("f32_ntsc_320px_72x40_to_320x120", GLSL + "/ntsc/ntsc-320px.glslp", mixed(72, 40, 41), 320, 120, f32=True)
        run_case("f32_xbr_lv3_48x40_to_331x217", GLSL + "/xbr/xbr-lv3.glslp", pixel_art(48, 40, 11), 331, 217, f32=True)
        run_case("f32_mix_frames_48x36_to_120x90_f3", GLSL + "/motionblur/mix_frames.glslp", moving(48, 36, 3, 21), 120, 90, f32=True)
        luts = royale_luts(d)
        run_case("f32_crt_royale_64x48_to_128x96", GLSL + "/crt/crt-royale.glslp", mixed(64, 48, 5), 128, 96, luts=luts, f32=True)
    case_crt_royale_mask_active(f32=True)


def case_royale_fake_bloom():
    """crt/crt-royale-fake-bloom.glslp (9 passes): crt-royale's passes 0-7 and its last pass, with the two
    PHOSPHOR_BLOOM_FAKE variants of bloom-approx and scanlines-horizontal-apply-mask.  Rendered as llvmpipe
    does (pass 6 discards everything, see case_crt_royale_mask_active) and with that pass active."""
    import shutil
    with tempfile.TemporaryDirectory() as d:
        luts = royale_luts(d)
        run_case("crt_royale_fake_bloom_160x120_to_320x240", GLSL + "/crt/crt-royale-fake-bloom.glslp", mixed(160, 120, 5), 320, 240,
                 frames=2, luts=luts)
        dst = os.path.join(d, "shaders_glsl")
        os.makedirs(os.path.join(dst, "crt", "shaders"))
        shutil.copytree(GLSL + "/crt/shaders/crt-royale", dst + "/crt/shaders/crt-royale")
        shutil.copytree(GLSL + "/blurs", dst + "/blurs")
        shutil.copy(GLSL + "/crt/crt-royale-fake-bloom.glslp", dst + "/crt/crt-royale-fake-bloom.glslp")
        f = dst + "/crt/shaders/crt-royale/src/crt-royale-mask-resize-horizontal.glsl"
        txt = open(f).read()
        needle = "max(tile_uv_wrap.x, tile_uv_wrap.y) <= mask_resize_num_tiles"
        assert txt.count(needle) == 1
        open(f, "w").write(txt.replace(needle, "0.0 <= mask_resize_num_tiles"))
        run_case("crt_royale_fake_bloom_maskon_128x96_to_400x300", dst + "/crt/crt-royale-fake-bloom.glslp", noise(128, 96, 6), 400, 300,
                 luts=luts)
        run_case("f32_crt_royale_fake_bloom_maskon_64x48_to_128x96", dst + "/crt/crt-royale-fake-bloom.glslp", mixed(64, 48, 5), 128, 96,
                 luts=luts, f32=True)


def case_hyllian_glow():
    """crt/crt-hyllian-glow.glslp: the reference's smoke-test default (6 passes, mip-mapped input on pass 3)."""
    P = GLSL + "/crt/crt-hyllian-glow.glslp"
    run_case("crt_hyllian_glow_96x64_to_256x192", P, mixed(96, 64, 70), 256, 192)
    run_case("crt_hyllian_glow_80x60_to_250x190", P, noise(80, 60, 71), 250, 190)      # viewport / 4 is not integral
    run_case("crt_hyllian_glow_params_64x48_to_200x150", P, mixed(64, 48, 72), 200, 150,
             params=[("BEAM_PROFILE", 3.0), ("HFILTER_SHARPNESS", 0.6), ("CRT_ANTI_RINGING", 0.5), ("PHOSPHOR_LAYOUT", 2.0),
                     ("MASK_INTENSITY", 0.7), ("GLOW_ROLLOFF", 2.2), ("BLOOM_STRENGTH", 0.6)])
    run_case("f32_crt_hyllian_glow_64x48_to_160x120", P, mixed(64, 48, 73), 160, 120, f32=True)
    run_case("f32_crt_hyllian_glow_64x48_to_150x110", P, noise(64, 48, 74), 150, 110, f32=True,
             params=[("HFILTER_SHARPNESS", 0.7), ("CRT_ANTI_RINGING", 0.6), ("MASK_INTENSITY", 0.7), ("PHOSPHOR_LAYOUT", 5.0)])


def case_side_by_side():
    S = GLSL + "/stereoscopic-3d/"
    run_case("side_by_side_64x48_to_320x240", S + "side-by-side.glslp", mixed(64, 48, 250), 320, 240)
    run_case("sbs_warp_mobile_64x36_to_320x180", S + "sbs-warp-mobile-16x9.glslp", mixed(64, 36, 251), 320, 180)
    with tempfile.TemporaryDirectory() as d:
        p = write_preset(d, 'shaders = 1\nshader0 = %s/stereoscopic-3d/shaders/side-by-side-simple.glsl\n' % GLSL)
        prm = [("eye_sep", 0.2), ("y_loc", 0.15), ("BOTH", 0.4), ("ana_zoom", 0.9), ("WIDTH", 2.5), ("HEIGHT", 1.5), ("warpX", 0.2), ("warpY", 0.05), ("pulfrich", 0.25)]
        run_case("side_by_side_bare_params_40x30_to_233x171", p, noise(40, 30, 252), 233, 171, params=prm)
        run_case("f32_side_by_side_bare_params_48x36_to_200x150", p, mixed(48, 36, 253), 200, 150, params=prm, f32=True)


def case_sameboy_lcd():
    H = GLSL + "/handheld/"
    run_case("sameboy_lcd_64x48_to_320x240", H + "sameboy-lcd.glslp", mixed(64, 48, 260), 320, 240)
    run_case("sameboy_lcd_params_40x30_to_233x171", H + "sameboy-lcd.glslp", noise(40, 30, 261), 233, 171,
             params=[("COLOR_LOW", 0.6), ("COLOR_HIGH", 1.2), ("SCANLINE_DEPTH", 0.4)])
    run_case("f32_sameboy_lcd_48x36_to_200x150", H + "sameboy-lcd.glslp", mixed(48, 36, 262), 200, 150, f32=True)
    run_case("sameboy_lcd_gbc_color_motionblur_48x36_to_200x150_f4", H + "sameboy-lcd-gbc-color-motionblur.glslp", moving(48, 36, 4, 263), 200, 150)


def case_crt_consumer():
    C = GLSL + "/crt/crt-consumer.glslp"
    run_case("crt_consumer_64x48_to_320x240", C, mixed(64, 48, 270), 320, 240)
    run_case("crt_consumer_params_40x30_to_233x171_f2", C, noise(40, 30, 271), 233, 171, frames=2,
             params=[("blurx", 0.6), ("warpx", 0.06), ("warpy", 0.08), ("corner", 0.05), ("Shadowmask", 2.0), ("masksize", 2.0), ("slotmask", 1.0), ("slotwidth", 3.0),
                     ("double_slot", 2.0), ("GAMMA_IN", 2.2), ("GAMMA_OUT", 2.0), ("glow", 0.1), ("sat", 1.2), ("contrast", 1.1), ("nois", 8.0), ("WP", -20.0),
                     ("inter", 1.0), ("vignette", 1.0), ("vpower", 0.3), ("vstr", 30.0)])
    run_case("f32_crt_consumer_48x36_to_200x150", C, mixed(48, 36, 272), 200, 150, f32=True)


def case_reverse_aa():
    A = GLSL + "/anti-aliasing/reverse-aa.glslp"
    run_case("reverse_aa_64x48_to_320x240", A, mixed(64, 48, 280), 320, 240)
    run_case("reverse_aa_params_40x30_to_233x171", A, noise(40, 30, 281), 233, 171, params=[("REVERSEAA_SHARPNESS", 3.0)])
    run_case("f32_reverse_aa_48x36_to_200x150", A, mixed(48, 36, 282), 200, 150, f32=True)


def case_advanced_aa():
    A = GLSL + "/anti-aliasing/advanced-aa.glslp"
    run_case("advanced_aa_64x48_to_320x240", A, mixed(64, 48, 290), 320, 240)
    run_case("advanced_aa_params_40x30_to_233x171", A, noise(40, 30, 291), 233, 171, params=[("AA_RESOLUTION_X", 64.0), ("AA_RESOLUTION_Y", 48.0)])
    run_case("f32_advanced_aa_48x36_to_200x150", A, mixed(48, 36, 292), 200, 150, f32=True)


def case_lottes():
    L, F = GLSL + "/crt/crt-lottes.glslp", GLSL + "/crt/fakelottes.glslp"
    run_case("crt_lottes_64x48_to_320x240", L, mixed(64, 48, 230), 320, 240)
    run_case("crt_lottes_params_40x30_to_233x171", L, noise(40, 30, 231), 233, 171,
             params=[("hardScan", -6.0), ("hardPix", -4.0), ("warpX", 0.06), ("warpY", 0.02), ("maskDark", 0.7), ("maskLight", 1.3), ("scaleInLinearGamma", 0.0),
                     ("shadowMask", 1.0), ("brightBoost", 1.2), ("bloomAmount", 0.3), ("shape", 3.0)])
    for m in (0.0, 2.0, 4.0):
        run_case("crt_lottes_mask%d_48x36_to_200x150" % int(m), L, mixed(48, 36, 232), 200, 150, params=[("shadowMask", m)])
    run_case("f32_crt_lottes_48x36_to_200x150", L, mixed(48, 36, 233), 200, 150, f32=True)
    run_case("fakelottes_64x48_to_320x240", F, mixed(64, 48, 234), 320, 240)
    run_case("fakelottes_params_40x30_to_233x171", F, noise(40, 30, 235), 233, 171,
             params=[("shadowMask", 3.0), ("SCANLINE_SINE_COMP_B", 0.6), ("warpX", 0.05), ("maskDark", 0.8), ("crt_gamma", 2.2), ("monitor_gamma", 2.0),
                     ("SCANLINE_SINE_COMP_A", 0.05), ("SCANLINE_BASE_BRIGHTNESS", 0.9)])
    run_case("f32_fakelottes_48x36_to_200x150", F, mixed(48, 36, 236), 200, 150, f32=True)


def case_jinc2():
    run_case("jinc2_sharper_64x48_to_320x240", GLSL + "/windowed/jinc2-sharper.glslp", mixed(64, 48, 220), 320, 240)
    run_case("jinc2_sharper_40x30_to_233x171", GLSL + "/windowed/jinc2-sharper.glslp", noise(40, 30, 221), 233, 171)
    run_case("f32_jinc2_sharper_48x36_to_200x150", GLSL + "/windowed/jinc2-sharper.glslp", mixed(48, 36, 222), 200, 150, f32=True)
    run_case("tvout_jinc_sharpen_64x48_to_320x240_f2", GLSL + "/presets/tvout/tvout-jinc-sharpen.glslp", mixed(64, 48, 223), 320, 240, frames=2)


def case_interlacing():
    """misc/interlacing.glsl: behind tvout + image-adjustment (pass index 2), alone on a > 400-line source over three frames (the field alternates with
    FrameCount) and on a short source (fixed lines), parameters moved."""
    run_case("tvout_interlacing_64x48_to_320x240", GLSL + "/presets/tvout+interlacing/tvout+interlacing.glslp", mixed(64, 48, 210), 320, 240)
    with tempfile.TemporaryDirectory() as d:
        p = write_preset(d, 'shaders = 1\nshader0 = %s/misc/interlacing.glsl\nfilter_linear0 = false\n' % GLSL)
        run_case("interlacing_bare_40x420_to_160x420_f3", p, mixed(40, 420, 211), 160, 420, frames=3, params=[("percent", 0.25)])
        run_case("interlacing_bare_40x420_to_120x300_f2", p, noise(40, 420, 212), 120, 300, frames=2, params=[("percent", 0.5), ("top_field_first", 1.0)])
        run_case("interlacing_bare_48x36_to_200x150", p, mixed(48, 36, 213), 200, 150, params=[("percent", 0.3), ("enable_480i", 0.0)])
        run_case("f32_interlacing_bare_40x420_to_160x420_f2", p, mixed(40, 420, 214), 160, 420, frames=2, f32=True, params=[("percent", 0.25)])


def case_tvout():
    """crt/shaders/tvout-tweaks.glsl and misc/image-adjustment.glsl: in the reference's presets (image-adjustment at pass index 3 in the 4-pass one) and
    alone with their parameters moved (composite cross-talk and colour levels on; zoom, shift, overscan, masks, grain by FrameCount, sharpen)."""
    P = GLSL + "/presets/"
    run_case("tvout_64x48_to_320x240", P + "tvout/tvout.glslp", mixed(64, 48, 200), 320, 240)
    run_case("tvout_ntsc_256px_svideo_72x40_to_300x171", P + "tvout/tvout+ntsc-256px-svideo.glslp", mixed(72, 40, 201), 300, 171)
    run_case("retro_v2_image_adjustment_40x30_to_233x171", P + "retro-v2+image-adjustment.glslp", noise(40, 30, 202), 233, 171)
    with tempfile.TemporaryDirectory() as d:
        pt = write_preset(d, 'shaders = 1\nshader0 = %s/crt/shaders/tvout-tweaks.glsl\nfilter_linear0 = false\n' % GLSL)
        tprm = [("TVOUT_RESOLUTION", 192.0), ("TVOUT_COMPOSITE_CONNECTION", 1.0), ("TVOUT_TV_COLOR_LEVELS", 1.0), ("TVOUT_RESOLUTION_Y", 224.0),
                ("TVOUT_RESOLUTION_I", 64.0), ("TVOUT_RESOLUTION_Q", 32.0)]
        run_case("tvout_tweaks_bare_params_64x48_to_256x192", pt, mixed(64, 48, 203), 256, 192, params=tprm)
        run_case("f32_tvout_tweaks_bare_params_48x36_to_200x150", pt, noise(48, 36, 204), 200, 150, params=tprm, f32=True)
        pi = write_preset(d, 'shaders = 1\nshader0 = %s/misc/image-adjustment.glsl\nfilter_linear0 = false\n' % GLSL)
        iprm = [("ia_target_gamma", 2.4), ("ia_monitor_gamma", 2.0), ("ia_overscan_percent_x", 4.0), ("ia_overscan_percent_y", -3.0), ("ia_saturation", 1.3),
                ("ia_contrast", 1.1), ("ia_luminance", 0.9), ("ia_black_level", 0.03), ("ia_bright_boost", 0.1), ("ia_R", 1.1), ("ia_G", 0.95), ("ia_B", 1.05),
                ("ia_ZOOM", 1.2), ("ia_XPOS", 0.03), ("ia_YPOS", -0.02), ("ia_TOPMASK", 0.05), ("ia_BOTMASK", 0.04), ("ia_LMASK", 0.03), ("ia_RMASK", 0.02),
                ("ia_GRAIN_STR", 12.0), ("ia_SHARPEN", 0.4)]   # ia_FLIP_*: the shader moves the quad half off the target (not restated, refused)
        run_case("image_adjustment_bare_params_64x48_to_256x192_f3", pi, mixed(64, 48, 205), 256, 192, params=iprm, frames=3)
        run_case("f32_image_adjustment_bare_params_48x36_to_200x150_f2", pi, noise(48, 36, 206), 200, 150, params=iprm, f32=True, frames=2)


def case_ntsc_gauss():
    N = GLSL + "/ntsc/ntsc-256px-svideo-gauss-scanline.glslp"
    run_case("ntsc_gauss_scanline_96x64_to_320x240", N, mixed(96, 64, 190), 320, 240)
    run_case("ntsc_gauss_scanline_params_72x40_to_300x171", N, noise(72, 40, 191), 300, 171, params=[("NTSC_CRT_GAMMA", 2.2), ("NTSC_DISPLAY_GAMMA", 1.8)])
    run_case("f32_ntsc_gauss_scanline_72x40_to_256x160", N, mixed(72, 40, 192), 256, 160, f32=True)


def case_sameboy():
    pal = np.load(os.path.join(HERE, "lut_palette_synthetic.npy"))
    with tempfile.TemporaryDirectory() as d:
        raw = os.path.join(d, "pal.rgba")
        pal.tofile(raw)
        fr = moving(48, 36, 9, 240)
        fr[..., 0] = (fr[..., 0] // 85) * 85
        run_case("sameboy_dmg_response_time_48x36_to_48x36_f9", GLSL + "/handheld/sameboy-dmg-response-time.glslp", fr, 48, 36,
                 luts=[("COLOR_PALETTE", (raw, pal.shape[1], pal.shape[0]))])
        run_case("sameboy_dmg_response_time_48x36_to_131x77_f4", GLSL + "/handheld/sameboy-dmg-response-time.glslp", fr[:4], 131, 77,
                 luts=[("COLOR_PALETTE", (raw, pal.shape[1], pal.shape[0]))])


def case_crt_potato():
    m = np.load(os.path.join(HERE, "lut_potato_mask_synthetic.npy"))
    with tempfile.TemporaryDirectory() as d:
        raw = os.path.join(d, "m.rgba")
        m.tofile(raw)
        luts = [("MASK", (raw, m.shape[1], m.shape[0]))]
        run_case("crt_potato_64x48_to_320x240", GLSL + "/crt/crt-potato-cool.glslp", mixed(64, 48, 180), 320, 240, luts=luts)
        run_case("crt_potato_40x30_to_233x171", GLSL + "/crt/crt-potato-cool.glslp", noise(40, 30, 181), 233, 171, luts=luts)
        run_case("f32_crt_potato_48x36_to_240x200", GLSL + "/crt/crt-potato-cool.glslp", mixed(48, 36, 182), 240, 200, luts=luts, f32=True)


def case_gb_palette():
    pal = np.load(os.path.join(HERE, "lut_palette_synthetic.npy"))
    with tempfile.TemporaryDirectory() as d:
        raw = os.path.join(d, "pal.rgba")
        pal.tofile(raw)
        grey = mixed(64, 48, 170)
        grey[..., 0] = (grey[..., 0] // 85) * 85      # the four Game Boy grey levels in red, plus everything in between on the gradient band
        grey[16:22] = mixed(64, 48, 171)[16:22]
        run_case("gb_palette_dmg_64x48_to_64x48", GLSL + "/handheld/gb-palette-dmg.glslp", grey, 64, 48, luts=[("COLOR_PALETTE", (raw, pal.shape[1], pal.shape[0]))])
        run_case("gb_palette_dmg_64x48_to_201x155", GLSL + "/handheld/gb-palette-dmg.glslp", grey, 201, 155, luts=[("COLOR_PALETTE", (raw, pal.shape[1], pal.shape[0]))])


def case_reshade_lut():
    """reshade/lut.glslp (16 slices) and reshade/gba.glslp (32 slices, LUT_Size from the preset file) on synthetic grades."""
    with tempfile.TemporaryDirectory() as d:
        luts = {}
        for n in (16, 32):
            img = np.load(os.path.join(HERE, "lut_color%d_synthetic.npy" % n))
            raw = os.path.join(d, "lut%d.rgba" % n)
            img.tofile(raw)
            luts[n] = [("SamplerLUT", (raw, img.shape[1], img.shape[0]))]
        run_case("reshade_lut_64x48_to_160x120", GLSL + "/reshade/lut.glslp", mixed(64, 48, 160), 160, 120, luts=luts[16])
        run_case("reshade_gba_40x30_to_97x61", GLSL + "/reshade/gba.glslp", noise(40, 30, 161), 97, 61, luts=luts[32])
        run_case("f32_reshade_lut_48x36_to_131x77", GLSL + "/reshade/lut.glslp", mixed(48, 36, 162), 131, 77, luts=luts[16], f32=True)


def case_imgborder():
    """borders/: imgborder-{sgb,gameboy-player}.glsl (one shader text) - alone, in front of crt-geom, and with every parameter moved."""
    B = GLSL + "/borders/"
    border = np.load(os.path.join(HERE, "lut_border_synthetic.npy"))
    with tempfile.TemporaryDirectory() as d:
        raw = os.path.join(d, "border.rgba")
        border.tofile(raw)
        luts = [("BORDER", (raw, border.shape[1], border.shape[0]))]
        run_case("imgborder_gameboy_player_60x40_to_304x224", B + "gameboy-player/gameboy-player.glslp", mixed(60, 40, 150), 304, 224, luts=luts)
        run_case("imgborder_sgb_crt_geom_1x_40x36_to_256x224", B + "sgb/sgb-crt-geom-1x.glslp", mixed(40, 36, 151), 256, 224, luts=luts)
        run_case("console_border_ngpc_3x_40x38_to_300x200", GLSL + "/handheld/console-border/ngpc-3x.glslp", mixed(40, 38, 154), 300, 200, luts=luts)
        p = write_preset(d, 'shaders = 1\nshader0 = %s/borders/resources/imgborder-sgb.glsl\ntextures = "BORDER"\nBORDER = "sgb.png"\nBORDER_linear = true\n' % GLSL)
        prm = [("box_scale", 2.0), ("location_x", 0.45), ("location_y", 0.6), ("in_res_x", 120.0), ("in_res_y", 90.0), ("border_on_top", 1.0),
               ("border_zoom_x", 1.3), ("border_zoom_y", 0.8), ("OS_MASK_TOP", 0.05), ("OS_MASK_BOTTOM", 0.1), ("OS_MASK_LEFT", 0.02), ("OS_MASK_RIGHT", 0.07)]
        run_case("imgborder_sgb_bare_params_40x30_to_233x171", p, noise(40, 30, 152), 233, 171, luts=luts, params=prm)
        run_case("f32_imgborder_sgb_bare_params_40x30_to_233x171", p, noise(40, 30, 153), 233, 171, luts=luts, params=prm, f32=True)


def case_lcd_grid():
    G = GLSL + "/handheld/lcd-grid.glslp"
    run_case("lcd_grid_64x48_to_320x240", G, mixed(64, 48, 140), 320, 240)
    run_case("lcd_grid_params_40x30_to_233x171", G, noise(40, 30, 141), 233, 171, params=[("GRID_STRENGTH", 0.3), ("gamma", 1.7)])
    run_case("f32_lcd_grid_params_48x36_to_240x180", G, mixed(48, 36, 142), 240, 180, f32=True, params=[("GRID_STRENGTH", 0.2), ("gamma", 2.6)])
    border = np.load(os.path.join(HERE, "lut_border_synthetic.npy"))
    with tempfile.TemporaryDirectory() as d:
        raw = os.path.join(d, "border.rgba")
        border.tofile(raw)
        run_case("console_border_gba_3x_48x32_to_300x200_f9", GLSL + "/handheld/console-border/gba-3x.glslp", moving(48, 32, 9, 143), 300, 200,
                 luts=[("BORDER", (raw, border.shape[1], border.shape[0]))])


def case_console_border():
    """handheld/console-border/: gb-pass-5.glsl lays a border image over the scaled frame.  The border here is the small synthetic
    image of tests/golden/lut_border_synthetic.npy handed to the runner as raw texels (the reference's PNGs are 1-2 MB artwork)."""
    C = GLSL + "/handheld/console-border/"
    border = np.load(os.path.join(HERE, "lut_border_synthetic.npy"))
    with tempfile.TemporaryDirectory() as d:
        raw = os.path.join(d, "border.rgba")
        border.tofile(raw)
        luts = [("BORDER", (raw, border.shape[1], border.shape[0]))]
        run_case("console_border_gba_lcd_grid_v2_3x_48x32_to_300x200_f4", C + "gba-lcd-grid-v2-3x.glslp", moving(48, 32, 4, 130), 300, 200, luts=luts)
        run_case("console_border_gbc_retro_v2_2x_40x36_to_233x171_f3", C + "gbc-retro-v2-2x.glslp", moving(40, 36, 3, 131), 233, 171, luts=luts)
        run_case("f32_console_border_gbc_retro_v2_2x_40x36_to_233x171_f3", C + "gbc-retro-v2-2x.glslp", moving(40, 36, 3, 132), 233, 171, luts=luts, f32=True)


def case_agb001():
    run_case("agb001_48x36_to_250x190", GLSL + "/handheld/agb001.glslp", mixed(48, 36, 120), 250, 190)
    run_case("f32_agb001_40x30_to_233x171", GLSL + "/handheld/agb001.glslp", noise(40, 30, 121), 233, 171, f32=True)
    run_case("agb001_gba_color_motionblur_48x36_to_250x190_f4", GLSL + "/handheld/agb001-gba-color-motionblur.glslp", moving(48, 36, 4, 122), 250, 190)


def case_retro_v2():
    R = GLSL + "/handheld/retro-v2.glslp"
    run_case("retro_v2_64x48_to_320x240", R, mixed(64, 48, 110), 320, 240)
    run_case("retro_v2_params_40x30_to_233x171", R, noise(40, 30, 111), 233, 171, params=[("RETRO_PIXEL_SIZE", 0.55)])
    run_case("f32_retro_v2_48x36_to_240x180", R, mixed(48, 36, 112), 240, 180, f32=True)
    run_case("retro_v2_gba_color_48x36_to_240x180", GLSL + "/presets/retro-v2+gba-color.glslp", mixed(48, 36, 113), 240, 180)
    run_case("retro_v2_vba_color_40x30_to_233x171", GLSL + "/presets/retro-v2+vba-color.glslp", noise(40, 30, 114), 233, 171)


def case_lcd_grid_v2():
    """handheld/lcd-grid-v2.glslp and its chains: with a colour pass behind it, and with motionblur/response-time in front (frame
    history through a pass 0 that is not the last pass)."""
    H = GLSL + "/handheld/"
    run_case("lcd_grid_v2_64x48_to_320x240", H + "lcd-grid-v2.glslp", mixed(64, 48, 90), 320, 240)
    run_case("lcd_grid_v2_40x30_to_233x171", H + "lcd-grid-v2.glslp", noise(40, 30, 91), 233, 171)
    run_case("lcd_grid_v2_params_48x36_to_240x180", H + "lcd-grid-v2.glslp", mixed(48, 36, 92), 240, 180,
             params=[("BGR", 1.0), ("gain", 1.2), ("gamma", 2.6), ("outgamma", 1.9), ("blacklevel", 0.07), ("ambient", 0.03), ("RSUBPIX_G", 0.15),
                     ("GSUBPIX_B", 0.1), ("BSUBPIX_R", 0.05)])
    run_case("f32_lcd_grid_v2_48x36_to_240x180", H + "lcd-grid-v2.glslp", mixed(48, 36, 93), 240, 180, f32=True)
    run_case("f32_lcd_grid_v2_params_40x30_to_233x171", H + "lcd-grid-v2.glslp", noise(40, 30, 94), 233, 171, f32=True,
             params=[("BGR", 1.0), ("gain", 0.8), ("gamma", 3.4), ("outgamma", 2.4), ("blacklevel", 0.02), ("ambient", 0.1)])
    with tempfile.TemporaryDirectory() as d:
        # without the preset files' own parameter block (which overrides custom values at draw time): every parameter moves
        p = write_preset(d, 'shaders = 1\nshader0 = %s/handheld/shaders/lcd-cgwg/lcd-grid-v2.glsl\nfilter_linear0 = false\nscale_type0 = viewport\n' % GLSL)
        prm = [("BGR", 1.0), ("gain", 1.2), ("gamma", 2.6), ("outgamma", 1.9), ("blacklevel", 0.07), ("ambient", 0.03), ("RSUBPIX_R", 0.9),
               ("RSUBPIX_G", 0.15), ("GSUBPIX_G", 0.8), ("GSUBPIX_B", 0.1), ("BSUBPIX_R", 0.05), ("BSUBPIX_B", 0.85)]
        run_case("lcd_grid_v2_bare_params_48x36_to_240x180", p, mixed(48, 36, 99), 240, 180, params=prm)
        run_case("lcd_grid_v2_bare_defaults_40x30_to_97x61", p, noise(40, 30, 100), 97, 61)
        run_case("f32_lcd_grid_v2_bare_params_40x30_to_233x171", p, noise(40, 30, 101), 233, 171, f32=True, params=prm)
    run_case("lcd_grid_v2_gba_color_48x36_to_240x180", H + "lcd-grid-v2-gba-color.glslp", mixed(48, 36, 95), 240, 180)
    run_case("lcd_grid_v2_gbc_color_48x36_to_200x150", H + "lcd-grid-v2-gbc-color.glslp", mixed(48, 36, 96), 200, 150)
    run_case("lcd_grid_v2_psp_color_motionblur_48x36_to_200x150_f5", H + "lcd-grid-v2-psp-color-motionblur.glslp", moving(48, 36, 5, 97), 200, 150)
    run_case("lcd_grid_v2_motionblur_48x36_to_200x150_f9", H + "lcd-grid-v2-motionblur.glslp", moving(48, 36, 9, 98), 200, 150)


def case_handheld_color():
    """handheld/{gba,gbc,gbc-gambatte,nds,palm,psp,vba}-color.glslp: 8-bit at two sizes and one float run each."""
    for k, n in enumerate(("gba", "gbc", "gbc-gambatte", "nds", "palm", "psp", "vba")):
        P = GLSL + "/handheld/%s-color.glslp" % n
        t = n.replace("-", "_")
        run_case("%s_color_64x48_to_160x120" % t, P, mixed(64, 48, 60 + k), 160, 120)
        run_case("f32_%s_color_48x36_to_131x77" % t, P, mixed(48, 36, 70 + k), 131, 77, f32=True)
    run_case("gba_color_params_40x30_to_97x61", GLSL + "/handheld/gba-color.glslp", noise(40, 30, 80), 97, 61, params=[("darken_screen", 0.35)])
    run_case("gbc_color_params_40x30_to_97x61", GLSL + "/handheld/gbc-color.glslp", noise(40, 30, 81), 97, 61, params=[("lighten_screen", 0.25)])
    run_case("vba_color_params_40x30_to_97x61", GLSL + "/handheld/vba-color.glslp", noise(40, 30, 82), 97, 61, params=[("darken_screen", -0.4)])


def case_history_more():
    """Two more frame-history shaders: stereoscopic-3d/shutter-to-side-by-side.glslp (PrevTexture; FrameCount parity selects
    the eye) and misc/anti-flicker.glsl (PrevTexture, Prev1Texture; no preset in the reference's tree: one-pass chain)."""
    S = GLSL + "/stereoscopic-3d/shutter-to-side-by-side.glslp"
    prm = [("ZOOM", 0.9), ("vert_pos", -0.03), ("horz_pos", 0.05), ("separation", 0.1), ("flicker", 0.5), ("height_mod", 1.3), ("swap_eye", 1.0)]
    run_case("shutter_3d_48x36_to_120x90_f4", S, moving(48, 36, 4, 51), 120, 90)
    run_case("shutter_3d_params_48x36_to_131x77_f5", S, moving(48, 36, 5, 52), 131, 77, params=prm)
    run_case("f32_shutter_3d_params_48x36_to_131x77_f5", S, moving(48, 36, 5, 53), 131, 77, params=prm, f32=True)
    with tempfile.TemporaryDirectory() as d:
        p = write_preset(d, 'shaders = 1\nshader0 = %s/misc/anti-flicker.glsl\nfilter_linear0 = false\n' % GLSL)
        run_case("anti_flicker_48x36_to_120x90_f6", p, flicker(48, 36, 6, 54), 120, 90)
        run_case("anti_flicker_params_40x30_to_40x30_f5", p, flicker(40, 30, 5, 55), 40, 30, params=[("lum_diff_thresh", 0.2)])
        run_case("f32_anti_flicker_48x36_to_120x90_f6", p, flicker(48, 36, 6, 56), 120, 90, f32=True)


def case_hyllian_layouts():
    """resolve2.glsl's twenty PHOSPHOR_LAYOUT masks (mask_weights, 129-400): one chain run per layout; the first five
    passes do not depend on it and are kept once, the last pass once per layout."""
    P = GLSL + "/crt/crt-hyllian-glow.glslp"
    rgb = mixed(48, 36, 75)
    merged = None
    for lay in range(20):
        tmp = "_tmp_hyllian_layout"
        run_case(tmp, P, rgb, 143, 101, params=[("PHOSPHOR_LAYOUT", float(lay)), ("MASK_INTENSITY", 0.7)])
        g = dict(np.load(os.path.join(HERE, tmp + ".npz")))
        os.remove(os.path.join(HERE, tmp + ".npz"))
        if merged is None:
            merged = {k: v for k, v in g.items() if k not in ("pass5", "sha256_last", "param_names", "param_values")}
        else:
            for i in range(5):
                assert np.array_equal(merged["pass%d" % i], g["pass%d" % i])
        merged["pass5_layout%d" % lay] = g["pass5"]
    merged["mask_intensity"] = np.float32(0.7)
    np.savez_compressed(os.path.join(HERE, "crt_hyllian_glow_layouts_48x36_to_143x101.npz"), **merged)
    print("wrote crt_hyllian_glow_layouts_48x36_to_143x101")


def case_xbr_lv2():
    P = GLSL + "/xbr/xbr-lv2.glslp"
    run_case("xbr_lv2_64x56_to_256x224", P, mixed(64, 56, 80), 256, 224)
    run_case("xbr_lv2_noise_40x36_to_240x216", P, noise(40, 36, 81), 240, 216)
    run_case("xbr_lv2_params_48x40_to_331x217", P, mixed(48, 40, 82), 331, 217, params=[("XBR_EQ_THRESHOLD", 25.0), ("XBR_LV2_COEFFICIENT", 1.4)])
    run_case("f32_xbr_lv2_48x40_to_331x217", P, mixed(48, 40, 83), 331, 217, f32=True)
    # "Preserve Small Details": the outer luma samples weighted with XBR_Y_WEIGHT * Y and the 7-term weighted distance
    run_case("xbr_lv2_details_64x56_to_256x224", P, mixed(64, 56, 84), 256, 224, params=[("small_details", 1.0)])
    run_case("xbr_lv2_details_noise_40x36_to_240x216", P, noise(40, 36, 85), 240, 216, params=[("small_details", 1.0), ("XBR_Y_WEIGHT", 60.0)])
    run_case("f32_xbr_lv2_details_48x40_to_331x217", P, pixel_art(48, 40, 86), 331, 217, f32=True, params=[("small_details", 1.0)])


def case_royale_ntsc():
    """crt/crt-royale-ntsc-*.glslp (14 passes): the scanlines-vertical pass sits at pass index 3, where the reference
    overrides TextureSize.y with the target's height (ShaderEngine.cpp:2418-2421); the last pass has mipmap_input."""
    with tempfile.TemporaryDirectory() as d:
        luts = royale_luts(d)
        run_case("crt_royale_ntsc_256px_svideo_96x64_to_320x240", GLSL + "/crt/crt-royale-ntsc-256px-svideo.glslp", mixed(96, 64, 90), 320, 240,
                 frames=2, luts=luts)
        run_case("crt_royale_ntsc_320px_composite_80x56_to_300x200", GLSL + "/crt/crt-royale-ntsc-320px-composite.glslp", noise(80, 56, 91), 300, 200,
                 luts=luts)


def case_stock_presets():
    """The reference's presets built from stock.glsl alone: bilinear (GL_RGB source, LINEAR) and
    sharp-bilinear-2x-prescale (NEAREST 2x, then a LINEAR copy of an RGBA8 target: llvmpipe's blit fast path)."""
    run_case("bilinear_64x48_to_237x171", GLSL + "/bilinear.glslp", mixed(64, 48, 95), 237, 171)
    run_case("sharp_bilinear_2x_64x48_to_300x210", GLSL + "/interpolation/sharp-bilinear-2x-prescale.glslp", mixed(64, 48, 96), 300, 210)
    run_case("sharp_bilinear_2x_120x90_to_160x100", GLSL + "/interpolation/sharp-bilinear-2x-prescale.glslp", noise(120, 90, 97), 160, 100)


def case_zfast():
    """crt/zfast-crt.glslp; its six parameters are the ones the reference hard-codes (ShaderEngine.cpp:2260-2294), so a
    value set by the user must not show (second case)."""
    P = GLSL + "/crt/zfast-crt.glslp"
    run_case("zfast_crt_96x64_to_301x217", P, mixed(96, 64, 100), 301, 217)
    run_case("zfast_crt_custom_ignored_80x60_to_320x240", P, noise(80, 60, 101), 320, 240, params=[("BLURSCALEX", 0.9), ("MASK_DARK", 0.6)])
    run_case("f32_zfast_crt_64x48_to_200x150", P, mixed(64, 48, 102), 200, 150, f32=True)


def case_easymode():
    P = GLSL + "/crt/crt-easymode.glslp"
    run_case("crt_easymode_96x64_to_301x217", P, mixed(96, 64, 110), 301, 217)
    run_case("crt_easymode_params_80x60_to_320x240", P, noise(80, 60, 111), 320, 240,
             params=[("SHARPNESS_H", 0.8), ("SHARPNESS_V", 0.6), ("MASK_DOT_WIDTH", 2.0), ("MASK_STAGGER", 3.0), ("MASK_SIZE", 2.0),
                     ("SCANLINE_BEAM_WIDTH_MIN", 1.0), ("SCANLINE_BEAM_WIDTH_MAX", 2.5), ("DILATION", 0.0), ("SCANLINE_STRENGTH", 0.7)])
    run_case("f32_crt_easymode_64x48_to_200x150", P, mixed(64, 48, 112), 200, 150, f32=True)


def case_nes_mini():
    P = GLSL + "/crt/crt-nes-mini.glslp"
    run_case("crt_nes_mini_96x64_to_301x217", P, mixed(96, 64, 120), 301, 217)
    run_case("crt_nes_mini_params_80x60_to_320x240", P, noise(80, 60, 121), 320, 240, params=[("SCANTHICK", 4.0), ("INTENSITY", 0.4), ("BRIGHTBOOST", 0.5)])
    run_case("f32_crt_nes_mini_64x48_to_200x150", P, mixed(64, 48, 122), 200, 150, f32=True)


def case_epx():
    """scalenx/epx.glslp: the only pass scales source x 2.0, so the last-pass rule leaves its size alone (not the viewport's)."""
    import numpy as np
    blocky = np.repeat(np.repeat(np.random.default_rng(140).integers(0, 4, (28, 40, 3), dtype=np.uint8) * 85, 2, 0), 2, 1)
    run_case("epx_80x56_to_300x200", GLSL + "/scalenx/epx.glslp", blocky, 300, 200)
    run_case("epx_mixed_64x48_to_64x48", GLSL + "/scalenx/epx.glslp", mixed(64, 48, 141), 64, 48)


def pixelart(w, h, seed):
    """Flat blocks of a six-colour palette, one-pixel diagonals and a noise patch: what ScaleFX's edge levels act on."""
    rng = np.random.default_rng(seed)
    pal = rng.integers(0, 256, (6, 3), dtype=np.uint8)
    low = rng.integers(0, 6, (h // 4 + 1, w // 4 + 1))
    img = pal[np.kron(low, np.ones((4, 4), int))[:h, :w]]
    for i in range(min(w, h)):
        img[i, (i * 2) % w] = pal[(i // 3) % 6]
    img[h // 2:h // 2 + 8, w // 2:w // 2 + 12] = rng.integers(0, 256, (8, 12, 3), dtype=np.uint8)
    return img


def case_scalefx():
    """scalefx/scalefx.glslp (5 passes, two RGBA32F metric targets, PassPrev2Texture / PassPrev5Texture = original frame);
    the last pass scales source x 3, so the viewport does not matter."""
    Q = GLSL + "/scalefx/scalefx.glslp"
    run_case("scalefx_48x40", Q, pixelart(48, 40, 1), 144, 120)
    run_case("scalefx_noise_37x29", Q, noise(37, 29, 171), 64, 64)
    run_case("scalefx_params_56x44", Q, pixelart(56, 44, 172), 100, 100, params=[("SFX_CLR", 0.35), ("SFX_SAA", 0.0), ("SFX_SCN", 0.0)])
    run_case("f32_scalefx_40x32", Q, pixelart(40, 32, 173), 120, 96, f32=True)


def case_crt_geom():
    """crt/crt-geom.glslp: curvature + Lanczos2 + beam profile; the vertex shader computes the stretch
    varyings (sin/cos/acos); a source of >= 400 lines turns the interlacing simulation on (FrameCount parity)."""
    P = GLSL + "/crt/crt-geom.glslp"
    run_case("crt_geom_96x64_to_301x217", P, mixed(96, 64, 130), 301, 217)
    run_case("crt_geom_params_80x60_to_320x240", P, noise(80, 60, 131), 320, 240,
             params=[("CRTgamma", 2.0), ("monitorgamma", 2.4), ("d", 2.0), ("R", 3.5), ("cornersize", 0.1), ("cornersmooth", 400.0),
                     ("x_tilt", 0.2), ("y_tilt", -0.15), ("overscan_x", 104.0), ("overscan_y", 97.0), ("DOTMASK", 0.5),
                     ("SHARPER", 2.0), ("scanline_weight", 0.25), ("lum", 0.1), ("SATURATION", 1.3)])
    run_case("crt_geom_flat_72x56_to_288x224", P, mixed(72, 56, 132), 288, 224, params=[("CURVATURE", 0.0), ("interlace_detect", 0.0)])
    run_case("crt_geom_interlace_40x400_to_160x300_f2", P, mixed(40, 400, 133), 160, 300, frames=2)
    run_case("f32_crt_geom_64x48_to_200x150", P, mixed(64, 48, 134), 200, 150, f32=True)
    run_case("f32_crt_geom_params_64x48_to_200x150", P, noise(64, 48, 135), 200, 150, f32=True,
             params=[("x_tilt", -0.3), ("y_tilt", 0.25), ("R", 1.5), ("d", 1.2), ("SATURATION", 0.7), ("lum", 0.2)])


def case_mip_rgba8():
    """mipmap_input on 8-bit textures: the GL_RGB source frame (mipmap_input0) and a plain RGBA8 render target, sampled
    by glow/blur_horiz (nine trilinear taps) at fractional LODs; llvmpipe generates and blends these levels in 8 bits."""
    with tempfile.TemporaryDirectory() as d:
        src = 'shaders = 1\nshader0 = %s/crt/shaders/glow/blur_horiz.glsl\nfilter_linear0 = true\nmipmap_input0 = true\nscale_type0 = source\nscale0 = %s\n'
        run_case("mip_source_96x64_s0.4", write_preset(d, src % (GLSL, "0.4")), noise(96, 64, 170), 200, 150)
        run_case("mip_source_125x95_s0.23", write_preset(d, src % (GLSL, "0.23")), mixed(125, 95, 171), 200, 150)
        fbo = ('shaders = 2\nshader0 = %s/stock.glsl\nfilter_linear0 = false\nscale_type0 = source\nscale0 = 1.0\n'
               'shader1 = %s/crt/shaders/glow/blur_horiz.glsl\nfilter_linear1 = true\nmipmap_input1 = true\nscale_type1 = source\nscale1 = %s\n')
        run_case("mip_rgba8_96x64_s0.37", write_preset(d, fbo % (GLSL, GLSL, "0.37")), noise(96, 64, 172), 200, 150)
        run_case("mip_rgba8_101x67_s0.6", write_preset(d, fbo % (GLSL, GLSL, "0.6")), mixed(101, 67, 173), 200, 150)


def case_mip_nearest():
    """mipmap_input WITHOUT filter_linear: GL_NEAREST_MIPMAP_NEAREST (ShaderEngine.cpp:1019-1030) on the GL_RGB source frame
    and on a plain RGBA8 render target - one level per quad, (exponent(rho^2) + 1) >> 1, NEAREST texel."""
    with tempfile.TemporaryDirectory() as d:
        src = 'shaders = 1\nshader0 = %s/crt/shaders/glow/blur_horiz.glsl\nfilter_linear0 = false\nmipmap_input0 = true\nscale_type0 = source\nscale0 = %s\n'
        run_case("mipnearest_source_96x64_s0.4", write_preset(d, src % (GLSL, "0.4")), noise(96, 64, 174), 200, 150)
        run_case("mipnearest_source_125x95_s0.17", write_preset(d, src % (GLSL, "0.17")), mixed(125, 95, 175), 200, 150)
        fbo = ('shaders = 2\nshader0 = %s/stock.glsl\nfilter_linear0 = false\nscale_type0 = source\nscale0 = 1.0\n'
               'shader1 = %s/crt/shaders/glow/blur_horiz.glsl\nfilter_linear1 = false\nmipmap_input1 = true\nscale_type1 = source\nscale1 = %s\n')
        run_case("mipnearest_rgba8_101x67_s0.6", write_preset(d, fbo % (GLSL, GLSL, "0.6")), mixed(101, 67, 176), 200, 150)


def case_lcd3x():
    Q = GLSL + "/handheld/lcd1x.glslp"
    run_case("lcd1x_64x48_to_192x144", Q, mixed(64, 48, 153), 192, 144)
    run_case("lcd1x_params_80x60_to_301x217", Q, noise(80, 60, 154), 301, 217, params=[("BRIGHTEN_SCANLINES", 3.0), ("BRIGHTEN_LCD", 1.5)])
    run_case("f32_lcd1x_64x48_to_200x150", Q, noise(64, 48, 155), 200, 150, f32=True)
    P = GLSL + "/handheld/lcd3x.glslp"
    run_case("lcd3x_64x48_to_192x144", P, mixed(64, 48, 150), 192, 144)
    run_case("lcd3x_params_80x60_to_301x217", P, noise(80, 60, 151), 301, 217, params=[("brighten_scanlines", 4.0), ("brighten_lcd", 1.5)])
    run_case("f32_lcd3x_64x48_to_200x150", P, noise(64, 48, 152), 200, 150, f32=True)


def case_bayer():
    P = GLSL + "/dithering/bayer-matrix-dithering.glslp"
    run_case("bayer_64x48_to_237x171", P, mixed(64, 48, 160), 237, 171)
    run_case("bayer_animated_80x60_to_320x240_f3", P, noise(80, 60, 161), 320, 240, frames=3, params=[("animate", 1.0), ("dither_size", 0.35)])


def case_interp():
    run_case("quilez_64x48_to_237x171", GLSL + "/interpolation/quilez.glslp", mixed(64, 48, 130), 237, 171)
    run_case("f32_quilez_64x48_to_200x150", GLSL + "/interpolation/quilez.glslp", noise(64, 48, 131), 200, 150, f32=True)
    run_case("smootheststep_64x48_to_237x171", GLSL + "/interpolation/smootheststep.glslp", mixed(64, 48, 135), 237, 171)
    run_case("f32_smootheststep_64x48_to_200x150", GLSL + "/interpolation/smootheststep.glslp", noise(64, 48, 136), 200, 150, f32=True)
    P = GLSL + "/interpolation/sharp-bilinear.glslp"
    run_case("sharp_bilinear_64x48_to_237x171", P, mixed(64, 48, 132), 237, 171)
    run_case("sharp_bilinear_manual_80x60_to_400x300", P, noise(80, 60, 133), 400, 300, params=[("AUTO_PRESCALE", 0.0), ("SHARP_BILINEAR_PRE_SCALE", 3.0)])
    run_case("f32_sharp_bilinear_64x48_to_200x150", P, noise(64, 48, 134), 200, 150, f32=True)


CASES = {"advanced_aa": case_advanced_aa, "reverse_aa": case_reverse_aa, "crt_consumer": case_crt_consumer, "sameboy_lcd": case_sameboy_lcd, "side_by_side": case_side_by_side, "sameboy": case_sameboy, "lottes": case_lottes, "jinc2": case_jinc2, "interlacing": case_interlacing, "tvout": case_tvout, "ntsc_gauss": case_ntsc_gauss, "crt_potato": case_crt_potato, "gb_palette": case_gb_palette, "reshade_lut": case_reshade_lut, "imgborder": case_imgborder, "lcd_grid": case_lcd_grid, "console_border": case_console_border, "agb001": case_agb001, "retro_v2": case_retro_v2, "lcd_grid_v2": case_lcd_grid_v2, "handheld_color": case_handheld_color, "history_more": case_history_more, "royale_fake_bloom_geom": case_royale_fake_bloom_geom, "hyllian_layouts": case_hyllian_layouts, "crt_royale_geom": case_crt_royale_geom, "motionblur": case_motionblur, "mip_rgba8": case_mip_rgba8, "history_size": case_history_size, "mip_nearest": case_mip_nearest, "crt_geom": case_crt_geom, "scalefx": case_scalefx, "bayer": case_bayer, "lcd3x": case_lcd3x, "epx": case_epx, "interp": case_interp, "nes_mini": case_nes_mini, "easymode": case_easymode, "zfast": case_zfast, "stock_presets": case_stock_presets, "royale_ntsc": case_royale_ntsc, "xbr_lv2": case_xbr_lv2, "hyllian_glow": case_hyllian_glow, "royale_fake_bloom": case_royale_fake_bloom, "present": case_present, "sampler_matrix": case_sampler_matrix, "float": case_float, "ntsc_family": case_ntsc_family, "feedback": case_feedback, "mix_frames": case_mix_frames, "ntsc": case_ntsc, "xbr": case_xbr, "scanline": case_scanline, "crt_pi": case_crt_pi, "crt_royale": case_crt_royale,
         "crt_royale_mask_active": case_crt_royale_mask_active}

if __name__ == "__main__":
    for c in (sys.argv[1:] or list(CASES)):
        CASES[c]()
