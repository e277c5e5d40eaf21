"""HIP shader chain (through the C ABI) against the golden vectors and the oracle.
Bit-exact: every stored byte must equal the oracle's, which itself is pinned to llvmpipe."""
import os

import numpy as np
import pytest

import chain_specs
from oracle_chain import run_chain

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")

GOLDEN_CASES = {
    "advanced_aa_64x48_to_320x240": "advanced-aa",
    "advanced_aa_params_40x30_to_233x171": "advanced-aa",
    "reverse_aa_64x48_to_320x240": "reverse-aa",
    "reverse_aa_params_40x30_to_233x171": "reverse-aa",
    "crt_consumer_64x48_to_320x240": "crt-consumer",
    "crt_consumer_params_40x30_to_233x171_f2": "crt-consumer",
    "sameboy_lcd_64x48_to_320x240": "sameboy-lcd",
    "sameboy_lcd_params_40x30_to_233x171": "sameboy-lcd",
    "side_by_side_64x48_to_320x240": "side-by-side",
    "sbs_warp_mobile_64x36_to_320x180": "sbs-warp-mobile-16x9",
    "side_by_side_bare_params_40x30_to_233x171": "side-by-side-bare",
    "crt_lottes_64x48_to_320x240": "crt-lottes",
    "crt_lottes_params_40x30_to_233x171": "crt-lottes",
    "crt_lottes_mask0_48x36_to_200x150": "crt-lottes",
    "crt_lottes_mask2_48x36_to_200x150": "crt-lottes",
    "crt_lottes_mask4_48x36_to_200x150": "crt-lottes",
    "fakelottes_64x48_to_320x240": "fakelottes",
    "fakelottes_params_40x30_to_233x171": "fakelottes",
    "jinc2_sharper_64x48_to_320x240": "jinc2-sharper",
    "jinc2_sharper_40x30_to_233x171": "jinc2-sharper",
    "tvout_jinc_sharpen_64x48_to_320x240_f2": "tvout-jinc-sharpen",
    "tvout_interlacing_64x48_to_320x240": "tvout+interlacing",
    "interlacing_bare_40x420_to_160x420_f3": "interlacing-bare",
    "interlacing_bare_40x420_to_120x300_f2": "interlacing-bare",
    "interlacing_bare_48x36_to_200x150": "interlacing-bare",
    "tvout_64x48_to_320x240": "tvout",
    "tvout_ntsc_256px_svideo_72x40_to_300x171": "tvout+ntsc-256px-svideo",
    "retro_v2_image_adjustment_40x30_to_233x171": "retro-v2+image-adjustment",
    "tvout_tweaks_bare_params_64x48_to_256x192": "tvout-tweaks-bare",
    "image_adjustment_bare_params_64x48_to_256x192_f3": "image-adjustment-bare",
    "ntsc_gauss_scanline_96x64_to_320x240": "ntsc-256px-svideo-gauss-scanline",
    "ntsc_gauss_scanline_params_72x40_to_300x171": "ntsc-256px-svideo-gauss-scanline",
    "crt_potato_64x48_to_320x240": "crt-potato-cool",
    "crt_potato_40x30_to_233x171": "crt-potato-cool",
    "gb_palette_dmg_64x48_to_64x48": "gb-palette-dmg",
    "gb_palette_dmg_64x48_to_201x155": "gb-palette-dmg",
    "console_border_ngpc_3x_40x38_to_300x200": "ngpc-3x",
    "reshade_lut_64x48_to_160x120": "reshade-lut",
    "reshade_gba_40x30_to_97x61": "reshade-gba",
    "imgborder_gameboy_player_60x40_to_304x224": "gameboy-player",
    "imgborder_sgb_crt_geom_1x_40x36_to_256x224": "sgb-crt-geom-1x",
    "imgborder_sgb_bare_params_40x30_to_233x171": "imgborder-sgb-bare",
    "lcd_grid_64x48_to_320x240": "lcd-grid",
    "lcd_grid_params_40x30_to_233x171": "lcd-grid",
    "agb001_48x36_to_250x190": "agb001",
    "retro_v2_64x48_to_320x240": "retro-v2",
    "retro_v2_params_40x30_to_233x171": "retro-v2",
    "retro_v2_gba_color_48x36_to_240x180": "retro-v2+gba-color",
    "retro_v2_vba_color_40x30_to_233x171": "retro-v2+vba-color",
    # handheld/lcd-grid-v2.glslp and its chains (kernels/pass_lcd_grid.hip); "bare" = without the preset files' parameter block
    "lcd_grid_v2_64x48_to_320x240": "lcd-grid-v2",
    "lcd_grid_v2_40x30_to_233x171": "lcd-grid-v2",
    "lcd_grid_v2_params_48x36_to_240x180": "lcd-grid-v2",
    "lcd_grid_v2_bare_params_48x36_to_240x180": "lcd-grid-v2-bare",
    "lcd_grid_v2_bare_defaults_40x30_to_97x61": "lcd-grid-v2-bare",
    "lcd_grid_v2_gba_color_48x36_to_240x180": "lcd-grid-v2-gba-color",
    "lcd_grid_v2_gbc_color_48x36_to_200x150": "lcd-grid-v2-gbc-color",
    # handheld/<name>-color.glslp
    "gba_color_64x48_to_160x120": "gba-color",
    "gbc_color_64x48_to_160x120": "gbc-color",
    "gbc_gambatte_color_64x48_to_160x120": "gbc-gambatte-color",
    "nds_color_64x48_to_160x120": "nds-color",
    "palm_color_64x48_to_160x120": "palm-color",
    "psp_color_64x48_to_160x120": "psp-color",
    "vba_color_64x48_to_160x120": "vba-color",
    "gba_color_params_40x30_to_97x61": "gba-color",
    "gbc_color_params_40x30_to_97x61": "gbc-color",
    "vba_color_params_40x30_to_97x61": "vba-color",
    "scanline_320x240": "scanline",
    "scanline_64x48_to_160x100": "scanline",
    "crt_pi_96x64_to_192x128": "crt-pi",
    "crt_pi_80x60_to_250x190": "crt-pi",
    "ntsc_svideo_96x64_to_256x192": "ntsc-256px-svideo",        # BASELINE config 3, RGBA32F intermediate
    "ntsc_svideo_120x50_to_301x117": "ntsc-256px-svideo",
    "ntsc_256px_composite_80x48_to_200x144": "ntsc-256px",     # the rest of the ntsc family
    "ntsc_320px_composite_72x40_to_320x120": "ntsc-320px",
    "ntsc_320px_svideo_64x36_to_161x77": "ntsc-320px-svideo",
    "ntsc_3phase_linear_56x30_to_140x66": "ntsc-3phase-linear",
    "ntsc_3phase_plain_56x30_to_140x66": "ntsc-3phase-plain",
    "ntsc_2phase_linear_56x30_to_140x66": "ntsc-2phase-linear",
    "ntsc_2phase_plain_56x30_to_140x66": "ntsc-2phase-plain",
    "xbr_lv3_64x56_to_256x224": "xbr-lv3",                      # BASELINE config 5
    "xbr_lv3_48x40_to_331x217": "xbr-lv3",
    "xbr_lv3_noise_40x36_to_240x216": "xbr-lv3",
    "xbr_lv3_corner1_40x36_to_200x180": "xbr-lv3",
    "xbr_lv3_corner2_40x36_to_240x216": "xbr-lv3",
    "scalefx_48x40": "scalefx",                                 # 5 passes, two RGBA32F metric targets, PassPrev5 = original frame
    "scalefx_noise_37x29": "scalefx",
    "scalefx_params_56x44": "scalefx",
}


@pytest.mark.parametrize("case", sorted(GOLDEN_CASES))
def test_engine_matches_golden(case, preset_tree, rc_lib):
    from gpu_util import make_engine, run_engine
    g = np.load(os.path.join(GOLD, case + ".npz"))
    vw, vh = [int(v) for v in g["viewport"]]
    e = make_engine(preset_tree[GOLDEN_CASES[case]], vw, vh)
    if "param_names" in g:
        for name, v in zip(g["param_names"], g["param_values"]):
            assert e.setShaderParameter(str(name), float(v))
    for _ in range(int(g["frames"])):       # the golden run applied `frames` frames; the last one is kept
        final = run_engine(e, g["input_rgb"])
    n = int(g["n_passes"])
    for i in range(n):
        got = e.readPass(i, 0)
        ref = g["pass%d" % i]
        assert got.shape == ref.shape and got.dtype == ref.dtype
        same = got.view(np.uint32) == ref.view(np.uint32) if ref.dtype == np.float32 else got == ref
        assert same.all(), "pass %d differs: %d values" % (i, int((~same).sum()))
    assert np.array_equal(final[0], g["pass%d" % (n - 1)])
    e.shutdown()


def royale_luts():
    lut = np.load(os.path.join(GOLD, "lut_mask_slot_small_64.npy"))
    return {"mask_slot_texture_small": (lut, True, "repeat")}


ROYALE_GOLDEN = ["crt_royale_160x120_to_320x240", "crt_royale_128x96_to_400x300",
                 "crt_royale_maskon_160x120_to_320x240", "crt_royale_maskon_96x128_to_512x384",
                 "crt_royale_fake_bloom_160x120_to_320x240", "crt_royale_fake_bloom_maskon_128x96_to_400x300",
                 "crt_royale_ntsc_256px_svideo_96x64_to_320x240", "crt_royale_ntsc_320px_composite_80x56_to_300x200"]


@pytest.mark.parametrize("case", ROYALE_GOLDEN)
def test_royale_matches_oracle_and_golden(case, preset_tree, rc_lib):
    """All 12 crt-royale passes (9 of crt-royale-fake-bloom): bit-exact against the oracle run on the same
    input, and byte for byte against the llvmpipe golden vectors (every pass, sRGB8 targets included)."""
    from gpu_util import make_engine, run_engine
    from retrocapture_amd import engine as eng
    g = np.load(os.path.join(GOLD, case + ".npz"))
    vw, vh = [int(v) for v in g["viewport"]]
    maskon = "maskon" in case
    frames = int(g["frames"])
    key = ("crt-royale-fake-bloom" if "fake_bloom" in case else "crt-royale-ntsc-256px-svideo" if "ntsc_256px" in case
           else "crt-royale-ntsc-320px-composite" if "ntsc_320px" in case else "crt-royale")
    passes = eng.preset_dump(preset_tree[key])["passes"]
    n = len(passes)
    assert n == (9 if "fake_bloom" in case else 14 if "ntsc" in case else 12)
    want = run_chain(passes, g["input_rgb"], vw, vh, frame_count=frames, luts=royale_luts(), flags=1 if maskon else 0)
    e = make_engine(preset_tree[key], vw, vh)
    e.setUndefinedVaryingZero(maskon)
    for _ in range(frames):                 # the golden run applied `frames` frames; the last one is kept
        final = run_engine(e, g["input_rgb"])
    for i in range(n):
        got = e.readPass(i, 0)
        assert got.shape == want[i].shape, (i, got.shape, want[i].shape)
        assert np.array_equal(got, want[i]), "pass %d vs oracle: %d differing values" % (i, int((got != want[i]).sum()))
    # ... and every pass of the engine's own chain equals what llvmpipe rendered, byte for byte (sRGB8 passes included)
    for i in range(n):
        assert np.array_equal(e.readPass(i, 0), g["pass%d" % i]), "pass %d vs llvmpipe golden" % i
    assert np.array_equal(final[0], g["pass%d" % (n - 1)])
    e.shutdown()


ROYALE_GEOM_GOLDEN = ["crt_royale_geom_sphere_96x72_to_240x180", "crt_royale_geom_sphere_alt_tilt_96x72_to_240x180",
                      "crt_royale_geom_cylinder_96x72_to_240x180", "crt_royale_geom_flat_overscan_96x72_to_240x180",
                      "crt_royale_geom_sphere_128x96_to_401x299",
                      # crt-royale-fake-bloom: the last pass is mip-mapped, every tex2Daa tap takes its LOD from the pixel quad
                      "crt_royale_fake_bloom_geom_sphere_maskon_96x72_to_240x180", "crt_royale_fake_bloom_geom_cylinder_tilt_maskon_96x72_to_240x180",
                      "crt_royale_fake_bloom_geom_flat_overscan_maskon_96x72_to_240x180"]


@pytest.mark.parametrize("case", ROYALE_GEOM_GOLDEN)
def test_royale_curved_geometry_and_overscan_match_llvmpipe(case, preset_tree, rc_lib):
    """geom_mode_runtime 1..3 (sphere, alt. sphere, cylinder), tilt, overscan and aa_cubic_c: the last pass's tex2Daa12x /
    ray-cast form (kernels/pass_royale_last_general.hip), every pass byte for byte what llvmpipe rendered."""
    from gpu_util import make_engine, run_engine
    g = np.load(os.path.join(GOLD, case + ".npz"))
    vw, vh = [int(v) for v in g["viewport"]]
    e = make_engine(preset_tree["crt-royale-fake-bloom" if "fake_bloom" in case else "crt-royale"], vw, vh)
    e.setUndefinedVaryingZero("maskon" in case)
    for name, v in zip(g["param_names"], g["param_values"]):
        assert e.setShaderParameter(str(name), float(v))
    final = run_engine(e, g["input_rgb"])
    n = int(g["n_passes"])
    for i in range(n):
        assert np.array_equal(e.readPass(i, 0), g["pass%d" % i]), "pass %d vs llvmpipe golden" % i
    assert np.array_equal(final[0], g["pass%d" % (n - 1)])
    e.shutdown()


@pytest.mark.parametrize("params", [{"geom_mode_runtime": 1.0}, {"geom_mode_runtime": 2.0, "geom_tilt_angle_x": -0.35, "geom_tilt_angle_y": 0.3, "geom_radius": 1.1},
                                    {"geom_mode_runtime": 3.0, "geom_view_dist": 0.75, "geom_overscan_y": 1.25, "border_size": 0.1, "border_compress": 1.0},
                                    {"geom_overscan_x": 0.75, "geom_overscan_y": 1.5, "aa_cubic_c": 1.0, "lcd_gamma": 1.8}])
def test_royale_general_last_pass_floats_match_oracle(params, preset_tree, rc_lib):
    """The same form into an RGBA32F target at a larger size: the kernel's floats are the oracle's bit for bit (the oracle's
    are llvmpipe's: tests/test_oracle_golden.py, f32_crt_royale_geom_*), NaN pixels included."""
    from gpu_util import make_engine, run_engine
    from retrocapture_amd import engine as eng
    rgb = np.random.default_rng(77).integers(0, 256, (150, 200, 3), dtype=np.uint8)
    vw, vh = 517, 389
    passes = eng.preset_dump(preset_tree["crt-royale-f32-last"])["passes"]
    want = run_chain(passes, rgb, vw, vh, luts=royale_luts(), custom=params)
    e = make_engine(preset_tree["crt-royale-f32-last"], vw, vh)
    for k, v in params.items():
        assert e.setShaderParameter(k, v)
    run_engine(e, rgb)
    assert np.array_equal(e.readPass(10, 0), want[10])
    got = e.readPass(11, 0)
    assert got.dtype == np.float32 and got.shape == want[11].shape
    same = (got.view(np.uint32) == want[11].view(np.uint32)) | (np.isnan(got) & np.isnan(want[11]))
    assert same.all(), "%d of %d float components differ" % (int((~same).sum()), same.size)
    e.shutdown()


@pytest.mark.parametrize("w,h,vw,vh", [(160, 120, 320, 240), (96, 128, 517, 389), (300, 40, 300, 40)])
def test_royale_specialised_and_general_forms_agree(w, h, vw, vh, preset_tree, rc_lib):
    """Passes with a specialised form (P0: byte map of the nearest texel) must equal their general
    form byte for byte on every pass."""
    from gpu_util import make_engine, run_engine
    frames = np.random.default_rng(w + vh).integers(0, 256, (2, h, w, 3), dtype=np.uint8)
    e = make_engine(preset_tree["crt-royale"], vw, vh)
    e.setUndefinedVaryingZero(True)   # mask active: passes 7-10 carry signal
    a = run_engine(e, frames)
    pa = [e.readPass(i, 1) for i in range(12)]
    e.setGeneralKernelsOnly(True)
    b = run_engine(e, frames)
    pb = [e.readPass(i, 1) for i in range(12)]
    for i in range(12):
        assert np.array_equal(pa[i], pb[i]), "pass %d" % i
    assert np.array_equal(a, b)
    e.shutdown()


@pytest.mark.parametrize("case", ["xbr_lv2_64x56_to_256x224", "xbr_lv2_noise_40x36_to_240x216", "xbr_lv2_params_48x40_to_331x217",
                                  "xbr_lv2_details_64x56_to_256x224", "xbr_lv2_details_noise_40x36_to_240x216"])
def test_xbr_lv2_matches_oracle_and_golden(case, preset_tree, rc_lib):
    """xbr/xbr-lv2.glslp, both branches of small_details: bit-exact against the oracle and against llvmpipe (the shader reads
    an unassigned variable: what the GL makes of it is restated, oracle/rc_passes_ntsc_xbr.c)."""
    from gpu_util import make_engine, run_engine
    from retrocapture_amd import engine as eng
    g = np.load(os.path.join(GOLD, case + ".npz"))
    vw, vh = [int(v) for v in g["viewport"]]
    passes = eng.preset_dump(preset_tree["xbr-lv2"])["passes"]
    custom = dict(zip([str(x) for x in g["param_names"]], [float(v) for v in g["param_values"]])) if "param_names" in g else {}
    want = run_chain(passes, g["input_rgb"], vw, vh, frame_count=1, custom=custom)[0]
    e = make_engine(preset_tree["xbr-lv2"], vw, vh)
    for k, v in custom.items():
        assert e.setShaderParameter(k, v)
    got = run_engine(e, g["input_rgb"])[0]
    assert np.array_equal(got, want)
    e.setGeneralKernelsOnly(True)       # the run-time sampler form gives the same bytes
    assert np.array_equal(run_engine(e, g["input_rgb"])[0], got)
    e.setGeneralKernelsOnly(False)
    assert np.array_equal(got, g["pass0"])     # byte-exact against llvmpipe (line equations pinned in situ)
    e.shutdown()


@pytest.mark.parametrize("case,key", [("bayer_64x48_to_237x171", "bayer"), ("bayer_animated_80x60_to_320x240_f3", "bayer"),
                                      ("lcd1x_64x48_to_192x144", "lcd1x"), ("lcd1x_params_80x60_to_301x217", "lcd1x"),
                                      ("lcd3x_64x48_to_192x144", "lcd3x"), ("lcd3x_params_80x60_to_301x217", "lcd3x"), ("epx_80x56_to_300x200", "epx"), ("epx_mixed_64x48_to_64x48", "epx"), ("quilez_64x48_to_237x171", "quilez"), ("smootheststep_64x48_to_237x171", "smootheststep"), ("sharp_bilinear_64x48_to_237x171", "sharp-bilinear"),
                                      ("sharp_bilinear_manual_80x60_to_400x300", "sharp-bilinear"),
                                      ("crt_nes_mini_96x64_to_301x217", "crt-nes-mini"), ("crt_nes_mini_params_80x60_to_320x240", "crt-nes-mini"),
                                      ("crt_geom_96x64_to_301x217", "crt-geom"), ("crt_geom_params_80x60_to_320x240", "crt-geom"),
                                      ("crt_geom_flat_72x56_to_288x224", "crt-geom"), ("crt_geom_interlace_40x400_to_160x300_f2", "crt-geom"),
                                      ("crt_easymode_96x64_to_301x217", "crt-easymode"), ("crt_easymode_params_80x60_to_320x240", "crt-easymode"),
                                      ("zfast_crt_96x64_to_301x217", "zfast-crt"), ("zfast_crt_custom_ignored_80x60_to_320x240", "zfast-crt"),
                                      ("bilinear_64x48_to_237x171", "bilinear"), ("sharp_bilinear_2x_64x48_to_300x210", "sharp-bilinear-2x"),
                                      ("sharp_bilinear_2x_120x90_to_160x100", "sharp-bilinear-2x")])
def test_stock_presets_match_llvmpipe_golden(case, key, preset_tree, rc_lib):
    """The reference's presets made of stock.glsl alone, every pass bit-exact against llvmpipe (the second pass of
    sharp-bilinear-2x-prescale is a LINEAR copy of an RGBA8 target: llvmpipe's blit fast path)."""
    from gpu_util import make_engine, run_engine
    g = np.load(os.path.join(GOLD, case + ".npz"))
    vw, vh = [int(v) for v in g["viewport"]]
    e = make_engine(preset_tree[key], vw, vh)
    if "param_names" in g:   # zfast-crt: the reference overwrites these uniforms after the user's values (ShaderEngine.cpp:2260-2294)
        for k, v in zip(g["param_names"], g["param_values"]):
            assert e.setShaderParameter(str(k), float(v))
    for _ in range(int(g["frames"])):       # the golden run applied `frames` frames (FrameCount 1, 2, ...); the last one is kept
        final = run_engine(e, g["input_rgb"])
    n = int(g["n_passes"])
    for i in range(n):
        assert np.array_equal(e.readPass(i, 0), g["pass%d" % i]), "pass %d" % i
    assert np.array_equal(final[0], g["pass%d" % (n - 1)])
    if int(g["frames"]) == 1:
        e.setGeneralKernelsOnly(True)     # the run-time-sampler forms give the same bytes
        assert np.array_equal(run_engine(e, g["input_rgb"])[0], final[0])
    e.shutdown()


HYLLIAN_GOLDEN = ["crt_hyllian_glow_96x64_to_256x192", "crt_hyllian_glow_80x60_to_250x190", "crt_hyllian_glow_params_64x48_to_200x150"]


@pytest.mark.parametrize("case", HYLLIAN_GOLDEN)
def test_hyllian_glow_matches_oracle_and_golden(case, preset_tree, rc_lib):
    """crt/crt-hyllian-glow.glslp (the reference's smoke-test default): all 6 passes bit-exact against the oracle on
    the same input - including pass 3, which samples a mip-mapped input (chain built on the device, trilinear LOD
    from the pixel quad) - and the final RGBA8 pass against the llvmpipe golden within the sRGB-encode residual
    of the passes before it."""
    from gpu_util import make_engine, run_engine
    from retrocapture_amd import engine as eng
    g = np.load(os.path.join(GOLD, case + ".npz"))
    vw, vh = [int(v) for v in g["viewport"]]
    passes = eng.preset_dump(preset_tree["crt-hyllian-glow"])["passes"]
    custom = dict(zip([str(x) for x in g["param_names"]], [float(v) for v in g["param_values"]])) if "param_names" in g else {}
    want = run_chain(passes, g["input_rgb"], vw, vh, frame_count=1, custom=custom)
    e = make_engine(preset_tree["crt-hyllian-glow"], vw, vh)
    for k, v in custom.items():
        assert e.setShaderParameter(k, v)
    final = run_engine(e, g["input_rgb"])
    for i in range(6):
        got = e.readPass(i, 0)
        assert got.shape == want[i].shape, (i, got.shape, want[i].shape)
        assert np.array_equal(got, want[i]), "pass %d vs oracle: %d differing values" % (i, int((got != want[i]).sum()))
    # end to end against llvmpipe: byte-exact (the mip levels are drawn with the GL blitter's own quad, oracle/rc_varying.c)
    assert np.array_equal(final[0], g["pass5"])
    for i in range(6):
        assert np.array_equal(e.readPass(i, 0), g["pass%d" % i]), "pass %d vs llvmpipe" % i
    e.shutdown()


@pytest.mark.parametrize("w,h,vw,vh", [(96, 64, 256, 192), (80, 60, 250, 190), (320, 240, 1366, 768)])
def test_hyllian_glow_specialised_and_general_forms_agree(w, h, vw, vh, preset_tree, rc_lib):
    """The byte-map forms of passes 0 and 2 and the texel-LUT form of pass 1 against their general forms, every pass."""
    from gpu_util import make_engine, run_engine
    frames = np.random.default_rng(w + vw).integers(0, 256, (2, h, w, 3), dtype=np.uint8)
    e = make_engine(preset_tree["crt-hyllian-glow"], vw, vh)
    a = run_engine(e, frames)
    pa = [e.readPass(i, 1) for i in range(6)]
    e.setGeneralKernelsOnly(True)
    b = run_engine(e, frames)
    for i in range(6):
        assert np.array_equal(pa[i], e.readPass(i, 1)), "pass %d" % i
    assert np.array_equal(a, b)
    e.shutdown()


def test_image_adjustment_flip_is_refused(preset_tree, rc_lib):
    """misc/image-adjustment.glsl's ia_FLIP_* move the quad itself half off the target (clipped geometry): refused with the reason, not mis-rendered."""
    from gpu_util import make_engine, run_engine, to_device_rgba
    from retrocapture_amd.engine import RcError
    frame = np.random.default_rng(3).integers(0, 256, (1, 30, 40, 3), dtype=np.uint8)
    e = make_engine(preset_tree["image-adjustment-bare"], 120, 90)
    assert run_engine(e, frame).shape == (1, 90, 120, 4)
    assert e.setShaderParameter("ia_FLIP_HORZ", 1.0)
    with pytest.raises(RcError, match="ia_FLIP"):
        e.applyShader(to_device_rgba(frame), 40, 30)
    e.shutdown()


def test_hyllian_all_phosphor_layouts_match_llvmpipe(preset_tree, rc_lib):
    """resolve2.glsl's twenty PHOSPHOR_LAYOUT masks (the shader's mask_weights tables, layout 12's undefined row index
    included): every pass byte for byte what llvmpipe rendered."""
    from gpu_util import make_engine, run_engine
    g = np.load(os.path.join(GOLD, "crt_hyllian_glow_layouts_48x36_to_143x101.npz"))
    e = make_engine(preset_tree["crt-hyllian-glow"], 143, 101)
    assert e.setShaderParameter("MASK_INTENSITY", float(g["mask_intensity"]))
    for lay in range(20):
        assert e.setShaderParameter("PHOSPHOR_LAYOUT", float(lay))
        final = run_engine(e, g["input_rgb"])
        for i in range(5):
            assert np.array_equal(e.readPass(i, 0), g["pass%d" % i]), (lay, i)
        assert np.array_equal(final[0], g["pass5_layout%d" % lay]), "layout %d" % lay
    e.shutdown()


def test_hyllian_glow_1080p_batch(preset_tree, rc_lib):
    """Full size (1080p, 3 frames in one launch per pass; rows of the final pass spot-checked against the oracle
    through golden-free properties: frames are processed independently, so each frame of the batch equals the
    same frame run alone)."""
    from gpu_util import make_engine, run_engine, to_device_rgba
    frames = np.random.default_rng(31).integers(0, 256, (3, 270, 480, 3), dtype=np.uint8)
    e = make_engine(preset_tree["crt-hyllian-glow"], 1920, 1080)
    out = run_engine(e, frames)
    assert out.shape == (3, 1080, 1920, 4) and out[..., :3].std() > 10
    for k in range(3):
        assert np.array_equal(run_engine(e, frames[k:k + 1])[0], out[k]), k
    e2 = make_engine(preset_tree["crt-hyllian-glow"], 1920, 1080, chunk=2)     # two launches per pass: 2 + 1 frames, own mip chains
    assert np.array_equal(run_engine(e2, frames), out)
    e2.shutdown()
    e.shutdown()


def test_fake_bloom_forms_agree_and_mipmap_input_rule(preset_tree, rc_lib):
    """crt-royale-fake-bloom: specialised and general kernel forms agree on all 9 passes, at full size too (a
    1080p frame, last pass only); its last pass declares mipmap_input, which is accepted because that pass
    renders 1:1, and refused (apply returns the input, as on any failed pass) where it would need mip levels."""
    from gpu_util import make_engine, run_engine
    frames = np.random.default_rng(77).integers(0, 256, (2, 96, 128, 3), dtype=np.uint8)
    e = make_engine(preset_tree["crt-royale-fake-bloom"], 400, 300)
    e.setUndefinedVaryingZero(True)
    a = run_engine(e, frames)
    pa = [e.readPass(i, 1) for i in range(9)]
    e.setGeneralKernelsOnly(True)
    b = run_engine(e, frames)
    for i in range(9):
        assert np.array_equal(pa[i], e.readPass(i, 1)), "pass %d" % i
    assert np.array_equal(a, b) and a.std() > 10
    big = np.random.default_rng(78).integers(0, 256, (1, 1080, 1920, 3), dtype=np.uint8)
    e.setViewport(1920, 1080)
    g = run_engine(e, big)
    e.setGeneralKernelsOnly(False)
    s = run_engine(e, big)
    assert np.array_equal(g, s) and s.shape == (1, 1080, 1920, 4)
    e.shutdown()


def test_royale_interlaced_source_and_batch(preset_tree, rc_lib):
    """A 480-line source is 'interlaced' for crt-royale (288.5 < lines < 576.5): pass 0 bobs fields
    and pass 1 doubles the scanline step, both depending on FrameCount.  Batch of 3 frames."""
    from gpu_util import make_engine, run_engine
    from retrocapture_amd import engine as eng
    rng = np.random.default_rng(21)
    frames = rng.integers(0, 256, (3, 480, 96, 3), dtype=np.uint8)
    passes = eng.preset_dump(preset_tree["crt-royale"])["passes"]
    e = make_engine(preset_tree["crt-royale"], 160, 360, chunk=2)
    final = run_engine(e, frames)
    for k in range(3):
        want = run_chain(passes, frames[k], 160, 360, frame_count=k + 1, luts=royale_luts())
        assert np.array_equal(final[k], want[-1]), "frame %d" % k
    e.shutdown()


@pytest.mark.parametrize("key,w,h,vw,vh", [
    ("scanline", 33, 17, 33, 17),       # ragged: not a multiple of the 64x4 workgroup
    ("scanline", 320, 240, 320, 240),   # BASELINE config 1
    ("crt-pi", 64, 64, 64, 64),         # square target (diagonal through pixel centres)
    ("crt-pi", 100, 37, 301, 111),
    ("crt-pi", 1, 1, 5, 3),             # minimum size
    ("stock", 40, 30, 80, 60),
    ("ntsc-256px-svideo", 70, 33, 140, 99),
    ("ntsc-320px", 70, 33, 140, 99),
    ("ntsc-2phase-plain", 31, 20, 60, 41),
    ("ntsc-256px-svideo", 1, 1, 3, 2),
    ("xbr-lv3", 37, 29, 259, 203),      # 7x, ragged
    ("scalefx", 70, 41, 64, 64),         # output 3x the source whatever the viewport
    ("xbr-lv3", 2, 2, 9, 7),
    ("xbr-lv3", 64, 56, 64, 56),        # 1:1
])
def test_engine_matches_oracle(key, w, h, vw, vh, preset_tree, rc_lib):
    from gpu_util import make_engine, run_engine
    from retrocapture_amd import engine as eng
    rgb = np.random.default_rng(w * 1000 + h).integers(0, 256, (h, w, 3), dtype=np.uint8)
    passes = eng.preset_dump(preset_tree[key])["passes"]
    want = run_chain(passes, rgb, vw, vh, frame_count=1)
    e = make_engine(preset_tree[key], vw, vh)
    final = run_engine(e, rgb)
    for i, o in enumerate(want):
        got = e.readPass(i, 0)
        assert np.array_equal(got, o), "pass %d: %d differing values" % (i, int((got != o).sum()))
    assert np.array_equal(final[0], want[-1])
    e.shutdown()


HISTORY_CASES = {# the history re-draw through a pass 0 that reads its size uniforms: they are stale (pass 0's own), this repository's fixture shader
                 "history_size_48x36_to_120x90_f4": "history-size", "history_size_params_40x30_to_131x77_f9": "history-size",
                 "history_size_single_48x36_to_100x75_f3": "history-size-single",
                 "mix_frames_72x40_to_72x40_f3": "mix-frames", "mix_frames_48x36_to_120x90_f9": "mix-frames",
                 "motionblur_simple_48x36_to_120x90_f9": "motionblur-simple", "motionblur_simple_40x30_to_40x30_f3": "motionblur-simple",
                 "braid_rewind_48x36_to_120x90_f8": "braid-rewind", "response_time_48x36_to_120x90_f9": "response-time",
                 "response_time_params_40x30_to_100x75_f4": "response-time", "mix_frames_smart_48x36_to_120x90_f8": "mix-frames-smart",
                 "mix_frames_smart_params_40x30_to_40x30_f7": "mix-frames-smart",
                 "shutter_3d_48x36_to_120x90_f4": "shutter-3d", "shutter_3d_params_48x36_to_131x77_f5": "shutter-3d",
                 "anti_flicker_48x36_to_120x90_f6": "anti-flicker", "anti_flicker_params_40x30_to_40x30_f5": "anti-flicker",
                 # frame history through a pass 0 (response-time) that is not the last pass: the ring holds final outputs
                 "lcd_grid_v2_psp_color_motionblur_48x36_to_200x150_f5": "lcd-grid-v2-psp-color-motionblur",
                 "lcd_grid_v2_motionblur_48x36_to_200x150_f9": "lcd-grid-v2-motionblur",
                 "agb001_gba_color_motionblur_48x36_to_250x190_f4": "agb001-gba-color-motionblur",
                 # handheld/console-border/: a border image (LUT) laid over the frame by the last pass, which sits at pass index 3
                 "console_border_gba_lcd_grid_v2_3x_48x32_to_300x200_f4": "gba-lcd-grid-v2-3x",
                 "console_border_gbc_retro_v2_2x_40x36_to_233x171_f3": "gbc-retro-v2-2x",
                 "console_border_gba_3x_48x32_to_300x200_f9": "gba-3x",
                 "sameboy_dmg_response_time_48x36_to_48x36_f9": "sameboy-dmg-response-time",
                 "sameboy_dmg_response_time_48x36_to_131x77_f4": "sameboy-dmg-response-time",
                 "sameboy_lcd_gbc_color_motionblur_48x36_to_200x150_f4": "sameboy-lcd-gbc-color-motionblur"}


@pytest.mark.parametrize("case", sorted(HISTORY_CASES))
@pytest.mark.parametrize("as_batch", [False, True])
def test_frame_history_matches_golden(case, as_batch, preset_tree, rc_lib):
    """The five motionblur/ presets (PrevTexture .. Prev6Texture), shutter-to-side-by-side and anti-flicker: output of the last frame and the whole history ring
    (first-frame rule, recursion through pass 0's program, wrap of the 7-deep ring with its recycled-and-cleared oldest
    texture) against llvmpipe, with the frames applied one call at a time and as one batch."""
    from gpu_util import make_engine, run_engine
    g = np.load(os.path.join(GOLD, case + ".npz"))
    vw, vh = [int(v) for v in g["viewport"]]
    frames = g["input_rgb"]
    e = make_engine(preset_tree[HISTORY_CASES[case]], vw, vh)
    if "param_names" in g:
        for k, v in zip(g["param_names"], g["param_values"]):
            assert e.setShaderParameter(str(k), float(v))
    if as_batch:
        final = run_engine(e, frames)[-1]
    else:
        for f in range(frames.shape[0]):
            final = run_engine(e, frames[f])[0]
    n = int(g["n_passes"])
    for i in range(n):
        assert np.array_equal(e.readPass(i, frames.shape[0] - 1 if as_batch else 0), g["pass%d" % i]), "pass %d" % i
    assert np.array_equal(final, g["pass%d" % (n - 1)])
    assert e.historyCount() == int(g["n_history"])
    for k in range(e.historyCount()):
        assert np.array_equal(e.readHistory(k), g["history%d" % k]), "history %d" % k
    e.shutdown()


@pytest.mark.parametrize("w,h,vw,vh", [(96, 64, 256, 192), (33, 20, 100, 77), (640, 480, 640, 480)])
def test_ntsc_row_staged_and_general_forms_agree(w, h, vw, vh, preset_tree, rc_lib):
    """ntsc pass 2 stages one source row segment per wave when the host has verified the tap pattern;
    the general per-tap form must give the same bytes."""
    from gpu_util import make_engine, run_engine
    frames = np.random.default_rng(w * 3 + vh).integers(0, 256, (2, h, w, 3), dtype=np.uint8)
    e = make_engine(preset_tree["ntsc-256px-svideo"], vw, vh)
    a = run_engine(e, frames)
    e2 = make_engine(preset_tree["ntsc-256px-svideo"], vw, vh)   # a fresh engine: same FrameCount sequence
    e2.setGeneralKernelsOnly(True)
    b = run_engine(e2, frames)
    assert np.array_equal(a, b)
    e.shutdown()
    e2.shutdown()


@pytest.mark.parametrize("case", ["feedback_persist_64x40_to_64x40_f1", "feedback_persist_64x40_to_64x40_f2",
                                  "feedback_persist_64x40_to_150x90_f5"])
@pytest.mark.parametrize("as_batch", [False, True])
def test_pass_feedback_matches_golden(case, as_batch, preset_tree, rc_lib):
    """PassFeedback binding + ping-pong swap (fixture shader of this repository run on llvmpipe): the
    frame that first creates a feedback texture loses the declaring pass's draw, later frames blend
    with the previous frame's outputs of pass 0 and of the pass itself."""
    from gpu_util import make_engine, run_engine
    g = np.load(os.path.join(GOLD, case + ".npz"))
    vw, vh = [int(v) for v in g["viewport"]]
    frames = g["input_rgb"]
    e = make_engine(preset_tree["feedback-persist"], vw, vh)
    if "param_names" in g:
        for name, v in zip(g["param_names"], g["param_values"]):
            assert e.setShaderParameter(str(name), float(v))
    if as_batch:
        final = run_engine(e, frames)[-1]
        last = frames.shape[0] - 1
    else:
        for f in range(frames.shape[0]):
            final = run_engine(e, frames[f])[0]
        last = 0
    assert np.array_equal(final, g["pass1"])
    assert np.array_equal(e.readPass(0, last), g["pass0"])
    e.shutdown()


def test_pass_feedback_survives_a_viewport_change(preset_tree, rc_lib):
    """The viewport changes between frames: the passes that scale with it get new render targets, and the
    reference then deletes their feedback partners, which come back empty (and with the lost draw) when a program
    next asks (ShaderEngine.cpp:918-933, :1285-1347).  Engine against the oracle's restatement of that rule,
    frame by frame; a growing viewport would read out of bounds if the partner kept its old allocation."""
    from gpu_util import make_engine, run_engine
    from oracle_chain import ChainState
    from retrocapture_amd import engine as eng
    rng = np.random.default_rng(31)
    frames = rng.integers(0, 256, (7, 40, 64, 3), dtype=np.uint8)
    viewports = [(64, 40), (64, 40), (150, 90), (150, 90), (150, 90), (97, 61), (97, 61)]
    passes = eng.preset_dump(preset_tree["feedback-persist"])["passes"]
    e = make_engine(preset_tree["feedback-persist"], *viewports[0])
    st = ChainState()
    for f, (vw, vh) in enumerate(viewports):
        e.setViewport(vw, vh)
        final = run_engine(e, frames[f])[0]
        want = run_chain(passes, frames[f], vw, vh, frame_count=f + 1, state=st)
        assert final.shape == want[-1].shape, (f, final.shape, want[-1].shape)
        assert np.array_equal(e.readPass(0, 0), want[0]), "frame %d pass 0" % f
        assert np.array_equal(final, want[-1]), "frame %d" % f
    e.shutdown()


def test_ntsc_full_size_batch(preset_tree, rc_lib):
    """BASELINE config 3 at full size: 1920x1080 source, 1024x1080 RGBA32F intermediate, 512x1080 output;
    a batch of 3 frames (FrameCount 1..3 drives the chroma phase), every byte against the oracle."""
    from gpu_util import make_engine, run_engine
    from retrocapture_amd import engine as eng
    import oracle_lib
    oracle_lib.set_threads(8)
    try:
        rng = np.random.default_rng(31)
        frames = rng.integers(0, 256, (3, 1080, 1920, 3), dtype=np.uint8)
        passes = eng.preset_dump(preset_tree["ntsc-256px-svideo"])["passes"]
        e = make_engine(preset_tree["ntsc-256px-svideo"], 1920, 1080, chunk=2)
        final = run_engine(e, frames)
        assert final.shape == (3, 1080, 512, 4)
        for k in range(3):
            want = run_chain(passes, frames[k], 1920, 1080, frame_count=k + 1)
            assert want[0].shape == (1080, 1024, 4) and want[0].dtype == np.float32
            assert np.array_equal(final[k], want[-1]), "frame %d: %d bytes" % (k, int((final[k] != want[-1]).sum()))
        e.shutdown()
    finally:
        oracle_lib.set_threads(1)


def test_xbr_full_size_properties(preset_tree, rc_lib):
    """BASELINE config 5 at full size: 256x224 -> 3840x2160.  A 96-row band against the oracle,
    determinism, batch-slot independence, opaque alpha, and the xBR invariant that a pixel no rule
    fires on keeps its source colour (flat regions are copied)."""
    from gpu_util import make_engine, run_engine
    from oracle_lib import Tex, run_pass_rows
    import sys
    sys.path.insert(0, GOLD)
    from make_golden import pixel_art
    frame = pixel_art(256, 224, 41)
    frame[150:, 180:] = (9, 200, 77)                       # a flat block
    e = make_engine(preset_tree["xbr-lv3"], 3840, 2160, chunk=2)
    a = run_engine(e, np.stack([frame, frame[:, ::-1].copy(), frame]))
    assert a.shape == (3, 2160, 3840, 4)
    assert np.array_equal(a[0], a[2]) and (a[..., 3] == 255).all()
    assert (a[0][1500:2100, 2800:3800, :3] == np.array([9, 200, 77], np.uint8)).all()
    src = np.concatenate([frame, np.full((224, 256, 1), 255, np.uint8)], -1)
    t = Tex(src, "rgbx8", False, "clamp_to_edge")
    params = [d for _, d in chain_specs.SHADERS["xbr/shaders/xbr-lv3.glsl"]["params"]]
    rows = run_pass_rows("xbr_lv3", t, 3840, 2160, 1000, 1096, params=params)
    assert np.array_equal(a[0][1000:1096], rows)
    e.shutdown()


@pytest.mark.parametrize("w,h,vw,vh", [(64, 56, 960, 840), (37, 29, 259, 203), (50, 40, 333, 217), (31, 17, 40, 23)])
def test_xbr_general_and_per_source_pixel_forms_agree(w, h, vw, vh, preset_tree, rc_lib):
    """xbr-lv3 runs as rules-per-source-pixel + blend when the host has verified the sampling
    pattern; the general one-kernel form must give the same bytes, and both equal the oracle."""
    from gpu_util import make_engine, run_engine
    from retrocapture_amd import engine as eng
    import sys
    sys.path.insert(0, GOLD)
    from make_golden import pixel_art
    frames = np.stack([pixel_art(w, h, 50 + w), np.random.default_rng(w).integers(0, 256, (h, w, 3), dtype=np.uint8)])
    e = make_engine(preset_tree["xbr-lv3"], vw, vh)
    fast = run_engine(e, frames)
    e.setGeneralKernelsOnly(True)
    general = run_engine(e, frames)
    assert np.array_equal(fast, general)
    passes = eng.preset_dump(preset_tree["xbr-lv3"])["passes"]
    for k in range(2):
        want = run_chain(passes, frames[k], vw, vh)
        assert np.array_equal(general[k], want[-1]), "frame %d" % k
    e.shutdown()


def test_batch_equals_single_frames(preset_tree, rc_lib):
    """N frames in one call == N successive applyShader calls (FrameCount advances per frame)."""
    from gpu_util import make_engine, run_engine
    rng = np.random.default_rng(7)
    frames = rng.integers(0, 256, (5, 48, 64, 3), dtype=np.uint8)
    e1 = make_engine(preset_tree["crt-pi"], 128, 96, chunk=2)
    batch = run_engine(e1, frames)
    e2 = make_engine(preset_tree["crt-pi"], 128, 96)
    for k in range(5):
        one = run_engine(e2, frames[k])
        assert np.array_equal(one[0], batch[k]), k
    e1.shutdown()
    e2.shutdown()


def test_parameters_roundtrip(preset_tree, rc_lib):
    from gpu_util import make_engine, run_engine
    from retrocapture_amd import engine as eng
    e = make_engine(preset_tree["crt-pi"], 96, 64)
    names = [p.name for p in e.getShaderParameters()]
    assert names == sorted(names) and "MASK_BRIGHTNESS" in names and len(names) == 8
    assert e.setShaderParameter("MASK_BRIGHTNESS", 5.0)          # clamped to max 1.0
    assert not e.setShaderParameter("NOPE", 1.0)
    p = {q.name: q for q in e.getShaderParameters()}["MASK_BRIGHTNESS"]
    assert p.value == 1.0 and abs(p.defaultValue - 0.7) < 1e-6
    assert e.setShaderParameter("SCANLINE_WEIGHT", 3.5)
    rgb = np.random.default_rng(1).integers(0, 256, (64, 96, 3), dtype=np.uint8)
    final = run_engine(e, rgb)
    passes = eng.preset_dump(preset_tree["crt-pi"])["passes"]
    want = run_chain(passes, rgb, 96, 64, custom={"MASK_BRIGHTNESS": 1.0, "SCANLINE_WEIGHT": 3.5})
    assert np.array_equal(final[0], want[-1])
    e.shutdown()


def test_inactive_engine_returns_input(rc_lib):
    import torch
    from retrocapture_amd import ShaderEngine
    e = ShaderEngine()
    assert e.init(0)
    x = torch.zeros((8, 8, 4), dtype=torch.uint8, device="cuda")
    ptr, w, h = e.applyShader(x, 8, 8)           # no preset loaded: reference returns the input
    assert ptr == x.data_ptr() and (w, h) == (8, 8)
    e.shutdown()


@pytest.mark.parametrize("src,vp,params", [((1080, 1920), (1920, 1080), {}), ((240, 320), (1920, 1080), {}), ((480, 640), (1003, 701), {}),
                                           ((1080, 1920), (640, 360), {}),        # minification: source rows are skipped
                                           ((224, 256), (1280, 960), {"INPUT_GAMMA": 1.8, "OUTPUT_GAMMA": 2.6, "BLOOM_FACTOR": 3.0, "MASK_BRIGHTNESS": 0.35}),
                                           ((224, 256), (800, 600), {"SCANLINE_GAP_BRIGHTNESS": 0.0, "SCANLINE_WEIGHT": 15.0}),
                                           # widths that leave 16 / 48 columns to the last wave: lanes beyond the right edge sit strips out,
                                           # and the per-wave list of uncertain pixels must not depend on them (round 4: it did)
                                           ((224, 256), (1040, 780), {}),
                                           ((270, 480), (1968, 1107), {})])
def test_crt_pi_table_form_equals_exact_form(src, vp, params, preset_tree, rc_lib):
    """crt-pi's strip kernel takes its two gamma pows from tables with measured bounds and sends what it cannot certify to the exact
    per-pixel form (kernels/pass_crt_pi.hip): every byte must equal the exact form's (rc_engine_set_general_kernels_only) - on noise,
    on smooth ramps (colours between the byte values), on black / white edges (thin lerps against 0: the tables' low end) - and a
    band of it the oracle's."""
    from gpu_util import make_engine, run_engine
    h, w = src
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    ramp = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) % 256)], -1).astype(np.uint8)
    bars = np.where(((xx // 3 + yy // 5) % 2)[..., None] == 0, 0, np.array([255, 1, 254])).astype(np.uint8)
    frames = np.stack([noise, ramp, bars])
    e = make_engine(preset_tree["crt-pi"], vp[0], vp[1])
    for k, v in params.items():
        assert e.setShaderParameter(k, v)
    fast = run_engine(e, frames).copy()
    e.setGeneralKernelsOnly(True)
    exact = run_engine(e, frames)
    assert np.array_equal(fast, exact), "%d differing bytes" % int((fast != exact).sum())
    e.shutdown()
    from oracle_lib import Tex, run_pass_rows
    t = Tex(np.concatenate([noise, np.full((h, w, 1), 255, np.uint8)], -1), "rgbx8", True, "clamp_to_border")
    vals = [params.get(n, d) for n, d in chain_specs.SHADERS["crt/shaders/crt-pi.glsl"]["params"]]
    y0 = vp[1] // 3
    rows = run_pass_rows("crt_pi", t, vp[0], vp[1], y0, y0 + 24, params=vals)
    assert np.array_equal(fast[0][y0:y0 + 24], rows)


def test_full_size_properties(preset_tree, rc_lib):
    """BASELINE config 2 at full size (1920x1080 crt-pi), checked through properties that do
    not need the slow oracle: determinism, frame independence inside a batch, agreement of a
    64-row band with the oracle, alpha = 255 everywhere."""
    from gpu_util import make_engine, run_engine
    from retrocapture_amd import engine as eng
    rng = np.random.default_rng(11)
    frame = rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
    e = make_engine(preset_tree["crt-pi"], 1920, 1080, chunk=2)
    a = run_engine(e, np.stack([frame, frame[::-1].copy(), frame]))
    assert np.array_equal(a[0], a[2])                      # same input, different batch slot
    assert (a[..., 3] == 255).all()
    b = run_engine(e, frame)
    assert np.array_equal(a[0], b[0])                      # deterministic across calls
    # oracle on a full-width band: crt-pi is row-local (taps within +-1 source row), so the
    # oracle run on the whole frame is affordable only for a few rows; use the C oracle's
    # row range instead of cropping (cropping would change texture coordinates)
    from oracle_lib import Tex, run_pass_rows
    src = np.concatenate([frame, np.full((1080, 1920, 1), 255, np.uint8)], -1)
    t = Tex(src, "rgbx8", True, "clamp_to_border")
    params = [d for _, d in chain_specs.SHADERS["crt/shaders/crt-pi.glsl"]["params"]]
    rows = run_pass_rows("crt_pi", t, 1920, 1080, 500, 564, params=params)
    assert np.array_equal(a[0][500:564], rows)
    e.shutdown()


@pytest.mark.parametrize("key", ["zfast-crt", "crt-hyllian-glow", "crt-royale", "crt-royale-fake-bloom", "crt-pi", "scanline", "ntsc-256px-svideo", "ntsc-320px", "xbr-lv3"])
def test_smoke_statistics_like_the_reference(key, preset_tree, rc_lib):
    """The reference's only end-to-end check (tools/smoke-test.sh:221-300) restated: on its synthetic
    colour-bar source (VideoCaptureTestPattern.cpp:65-101) the shaded frame is not black, has variance,
    keeps chroma, differs from the raw frame by a mean of at least 5 levels for the CRT presets, and two
    consecutive frames (the marker moves) differ."""
    from gpu_util import make_engine, run_engine
    import sys
    sys.path.insert(0, GOLD)
    from make_golden import bars
    w, h = 320, 240
    f0, f1 = bars(w, h, 0), bars(w, h, 40)
    e = make_engine(preset_tree[key], w, h)
    if key.startswith("crt-royale"):
        e.setUndefinedVaryingZero(True)
    out = run_engine(e, np.stack([f0, f1]))
    a, b = out[0][..., :3].astype(np.float64), out[1][..., :3].astype(np.float64)
    assert a.mean() > 8 and a.std() > 10, (a.mean(), a.std())
    assert np.abs(a[..., 0] - a[..., 2]).mean() > 3              # chroma present
    assert not np.array_equal(out[0], out[1])                    # the moving marker shows
    if key.startswith("crt") or key == "scanline":
        oh, ow = a.shape[:2]
        ys, xs = (np.arange(oh) * h // oh), (np.arange(ow) * w // ow)
        raw = f0[ys][:, xs].astype(np.float64)
        assert np.abs(a - raw).mean() >= 5
    e.shutdown()


def test_parameters_set_from_another_thread(preset_tree, rc_lib):
    """The reference shares its parameter maps between the HTTP thread and the GL thread unguarded
    (APIController.cpp:1774 vs ShaderEngine.cpp:2234); the library locks them: hammering setShaderParameter
    from a second thread while frames are applied must neither crash nor tear a value."""
    import threading
    from gpu_util import make_engine, run_engine
    e = make_engine(preset_tree["crt-pi"], 96, 64)
    rgb = np.random.default_rng(5).integers(0, 256, (64, 96, 3), dtype=np.uint8)
    stop = threading.Event()

    def hammer():
        k = 0
        while not stop.is_set():
            e.setShaderParameter("MASK_BRIGHTNESS", 0.5 if k & 1 else 0.9)
            k += 1

    t = threading.Thread(target=hammer)
    t.start()
    try:
        outs = [run_engine(e, rgb)[0] for _ in range(30)]
    finally:
        stop.set()
        t.join()
    e.setShaderParameter("MASK_BRIGHTNESS", 0.5)
    lo = run_engine(e, rgb)[0]
    e.setShaderParameter("MASK_BRIGHTNESS", 0.9)
    hi = run_engine(e, rgb)[0]
    for o in outs:                       # every frame used one of the two values, never a mixture
        assert np.array_equal(o, lo) or np.array_equal(o, hi)
    e.shutdown()


_MATRIX_SHADERS = {"stock": "stock.glsl", "scanline": "scanlines/shaders/scanline.glsl", "crt-pi": "crt/shaders/crt-pi.glsl",
                   "xbr": "xbr/shaders/xbr-lv3.glsl"}


@pytest.mark.parametrize("shader", sorted(_MATRIX_SHADERS))
@pytest.mark.parametrize("wrap", ["clamp_to_edge", "clamp_to_border", "repeat", "mirrored_repeat"])
def test_sampler_state_and_target_format_matrix(shader, wrap, tmp_path, rc_lib):
    """Every sampler state x target format reaches a kernel's general (run-time selected) form: a second
    pass (stock) reads the first pass's target with filter / wrap under test, and the first pass's target
    is RGBA8, sRGB8 or RGBA32F.  GPU == oracle for all of them."""
    from gpu_util import make_engine, run_engine
    from retrocapture_amd import engine as eng
    root = tmp_path / "shaders_glsl"
    root.mkdir()
    rng = np.random.default_rng(len(shader) * 7 + len(wrap))
    rgb = rng.integers(0, 256, (23, 31, 3), dtype=np.uint8)
    for lin in ("false", "true"):
        for fb in ("", "srgb_framebuffer0 = true\n", "float_framebuffer0 = true\n"):
            text = ("shaders = 2\nshader0 = %s\nfilter_linear0 = %s\nwrap_mode0 = %s\nscale_type0 = source\nscale0 = 2.0\n%s"
                    "shader1 = stock.glsl\nfilter_linear1 = %s\nwrap_mode1 = %s\n"
                    % (_MATRIX_SHADERS[shader], lin, wrap, fb, lin, wrap))
            p = root / ("m_%s_%s_%s.glslp" % (lin, wrap, "rgba8" if not fb else fb.split("_")[0]))
            p.write_text(text)
            passes = eng.preset_dump(str(p))["passes"]
            want = run_chain(passes, rgb, 77, 51)
            e = make_engine(str(p), 77, 51)
            final = run_engine(e, rgb)
            for i, o in enumerate(want):
                got = e.readPass(i, 0)
                same = got.view(np.uint32) == o.view(np.uint32) if o.dtype == np.float32 else got == o
                assert same.all(), "%s pass %d: %d values differ" % (p.name, i, int((~same).sum()))
            assert np.array_equal(final[0], want[-1])
            e.shutdown()


@pytest.mark.parametrize("wrap", ["clamp_to_edge", "clamp_to_border", "repeat", "mirrored_repeat"])
@pytest.mark.parametrize("tag", ["rgba8", "srgb8"])
def test_wrap_modes_match_llvmpipe_golden(wrap, tag, tmp_path, rc_lib):
    """All four wrap modes with LINEAR at chain level against llvmpipe (crt-pi then stock, both with the
    wrap mode under test); stock on a plain RGBA8 target with clamp-to-edge goes through llvmpipe's blit
    fast path, restated in k_stock_blit."""
    from gpu_util import make_engine, run_engine
    from test_oracle_golden import wrap_case_preset
    case = "wrap_%s_%s_40x30_to_97x71" % (wrap, tag)
    g = np.load(os.path.join(GOLD, case + ".npz"))
    e = make_engine(wrap_case_preset(tmp_path, case), 97, 71)
    final = run_engine(e, g["input_rgb"])
    p0 = e.readPass(0, 0)
    d0 = np.abs(p0.astype(np.int32) - g["pass0"].astype(np.int32))
    d1 = np.abs(final[0].astype(np.int32) - g["pass1"].astype(np.int32))
    if tag == "rgba8":
        assert d0.max() == 0
        assert d1.max() == 0      # clamp_to_edge: llvmpipe's blit fast path, restated
    else:
        assert d0.max() == 0      # sRGB8 target: llvmpipe's own encode, restated exactly
        assert d1.max() == 0
    e.shutdown()


@pytest.mark.parametrize("case", ["blit_nearest_60x45_to_540x405", "blit_linear_60x45_to_540x405", "blit_linear_160x120_to_233x150"])
def test_blit_fast_path_matches_llvmpipe(case, tmp_path, rc_lib):
    from gpu_util import make_engine, run_engine
    from test_oracle_golden import blit_case_preset, BLIT_CASES
    linear, vw, vh = BLIT_CASES[case]
    g = np.load(os.path.join(GOLD, case + ".npz"))
    e = make_engine(blit_case_preset(tmp_path, linear), vw, vh)
    final = run_engine(e, g["input_rgb"])
    assert np.array_equal(e.readPass(0, 0), g["pass0"])
    assert np.array_equal(final[0], g["pass1"])
    e.shutdown()


@pytest.mark.parametrize("params", [{}, {"x_tilt": -0.3, "y_tilt": 0.25, "R": 1.5, "d": 1.2, "SATURATION": 0.7, "lum": 0.2},
                                    {"x_tilt": 0.5, "y_tilt": -0.5, "R": 0.6, "d": 0.4, "overscan_x": 90.0, "cornersize": 0.2, "SHARPER": 3.0}])
def test_crt_geom_matches_oracle_at_size(params, preset_tree, rc_lib):
    """crt/crt-geom.glslp at 320x240 -> 1280x960 against the oracle (itself bit-identical to llvmpipe at float precision),
    strong tilts included: there the viewing ray misses the tube near the corners and the geometry is NaN (pow(NaN) = 0,
    texel 0), which must come out the same."""
    from gpu_util import make_engine, run_engine
    from retrocapture_amd import engine as eng
    rgb = np.random.default_rng(77).integers(0, 256, (240, 320, 3), dtype=np.uint8)
    passes = eng.preset_dump(preset_tree["crt-geom"])["passes"]
    want = run_chain(passes, rgb, 1280, 960, frame_count=1, custom=params)
    e = make_engine(preset_tree["crt-geom"], 1280, 960)
    for k, v in params.items():
        assert e.setShaderParameter(k, v)
    got = run_engine(e, rgb)[0]
    assert np.array_equal(got, want[0]), "%d differing bytes" % int((got != want[0]).sum())
    e.shutdown()


@pytest.mark.parametrize("case,key", [("mip_source_96x64_s0.4", "mip-source-0.4"), ("mip_source_125x95_s0.23", "mip-source-0.23"),
                                      ("mip_rgba8_96x64_s0.37", "mip-rgba8-0.37"), ("mip_rgba8_101x67_s0.6", "mip-rgba8-0.6"),
                                      # mipmap_input without filter_linear: GL_NEAREST_MIPMAP_NEAREST, one level per quad
                                      ("mipnearest_source_96x64_s0.4", "mipnearest-source-0.4"), ("mipnearest_source_125x95_s0.17", "mipnearest-source-0.17"),
                                      ("mipnearest_rgba8_101x67_s0.6", "mipnearest-rgba8-0.6")])
def test_mipmap_input_on_8bit_textures_matches_llvmpipe(case, key, preset_tree, rc_lib):
    """mipmap_input on the GL_RGB source frame and on a plain RGBA8 render target: llvmpipe builds those levels with the
    ordinary 8-bit LINEAR filter (not its blit fast path) and blends the two level samples in 8 bits
    (weight floor(frac(lod) * 256)); nine trilinear taps per pixel at fractional LODs, every pass byte-exact."""
    from gpu_util import make_engine, run_engine
    g = np.load(os.path.join(GOLD, case + ".npz"))
    vw, vh = [int(v) for v in g["viewport"]]
    e = make_engine(preset_tree[key], vw, vh)
    final = run_engine(e, g["input_rgb"])
    n = int(g["n_passes"])
    for i in range(n):
        assert np.array_equal(e.readPass(i, 0), g["pass%d" % i]), "pass %d" % i
    assert np.array_equal(final[0], g["pass%d" % (n - 1)])
    batch = np.stack([g["input_rgb"]] * 3)          # the chain is built per frame of a batch
    out = run_engine(e, batch)
    assert all(np.array_equal(out[k], final[0]) for k in range(3))
    e.shutdown()
