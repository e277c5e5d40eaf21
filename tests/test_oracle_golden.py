"""Oracle (CPU restatement) against the golden vectors produced by running the reference's GLSL
on Mesa llvmpipe (tests/golden/make_golden.py).  Pins the oracle; no GPU involved."""
import glob
import os

import numpy as np
import pytest

import chain_specs
from oracle_chain import run_chain

GOLD = os.path.join(os.path.dirname(__file__), "golden")

# golden file prefix -> preset text key
CASES = {
    "scalefx_48x40": "scalefx",
    "scalefx_noise_37x29": "scalefx",
    "scalefx_params_56x44": "scalefx",            # SFX_CLR 0.35, SFX_SAA 0, SFX_SCN 0
    "mip_source_96x64_s0.4": "mip-source-0.4",                  # mipmap_input0: the GL_RGB source frame, 8-bit levels and blend
    "mip_source_125x95_s0.23": "mip-source-0.23",
    "mip_rgba8_96x64_s0.37": "mip-rgba8-0.37",                  # mipmap_input on a plain RGBA8 render target
    "mip_rgba8_101x67_s0.6": "mip-rgba8-0.6",
    "mipnearest_source_96x64_s0.4": "mipnearest-source-0.4",    # mipmap_input without filter_linear: GL_NEAREST_MIPMAP_NEAREST
    "mipnearest_source_125x95_s0.17": "mipnearest-source-0.17",
    "mipnearest_rgba8_101x67_s0.6": "mipnearest-rgba8-0.6",
    "crt_geom_96x64_to_301x217": "crt-geom",
    "crt_geom_params_80x60_to_320x240": "crt-geom",            # tilt, overscan, corner, SHARPER 2, saturation ...
    "crt_geom_flat_72x56_to_288x224": "crt-geom",              # CURVATURE 0
    "crt_geom_interlace_40x400_to_160x300_f2": "crt-geom",     # >= 400 source lines: FrameCount parity shifts the field
    "scanline_320x240": "scanline",
    "scanline_64x48_to_160x100": "scanline",
    "crt_pi_96x64_to_192x128": "crt-pi",
    "crt_pi_80x60_to_250x190": "crt-pi",
    "crt_royale_160x120_to_320x240": "crt-royale",
    "crt_royale_128x96_to_400x300": "crt-royale",
    # the last pass's tex2Daa12x / ray-cast branch (geom_mode_runtime 1..3, overscan != 1; rc_passes_royale_last.c)
    "crt_royale_geom_sphere_96x72_to_240x180": "crt-royale",
    "crt_royale_geom_sphere_alt_tilt_96x72_to_240x180": "crt-royale",
    "crt_royale_geom_cylinder_96x72_to_240x180": "crt-royale",
    "crt_royale_geom_flat_overscan_96x72_to_240x180": "crt-royale",
    "crt_royale_geom_sphere_128x96_to_401x299": "crt-royale",
    # ... on crt-royale-fake-bloom, whose last pass is mip-mapped: per-tap LOD from the quad's top-left differences
    "crt_royale_fake_bloom_geom_sphere_maskon_96x72_to_240x180": "crt-royale-fake-bloom",
    "crt_royale_fake_bloom_geom_cylinder_tilt_maskon_96x72_to_240x180": "crt-royale-fake-bloom",
    "crt_royale_fake_bloom_geom_flat_overscan_maskon_96x72_to_240x180": "crt-royale-fake-bloom",
    "history_size_48x36_to_120x90_f4": "history-size",           # history re-draw with pass 0's stale size uniforms (fixture shader)
    "history_size_params_40x30_to_131x77_f9": "history-size",
    "history_size_single_48x36_to_100x75_f3": "history-size-single",
    "feedback_persist_64x40_to_64x40_f1": "feedback-persist",
    "feedback_persist_64x40_to_64x40_f2": "feedback-persist",
    "feedback_persist_64x40_to_150x90_f5": "feedback-persist",
    "motionblur_simple_48x36_to_120x90_f9": "motionblur-simple",   # Prev .. Prev6; 9 frames: the full ring recycles (and clears) its oldest texture
    "motionblur_simple_40x30_to_40x30_f3": "motionblur-simple",
    "braid_rewind_48x36_to_120x90_f8": "braid-rewind",
    # handheld/lcd-grid-v2.glslp and its chains (rc_passes_lcd.c); "bare" = the shader without the preset files' parameter block
    "lcd_grid_v2_64x48_to_320x240": "lcd-grid-v2",
    "lcd_grid_v2_40x30_to_233x171": "lcd-grid-v2",
    "lcd_grid_v2_params_48x36_to_240x180": "lcd-grid-v2",        # custom values of the 14 names the preset file sets are overridden by it; outgamma moves
    "lcd_grid_v2_bare_params_48x36_to_240x180": "lcd-grid-v2-bare",
    "lcd_grid_v2_bare_defaults_40x30_to_97x61": "lcd-grid-v2-bare",
    "lcd_grid_v2_gba_color_48x36_to_240x180": "lcd-grid-v2-gba-color",
    "lcd_grid_v2_gbc_color_48x36_to_200x150": "lcd-grid-v2-gbc-color",
    "lcd_grid_v2_psp_color_motionblur_48x36_to_200x150_f5": "lcd-grid-v2-psp-color-motionblur",   # frame history through a pass 0 that is not the last pass
    "lcd_grid_v2_motionblur_48x36_to_200x150_f9": "lcd-grid-v2-motionblur",
    "console_border_gba_lcd_grid_v2_3x_48x32_to_300x200_f4": "gba-lcd-grid-v2-3x",   # handheld/console-border/: border overlay (gb-pass-5) behind a history chain
    "console_border_gbc_retro_v2_2x_40x36_to_233x171_f3": "gbc-retro-v2-2x",
    "imgborder_gameboy_player_60x40_to_304x224": "gameboy-player",     # borders/: the frame inside a border image
    "imgborder_sgb_crt_geom_1x_40x36_to_256x224": "sgb-crt-geom-1x",
    "imgborder_sgb_bare_params_40x30_to_233x171": "imgborder-sgb-bare",
    "console_border_ngpc_3x_40x38_to_300x200": "ngpc-3x",
    "sameboy_lcd_64x48_to_320x240": "sameboy-lcd",
    "sameboy_lcd_params_40x30_to_233x171": "sameboy-lcd",
    "sameboy_lcd_gbc_color_motionblur_48x36_to_200x150_f4": "sameboy-lcd-gbc-color-motionblur",
    "side_by_side_64x48_to_320x240": "side-by-side",
    "sbs_warp_mobile_64x36_to_320x180": "sbs-warp-mobile-16x9",
    "side_by_side_bare_params_40x30_to_233x171": "side-by-side-bare",
    "advanced_aa_64x48_to_320x240": "advanced-aa",
    "advanced_aa_params_40x30_to_233x171": "advanced-aa",
    "reverse_aa_64x48_to_320x240": "reverse-aa",
    "reverse_aa_params_40x30_to_233x171": "reverse-aa",
    "crt_consumer_64x48_to_320x240": "crt-consumer",
    "crt_consumer_params_40x30_to_233x171_f2": "crt-consumer",     # masks, slot mask, noise seeded by FrameCount, vignette, warp, corner
    "crt_lottes_64x48_to_320x240": "crt-lottes",
    "crt_lottes_params_40x30_to_233x171": "crt-lottes",
    "crt_lottes_mask0_48x36_to_200x150": "crt-lottes",
    "crt_lottes_mask2_48x36_to_200x150": "crt-lottes",
    "crt_lottes_mask4_48x36_to_200x150": "crt-lottes",
    "fakelottes_64x48_to_320x240": "fakelottes",
    "fakelottes_params_40x30_to_233x171": "fakelottes",
    "jinc2_sharper_64x48_to_320x240": "jinc2-sharper",
    "jinc2_sharper_40x30_to_233x171": "jinc2-sharper",
    "tvout_jinc_sharpen_64x48_to_320x240_f2": "tvout-jinc-sharpen",          # 4 passes: tvout, image-adjustment, jinc2-sharper, interlacing (pass index 3)
    "tvout_interlacing_64x48_to_320x240": "tvout+interlacing",
    "interlacing_bare_40x420_to_160x420_f3": "interlacing-bare",    # > 400 source lines: the field alternates with FrameCount
    "interlacing_bare_40x420_to_120x300_f2": "interlacing-bare",
    "interlacing_bare_48x36_to_200x150": "interlacing-bare",
    "tvout_64x48_to_320x240": "tvout",                                               # crt/shaders/tvout-tweaks + misc/image-adjustment (instruction lists)
    "tvout_ntsc_256px_svideo_72x40_to_300x171": "tvout+ntsc-256px-svideo",         # image-adjustment at pass index 3 (TextureSize.y rule)
    "retro_v2_image_adjustment_40x30_to_233x171": "retro-v2+image-adjustment",
    "tvout_tweaks_bare_params_64x48_to_256x192": "tvout-tweaks-bare",
    "image_adjustment_bare_params_64x48_to_256x192_f3": "image-adjustment-bare",   # film grain seeded by FrameCount (3rd frame), zoom, shift, masks, sharpen
    "ntsc_gauss_scanline_96x64_to_320x240": "ntsc-256px-svideo-gauss-scanline",
    "ntsc_gauss_scanline_params_72x40_to_300x171": "ntsc-256px-svideo-gauss-scanline",
    "crt_potato_64x48_to_320x240": "crt-potato-cool",
    "crt_potato_40x30_to_233x171": "crt-potato-cool",
    "sameboy_dmg_response_time_48x36_to_48x36_f9": "sameboy-dmg-response-time",   # history (full ring) in front of a palette lookup
    "sameboy_dmg_response_time_48x36_to_131x77_f4": "sameboy-dmg-response-time",
    "gb_palette_dmg_64x48_to_64x48": "gb-palette-dmg",
    "gb_palette_dmg_64x48_to_201x155": "gb-palette-dmg",
    "reshade_lut_64x48_to_160x120": "reshade-lut",
    "reshade_gba_40x30_to_97x61": "reshade-gba",
    "lcd_grid_64x48_to_320x240": "lcd-grid",
    "lcd_grid_params_40x30_to_233x171": "lcd-grid",
    "console_border_gba_3x_48x32_to_300x200_f9": "gba-3x",      # motionblur-simple (Prev .. Prev6, full ring) in front of a 4-pass chain with a border LUT
    "agb001_48x36_to_250x190": "agb001",
    "agb001_gba_color_motionblur_48x36_to_250x190_f4": "agb001-gba-color-motionblur",
    "retro_v2_64x48_to_320x240": "retro-v2",
    "retro_v2_params_40x30_to_233x171": "retro-v2",
    "retro_v2_gba_color_48x36_to_240x180": "retro-v2+gba-color",
    "retro_v2_vba_color_40x30_to_233x171": "retro-v2+vba-color",
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
    "shutter_3d_48x36_to_120x90_f4": "shutter-3d",               # FrameCount parity selects the eye; PrevTexture held x flicker
    "shutter_3d_params_48x36_to_131x77_f5": "shutter-3d",        # all seven parameters changed
    "anti_flicker_48x36_to_120x90_f6": "anti-flicker",
    "anti_flicker_params_40x30_to_40x30_f5": "anti-flicker",
    "response_time_48x36_to_120x90_f9": "response-time",
    "response_time_params_40x30_to_100x75_f4": "response-time",
    "mix_frames_smart_48x36_to_120x90_f8": "mix-frames-smart",
    "mix_frames_smart_params_40x30_to_40x30_f7": "mix-frames-smart",
    "mix_frames_72x40_to_72x40_f3": "mix-frames",
    "mix_frames_48x36_to_120x90_f9": "mix-frames",
    "ntsc_256px_composite_80x48_to_200x144": "ntsc-256px",
    "ntsc_320px_composite_72x40_to_320x120": "ntsc-320px",
    "ntsc_320px_svideo_64x36_to_161x77": "ntsc-320px-svideo",
    "ntsc_3phase_linear_56x30_to_140x66": "ntsc-3phase-linear",
    "ntsc_3phase_plain_56x30_to_140x66": "ntsc-3phase-plain",
    "ntsc_2phase_linear_56x30_to_140x66": "ntsc-2phase-linear",
    "ntsc_2phase_plain_56x30_to_140x66": "ntsc-2phase-plain",
    "ntsc_svideo_96x64_to_256x192": "ntsc-256px-svideo",
    "ntsc_svideo_120x50_to_301x117": "ntsc-256px-svideo",
    "xbr_lv3_64x56_to_256x224": "xbr-lv3",
    "xbr_lv3_48x40_to_331x217": "xbr-lv3",
    "xbr_lv3_noise_40x36_to_240x216": "xbr-lv3",
    "xbr_lv3_corner1_40x36_to_200x180": "xbr-lv3",
    "xbr_lv3_corner2_40x36_to_240x216": "xbr-lv3",
    # 14 passes: two ntsc passes, then crt-royale with its scanlines-vertical pass at pass index 3, where the
    # reference overrides TextureSize.y (ShaderEngine.cpp:2418-2421), and a mip-mapped last pass
    "crt_royale_ntsc_256px_svideo_96x64_to_320x240": "crt-royale-ntsc-256px-svideo",
    "crt_royale_ntsc_320px_composite_80x56_to_300x200": "crt-royale-ntsc-320px-composite",
    "bayer_64x48_to_237x171": "bayer",
    "bayer_animated_80x60_to_320x240_f3": "bayer",     # animate = 1: the pattern scale follows FrameCount (3 frames applied)
    "lcd1x_64x48_to_192x144": "lcd1x",
    "lcd1x_params_80x60_to_301x217": "lcd1x",
    "lcd3x_64x48_to_192x144": "lcd3x",
    "lcd3x_params_80x60_to_301x217": "lcd3x",
    "epx_80x56_to_300x200": "epx",          # the only pass is source x 2.0: 160x112 whatever the viewport
    "epx_mixed_64x48_to_64x48": "epx",
    "quilez_64x48_to_237x171": "quilez",
    "smootheststep_64x48_to_237x171": "smootheststep",
    "sharp_bilinear_64x48_to_237x171": "sharp-bilinear",
    "sharp_bilinear_manual_80x60_to_400x300": "sharp-bilinear",
    "crt_nes_mini_96x64_to_301x217": "crt-nes-mini",
    "crt_nes_mini_params_80x60_to_320x240": "crt-nes-mini",   # BRIGHTBOOST set by the user: overwritten with 1.25 by the reference
    "crt_easymode_96x64_to_301x217": "crt-easymode",
    "crt_easymode_params_80x60_to_320x240": "crt-easymode",
    "zfast_crt_96x64_to_301x217": "zfast-crt",
    "zfast_crt_custom_ignored_80x60_to_320x240": "zfast-crt",   # custom BLURSCALEX / MASK_DARK: overwritten by the reference
    "bilinear_64x48_to_237x171": "bilinear",
    "sharp_bilinear_2x_64x48_to_300x210": "sharp-bilinear-2x",
    "sharp_bilinear_2x_120x90_to_160x100": "sharp-bilinear-2x",
    "xbr_lv2_64x56_to_256x224": "xbr-lv2",
    "xbr_lv2_noise_40x36_to_240x216": "xbr-lv2",
    "xbr_lv2_params_48x40_to_331x217": "xbr-lv2",
    "xbr_lv2_details_64x56_to_256x224": "xbr-lv2",              # small_details = 1
    "xbr_lv2_details_noise_40x36_to_240x216": "xbr-lv2",        # small_details = 1, XBR_Y_WEIGHT 60
    "crt_hyllian_glow_96x64_to_256x192": "crt-hyllian-glow",
    "crt_hyllian_glow_80x60_to_250x190": "crt-hyllian-glow",         # pass 3 is 63x48: not viewport / 4, fractional mip LOD
    "crt_hyllian_glow_params_64x48_to_200x150": "crt-hyllian-glow",
    "crt_royale_fake_bloom_160x120_to_320x240": "crt-royale-fake-bloom",
    "crt_royale_fake_bloom_maskon_128x96_to_400x300": "crt-royale-fake-bloom",
    # the same preset as a GL that reads 0 from pass 6's unwritten varying renders it
    "crt_royale_maskon_160x120_to_320x240": "crt-royale",
    "crt_royale_maskon_96x128_to_512x384": "crt-royale",
}

# Every golden of every preset is byte-exact against llvmpipe, the sRGB8 passes included (the sRGB8 encode is
# llvmpipe's own RSQRTPS-based conversion, verified for every float in [0,1]: oracle/probes/srgb_encode_sweep.py).
BAR = {}
CASE_PASS_BAR = {}


def border_luts():
    return {"BORDER": (np.load(os.path.join(GOLD, "lut_border_synthetic.npy")), True, "clamp_to_border")}


def luts_for(key):
    if key.startswith("crt-royale"):
        return royale_luts()
    if key == "crt-potato-cool":
        return {"MASK": (np.load(os.path.join(GOLD, "lut_potato_mask_synthetic.npy")), False, "repeat")}
    if key == "sameboy-dmg-response-time":
        return {"COLOR_PALETTE": (np.load(os.path.join(GOLD, "lut_palette_synthetic.npy")), False, "clamp_to_border")}
    if key == "gb-palette-dmg":
        return {"COLOR_PALETTE": (np.load(os.path.join(GOLD, "lut_palette_synthetic.npy")), False, "clamp_to_border")}
    if key in ("reshade-lut", "reshade-gba"):
        n = 16 if key == "reshade-lut" else 32
        return {"SamplerLUT": (np.load(os.path.join(GOLD, "lut_color%d_synthetic.npy" % n)), True, "clamp_to_border")}
    if key in ("gba-lcd-grid-v2-3x", "gbc-retro-v2-2x", "gba-3x", "sgb-crt-geom-1x", "gameboy-player", "imgborder-sgb-bare", "ngpc-3x"):
        return border_luts()
    return None


def royale_luts():
    lut = np.load(os.path.join(GOLD, "lut_mask_slot_small_64.npy"))
    return {"mask_slot_texture_small": (lut, True, "repeat")}


def preset_passes(tmp_path, key):
    """Parse our hand-written preset with the product's parser (host only, no GPU)."""
    from retrocapture_amd import engine
    tree = chain_specs.write_tree(str(tmp_path))
    dump = engine.preset_dump(tree[key])
    passes = PassList(dump["passes"])
    # the preset file's own `parameters` block: applied at draw time, over custom values (oracle_chain.run_chain)
    passes.globals = {k: v for k, v in dump.get("params", {}).items() if k != "parameters"}
    return passes


class PassList(list):
    globals = None


def run_sequence(passes, frames_rgb, vw, vh, **kw):
    """Frame-by-frame through the oracle with the reference's cross-frame state (history ring)."""
    from oracle_chain import ChainState
    st = ChainState()
    outs = None
    for f in range(frames_rgb.shape[0]):
        outs = run_chain(passes, frames_rgb[f], vw, vh, frame_count=f + 1, state=st, **kw)
    return outs, st


HISTORY_PRESETS = ("history-size", "history-size-single", "sameboy-lcd-gbc-color-motionblur", "sameboy-dmg-response-time", "gba-3x", "gba-lcd-grid-v2-3x", "gbc-retro-v2-2x", "agb001-gba-color-motionblur", "lcd-grid-v2-psp-color-motionblur", "lcd-grid-v2-motionblur", "mix-frames", "motionblur-simple", "braid-rewind", "response-time", "mix-frames-smart", "shutter-3d", "anti-flicker")


@pytest.mark.parametrize("case", sorted(c for c in CASES if CASES[c] in HISTORY_PRESETS))
def test_oracle_frame_history_matches_llvmpipe(case, tmp_path, rc_lib):
    """Presets that sample frame history: the last frame's output AND the ring's content after
    3 ... 9 frames (first-frame rule, recursion through pass 0's program, wrap of the 7-deep ring - whose oldest
    texture is recycled as the target of the history draw, cleared, while still bound as Prev6Texture)."""
    g = np.load(os.path.join(GOLD, case + ".npz"))
    passes = preset_passes(tmp_path, CASES[case])
    vw, vh = [int(v) for v in g["viewport"]]
    custom = dict(zip([str(n) for n in g["param_names"]], [float(v) for v in g["param_values"]])) if "param_names" in g else None
    outs, st = run_sequence(passes, g["input_rgb"], vw, vh, custom=custom, luts=luts_for(CASES[case]))
    for i in range(int(g["n_passes"])):
        assert np.array_equal(outs[i], g["pass%d" % i]), "pass %d" % i
    assert len(st.history) == int(g["n_history"]) == min(7, g["input_rgb"].shape[0])
    for k, hk in enumerate(st.history):
        assert np.array_equal(hk, g["history%d" % k]), "history %d" % k


@pytest.mark.parametrize("case", sorted(c for c in CASES if CASES[c] == "feedback-persist"))
def test_oracle_pass_feedback_matches_llvmpipe(case, tmp_path, rc_lib):
    """PassFeedback binding and ping-pong swap, pinned with this repository's fixture shader on
    llvmpipe: the frame that first creates a feedback texture loses the declaring pass's draw."""
    g = np.load(os.path.join(GOLD, case + ".npz"))
    passes = preset_passes(tmp_path, CASES[case])
    vw, vh = [int(v) for v in g["viewport"]]
    custom = dict(zip([str(n) for n in g["param_names"]], [float(v) for v in g["param_values"]])) if "param_names" in g else None
    outs, st = run_sequence(passes, g["input_rgb"], vw, vh, custom=custom)
    for i in range(int(g["n_passes"])):
        assert np.array_equal(outs[i], g["pass%d" % i]), "pass %d" % i


@pytest.mark.parametrize("case", sorted(c for c in CASES if CASES[c] not in HISTORY_PRESETS + ("feedback-persist",)))
def test_oracle_matches_llvmpipe(case, tmp_path, rc_lib):
    g = np.load(os.path.join(GOLD, case + ".npz"))
    key = CASES[case]
    passes = preset_passes(tmp_path, key)
    vw, vh = [int(v) for v in g["viewport"]]
    flags = 1 if "maskon" in case else 0
    luts = luts_for(key)
    golden = [g["pass%d" % i] for i in range(int(g["n_passes"]))]
    custom = dict(zip([str(n) for n in g["param_names"]], [float(v) for v in g["param_values"]])) \
        if "param_names" in g else None
    # every pass fed with the GOLDEN outputs of the passes before it (isolates each pass) ...
    outs = run_chain(passes, g["input_rgb"], vw, vh, frame_count=int(g["frames"]), luts=luts, flags=flags,
                     given=golden, custom=custom)
    assert len(outs) == int(g["n_passes"])
    for i, o in enumerate(outs):
        floor, maxdiff = CASE_PASS_BAR.get((case, i), BAR.get(key, (1.0, 0)))
        ref = g["pass%d" % i]
        assert o.shape == ref.shape, (i, o.shape, ref.shape)
        if ref.dtype == np.uint8:
            d = np.abs(o.astype(np.int32) - ref.astype(np.int32))
            exact = float((d == 0).mean())
            fmt = str(g["pass%d_fmt" % i])
            if fmt == "rgba8":
                assert d.max() == 0, "RGBA8 pass %d must be bit-exact: exact %.5f max %d" % (i, exact, d.max())
            assert d.max() <= maxdiff and exact >= floor, "pass %d: exact %.5f max %d" % (i, exact, d.max())
        else:
            assert np.array_equal(o.view(np.uint32), ref.view(np.uint32)) or np.allclose(o, ref, rtol=1e-6, atol=1e-7)
    # ... and the whole chain end to end on the oracle's own intermediates
    if key.startswith("crt-royale"):
        own = run_chain(passes, g["input_rgb"], vw, vh, frame_count=int(g["frames"]), luts=luts, flags=flags, custom=custom)
        for i in range(len(golden)):
            assert np.array_equal(own[i], golden[i]), "end to end, pass %d" % i


def test_hyllian_phosphor_layouts(tmp_path, rc_lib):
    """resolve2.glsl mask_weights: all twenty PHOSPHOR_LAYOUT tables against llvmpipe (one golden: the five passes in front
    are shared, the last pass is stored per layout)."""
    g = np.load(os.path.join(GOLD, "crt_hyllian_glow_layouts_48x36_to_143x101.npz"))
    passes = preset_passes(tmp_path, "crt-hyllian-glow")
    given = [g["pass%d" % i] for i in range(5)] + [g["pass5_layout0"]]
    for lay in range(20):
        outs = run_chain(passes, g["input_rgb"], 143, 101, given=given,
                         custom={"PHOSPHOR_LAYOUT": float(lay), "MASK_INTENSITY": float(g["mask_intensity"])})
        assert np.array_equal(outs[5], g["pass5_layout%d" % lay]), "layout %d" % lay
        if lay:
            assert not np.array_equal(outs[5], g["pass5_layout0"])


# Float-precision goldens: the same shaders with every render target forced to RGBA32F on llvmpipe
# (tests/golden/make_golden.py case_float), each pass fed the GL's own float output of the passes
# before it.  Floor = fraction of float components that must be bit-identical; below 1.0 only where a
# last-bit residual is known and documented.
FLOAT_CASES = {
    "f32_scalefx_40x32": ("scalefx", {}),
    "f32_crt_geom_64x48_to_200x150": ("crt-geom", {}),
    "f32_crt_geom_params_64x48_to_200x150": ("crt-geom", {}),   # strong tilt: 42 pixels whose ray misses the tube (NaN geometry)
    "f32_scanline_64x48_to_160x100": ("scanline", {}),
    "f32_crt_pi_80x60_to_250x190": ("crt-pi", {}),
    "f32_ntsc_svideo_96x64_to_256x192": ("ntsc-256px-svideo", {}),
    "f32_ntsc_320px_72x40_to_320x120": ("ntsc-320px", {}),
    "f32_xbr_lv3_48x40_to_331x217": ("xbr-lv3", {}),
    "f32_gba_color_48x36_to_131x77": ("gba-color", {}),
    "f32_gbc_color_48x36_to_131x77": ("gbc-color", {}),
    "f32_gbc_gambatte_color_48x36_to_131x77": ("gbc-gambatte-color", {}),
    "f32_nds_color_48x36_to_131x77": ("nds-color", {}),
    "f32_palm_color_48x36_to_131x77": ("palm-color", {}),
    "f32_psp_color_48x36_to_131x77": ("psp-color", {}),
    "f32_vba_color_48x36_to_131x77": ("vba-color", {}),
    "f32_imgborder_sgb_bare_params_40x30_to_233x171": ("imgborder-sgb-bare", {}),
    "f32_sameboy_lcd_48x36_to_200x150": ("sameboy-lcd", {}),
    "f32_side_by_side_bare_params_48x36_to_200x150": ("side-by-side-bare", {}),
    "f32_advanced_aa_48x36_to_200x150": ("advanced-aa", {}),
    "f32_reverse_aa_48x36_to_200x150": ("reverse-aa", {}),
    "f32_crt_consumer_48x36_to_200x150": ("crt-consumer", {}),
    "f32_crt_lottes_48x36_to_200x150": ("crt-lottes", {}),
    "f32_fakelottes_48x36_to_200x150": ("fakelottes", {}),
    "f32_jinc2_sharper_48x36_to_200x150": ("jinc2-sharper", {}),
    "f32_interlacing_bare_40x420_to_160x420_f2": ("interlacing-bare", {}),
    "f32_tvout_tweaks_bare_params_48x36_to_200x150": ("tvout-tweaks-bare", {}),
    "f32_image_adjustment_bare_params_48x36_to_200x150_f2": ("image-adjustment-bare", {}),
    "f32_ntsc_gauss_scanline_72x40_to_256x160": ("ntsc-256px-svideo-gauss-scanline", {}),
    "f32_crt_potato_48x36_to_240x200": ("crt-potato-cool", {}),
    "f32_reshade_lut_48x36_to_131x77": ("reshade-lut", {}),
    "f32_lcd_grid_params_48x36_to_240x180": ("lcd-grid", {}),
    "f32_agb001_40x30_to_233x171": ("agb001", {}),
    "f32_retro_v2_48x36_to_240x180": ("retro-v2", {}),
    "f32_lcd_grid_v2_48x36_to_240x180": ("lcd-grid-v2", {}),
    "f32_lcd_grid_v2_params_40x30_to_233x171": ("lcd-grid-v2", {}),
    "f32_lcd_grid_v2_bare_params_40x30_to_233x171": ("lcd-grid-v2-bare", {}),
    "f32_crt_royale_64x48_to_128x96": ("crt-royale", {}),
    "f32_crt_royale_maskon_64x48_to_128x96": ("crt-royale", {}),
    "f32_crt_royale_geom_sphere_64x48_to_128x96": ("crt-royale", {}),
    "f32_crt_royale_geom_sphere_alt_tilt_64x48_to_128x96": ("crt-royale", {}),
    "f32_crt_royale_geom_cylinder_64x48_to_128x96": ("crt-royale", {}),
    "f32_crt_royale_geom_flat_overscan_64x48_to_128x96": ("crt-royale", {}),
    "f32_crt_royale_fake_bloom_geom_sphere_maskon_64x48_to_128x96": ("crt-royale-fake-bloom", {}),
    "f32_crt_royale_fake_bloom_geom_cylinder_tilt_maskon_64x48_to_128x96": ("crt-royale-fake-bloom", {}),   # non-separable coordinates: pins the quad rule
    "f32_crt_royale_fake_bloom_geom_flat_overscan_maskon_64x48_to_128x96": ("crt-royale-fake-bloom", {}),
    # pass 8 (the last pass, mipmap_input = true at 1:1): llvmpipe's trilinear LOD is a hair above 0 on some pixel
    # quads and blends a 1e-7 share of mip level 1 into the sample; restated (rc_sampler.c), bit-identical
    "f32_crt_royale_fake_bloom_maskon_64x48_to_128x96": ("crt-royale-fake-bloom", {}),
    "f32_zfast_crt_64x48_to_200x150": ("zfast-crt", {}),
    "f32_crt_nes_mini_64x48_to_200x150": ("crt-nes-mini", {}),
    "f32_quilez_64x48_to_200x150": ("quilez", {}),
    "f32_lcd3x_64x48_to_200x150": ("lcd3x", {}),
    "f32_lcd1x_64x48_to_200x150": ("lcd1x", {}),
    "f32_smootheststep_64x48_to_200x150": ("smootheststep", {}),
    "f32_sharp_bilinear_64x48_to_200x150": ("sharp-bilinear", {}),
    "f32_crt_easymode_64x48_to_200x150": ("crt-easymode", {}),
    "f32_xbr_lv2_48x40_to_331x217": ("xbr-lv2", {}),
    "f32_xbr_lv2_details_48x40_to_331x217": ("xbr-lv2", {}),
    "f32_crt_hyllian_glow_64x48_to_160x120": ("crt-hyllian-glow", {}),   # all six passes bit-identical (mip-mapped pass 3 included)
    "f32_crt_hyllian_glow_64x48_to_150x110": ("crt-hyllian-glow", {}),
}


@pytest.mark.parametrize("case", sorted(FLOAT_CASES))
def test_oracle_arithmetic_at_float_precision(case, tmp_path, rc_lib):
    key, floors = FLOAT_CASES[case]
    g = np.load(os.path.join(GOLD, case + ".npz"))
    passes = preset_passes(tmp_path, key)
    vw, vh = [int(v) for v in g["viewport"]]
    n = int(g["n_passes"])
    golden = [g["pass%d" % i] for i in range(n)]
    custom = dict(zip([str(x) for x in g["param_names"]], [float(v) for v in g["param_values"]])) if "param_names" in g else None
    outs = run_chain(passes, g["input_rgb"], vw, vh, frame_count=int(g["frames"]),
                     luts=luts_for(key), flags=1 if "maskon" in case else 0,
                     given=golden, force_f32=True, custom=custom)
    for i, (o, r) in enumerate(zip(outs, golden)):
        same = (o.view(np.uint32) == r.view(np.uint32)) | (np.isnan(o) & np.isnan(r))
        frac = float(same[..., :3].mean())
        assert frac >= floors.get(i, 1.0), "pass %d: %.5f of the float components bit-identical" % (i, frac)
        ulp = np.abs(o.view(np.int32).astype(np.int64) - r.view(np.int32).astype(np.int64))[..., :3][~same[..., :3]]
        assert ulp.size == 0 or ulp.max() <= 9000, "pass %d: max %d ulp" % (i, int(ulp.max()))


FLOAT_HISTORY = {"f32_console_border_gbc_retro_v2_2x_40x36_to_233x171_f3": "gbc-retro-v2-2x", "f32_shutter_3d_params_48x36_to_131x77_f5": "shutter-3d", "f32_anti_flicker_48x36_to_120x90_f6": "anti-flicker",
                 "f32_mix_frames_48x36_to_120x90_f3": "mix-frames", "f32_response_time_48x36_to_120x90_f9": "response-time",
                 "f32_mix_frames_smart_48x36_to_120x90_f8": "mix-frames-smart", "f32_motionblur_simple_48x36_to_120x90_f9": "motionblur-simple"}


@pytest.mark.parametrize("case", sorted(FLOAT_HISTORY))
def test_frame_history_at_float_precision(case, tmp_path, rc_lib):
    g = np.load(os.path.join(GOLD, case + ".npz"))
    passes = preset_passes(tmp_path, FLOAT_HISTORY[case])
    vw, vh = [int(v) for v in g["viewport"]]
    custom = dict(zip([str(n) for n in g["param_names"]], [float(v) for v in g["param_values"]])) if "param_names" in g else None
    outs, st = run_sequence(passes, g["input_rgb"], vw, vh, force_f32=True, custom=custom, luts=luts_for(FLOAT_HISTORY[case]))
    last = g["pass%d" % (int(g["n_passes"]) - 1)]
    assert np.array_equal(outs[-1].view(np.uint32), last.view(np.uint32))


WRAP_CASES = ["wrap_%s_%s_40x30_to_97x71" % (w, t) for w in ("clamp_to_edge", "clamp_to_border", "repeat", "mirrored_repeat")
              for t in ("rgba8", "srgb8")]


def wrap_case_preset(tmp_path, case):
    _, rest = case.split("_", 1)
    wrap = rest[: rest.index("_40x30")].rsplit("_", 1)[0]
    tag = rest[: rest.index("_40x30")].rsplit("_", 1)[1]
    root = tmp_path / "shaders_glsl"
    root.mkdir(exist_ok=True)
    p = root / (case + ".glslp")
    p.write_text("shaders = 2\nshader0 = crt/shaders/crt-pi.glsl\nfilter_linear0 = true\nwrap_mode0 = %s\nscale_type0 = source\n"
                 "scale0 = 2.0\n%sshader1 = stock.glsl\nfilter_linear1 = true\nwrap_mode1 = %s\n"
                 % (wrap, "srgb_framebuffer0 = true\n" if tag == "srgb8" else "", wrap))
    return str(p)


@pytest.mark.parametrize("case", WRAP_CASES)
def test_oracle_wrap_modes_match_llvmpipe(case, tmp_path, rc_lib):
    from retrocapture_amd import engine
    g = np.load(os.path.join(GOLD, case + ".npz"))
    passes = engine.preset_dump(wrap_case_preset(tmp_path, case))["passes"]
    vw, vh = [int(v) for v in g["viewport"]]
    golden = [g["pass0"], g["pass1"]]
    outs = run_chain(passes, g["input_rgb"], vw, vh, given=golden)
    for i in range(2):
        d = np.abs(outs[i].astype(np.int32) - golden[i].astype(np.int32))
        assert d.max() == 0, "pass %d: %d bytes differ" % (i, int((d != 0).sum()))   # sRGB8 targets included


BLIT_CASES = {"blit_nearest_60x45_to_540x405": (False, 540, 405), "blit_linear_60x45_to_540x405": (True, 540, 405),
              "blit_linear_160x120_to_233x150": (True, 233, 150)}


def blit_case_preset(tmp_path, linear=False):
    root = tmp_path / "shaders_glsl"
    root.mkdir(exist_ok=True)
    p = root / ("blit_linear.glslp" if linear else "blit_nearest.glslp")
    p.write_text("shaders = 2\nshader0 = crt/shaders/crt-pi.glsl\nfilter_linear0 = true\nscale_type0 = source\nscale0 = 2.0\n"
                 "shader1 = stock.glsl\nfilter_linear1 = %s\nwrap_mode1 = clamp_to_edge\n" % ("true" if linear else "false"))
    return str(p)


@pytest.mark.parametrize("case", sorted(BLIT_CASES))
def test_oracle_blit_fast_path(case, tmp_path, rc_lib):
    """stock copying an RGBA8 target with clamp to edge goes through llvmpipe's blit fast path (16.16
    fixed-point coordinate stepping, byte lerps); at 4.5x every other sample sits exactly on a texel
    boundary."""
    from retrocapture_amd import engine
    linear, vw, vh = BLIT_CASES[case]
    g = np.load(os.path.join(GOLD, case + ".npz"))
    passes = engine.preset_dump(blit_case_preset(tmp_path, linear))["passes"]
    outs = run_chain(passes, g["input_rgb"], vw, vh, given=[g["pass0"], g["pass1"]])
    assert np.array_equal(outs[0], g["pass0"])
    assert np.array_equal(outs[1], g["pass1"])


def test_every_golden_file_has_a_case():
    names = {os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz"))} - {"llvmpipe_tables"}
    names = {n for n in names if not n.startswith("present_")}   # tests/test_present.py
    assert names <= set(CASES) | set(FLOAT_CASES) | set(WRAP_CASES) | EXTRA_GOLDEN


EXTRA_GOLDEN = set(FLOAT_HISTORY) | set(BLIT_CASES) | {"crt_hyllian_glow_layouts_48x36_to_143x101"}   # test_hyllian_phosphor_layouts
