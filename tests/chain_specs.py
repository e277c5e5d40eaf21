"""Hand-written test presets (our own text, same keys as the reference's .glslp files) and
the description the oracle chain needs for each shader identity."""
import os

def _ntsc_preset(pass1, pass2, width):
    """Same keys / values as the reference's ntsc/ntsc-{256,320}px*.glslp (their frame_count_mod0 line is
    dropped by the reference parser, quirk Q2)."""
    return ("shaders = 2\nshader0 = shaders/ntsc-pass1-%s.glsl\nshader1 = shaders/ntsc-pass2-%s.glsl\n"
            "filter_linear0 = false\nfilter_linear1 = false\nscale_type_x0 = absolute\nscale_type_y0 = source\n"
            "scale_x0 = %d\nscale_y0 = 1.0\nframe_count_mod0 = 2\nfloat_framebuffer0 = true\n"
            "scale_type1 = source\nscale_x1 = 0.5\nscale_y1 = 1.0\n" % (pass1, pass2, width))


# preset name -> (relative path below shaders_glsl/, text)
PRESETS = {
    "scanline": ("scanlines/scanline.glslp",
                 'shaders = "1"\nshader0 = "shaders/scanline.glsl"\n'),
    "crt-pi": ("crt/crt-pi.glslp",
               'shaders = "1"\nshader0 = "shaders/crt-pi.glsl"\nfilter_linear0 = "true"\n'
               'wrap_mode0 = "clamp_to_border"\nmipmap_input0 = "false"\nalias0 = ""\n'
               'float_framebuffer0 = "false"\nsrgb_framebuffer0 = "false"\n'),
    # same keys / values as the reference's ntsc/ntsc-256px-svideo.glslp (its frame_count_mod0 line
    # is dropped by the reference parser, quirk Q2) and xbr/xbr-lv3.glslp
    "ntsc-256px-svideo": ("ntsc/ntsc-256px-svideo.glslp", """shaders = 2
shader0 = shaders/ntsc-pass1-svideo-3phase.glsl
shader1 = shaders/ntsc-pass2-3phase-gamma.glsl
filter_linear0 = false
filter_linear1 = false
scale_type_x0 = absolute
scale_type_y0 = source
scale_x0 = 1024
scale_y0 = 1.0
frame_count_mod0 = 2
float_framebuffer0 = true
scale_type1 = source
scale_x1 = 0.5
scale_y1 = 1.0
"""),
    "ntsc-256px": ("ntsc/ntsc-256px.glslp", _ntsc_preset("composite-3phase", "3phase-gamma", 1024)),
    "ntsc-320px": ("ntsc/ntsc-320px.glslp", _ntsc_preset("composite-2phase", "2phase-gamma", 1280)),
    "ntsc-320px-svideo": ("ntsc/ntsc-320px-svideo.glslp", _ntsc_preset("svideo-2phase", "2phase-gamma", 1280)),
    # synthesized: the -linear and plain pass-2 epilogues behind the shipped geometry
    "ntsc-3phase-linear": ("ntsc/t-3phase-linear.glslp", _ntsc_preset("svideo-3phase", "3phase-linear", 1024)),
    "ntsc-3phase-plain": ("ntsc/t-3phase-plain.glslp", _ntsc_preset("composite-3phase", "3phase", 1024)),
    "ntsc-2phase-linear": ("ntsc/t-2phase-linear.glslp", _ntsc_preset("composite-2phase", "2phase-linear", 1280)),
    "ntsc-2phase-plain": ("ntsc/t-2phase-plain.glslp", _ntsc_preset("svideo-2phase", "2phase", 1280)),
    "xbr-lv3": ("xbr/xbr-lv3.glslp", 'shaders = 1\n\nshader0 = shaders/xbr-lv3.glsl\nfilter_linear0 = false\n'),
    "xbr-lv2": ("xbr/xbr-lv2.glslp", 'shaders = 1\n\nshader0 = shaders/xbr-lv2.glsl\nfilter_linear0 = false\n'),
    # same keys / values as the reference's motionblur/mix_frames.glslp
    # same keys / values as the reference's other motionblur/ presets
    "motionblur-simple": ("motionblur/motionblur-simple.glslp", 'shaders = 1\n\nshader0 = shaders/motionblur-simple.glsl\nfilter_linear0 = false\n'),
    "braid-rewind": ("motionblur/braid-rewind.glslp", 'shaders = 1\n\nshader0 = shaders/braid-rewind.glsl\nfilter_linear0 = false\n'),
    # handheld/<name>-color.glslp: same keys / values as the reference's files
    "gba-color": ("handheld/gba-color.glslp", 'shaders = 1\n\nshader0 = shaders/color/gba-color.glsl\n\nfilter_linear0 = false\nscale_type0 = source\nscale0 = 1.0\n'),
    "gbc-color": ("handheld/gbc-color.glslp", 'shaders = 1\n\nshader0 = shaders/color/gbc-color.glsl\n\nfilter_linear0 = false\nscale_type0 = source\nscale0 = 1.0\n'),
    "gbc-gambatte-color": ("handheld/gbc-gambatte-color.glslp", 'shaders = 1\n\nshader0 = shaders/color/gbc-gambatte-color.glsl\n\nfilter_linear0 = false\nscale_type0 = source\nscale0 = 1.0\n'),
    "nds-color": ("handheld/nds-color.glslp", 'shaders = 1\n\nshader0 = shaders/color/nds-color.glsl\n\nfilter_linear0 = false\nscale_type0 = source\nscale0 = 1.0\n'),
    "palm-color": ("handheld/palm-color.glslp", 'shaders = 1\n\nshader0 = shaders/color/palm-color.glsl\n\nfilter_linear0 = false\nscale_type0 = source\nscale0 = 1.0\n'),
    "psp-color": ("handheld/psp-color.glslp", 'shaders = 1\n\nshader0 = shaders/color/psp-color.glsl\n\nfilter_linear0 = false\nscale_type0 = source\nscale0 = 1.0\n'),
    "vba-color": ("handheld/vba-color.glslp", 'shaders = 1\n\nshader0 = shaders/color/vba-color.glsl\n\nfilter_linear0 = false\nscale_type0 = source\nscale0 = 1.0\n'),
    # Same keys / values as the reference's stereoscopic-3d/shutter-to-side-by-side.glslp
    "shutter-3d": ("stereoscopic-3d/shutter-to-side-by-side.glslp", 'shaders = 1\n\nshader0 = shaders/shutter-3d.glsl\nwrap_mode0 = edge\n'),
    # misc/anti-flicker.glsl has no preset in the reference's tree: a one-pass chain of this repository
    "anti-flicker": ("misc/anti-flicker.glslp", 'shaders = 1\n\nshader0 = misc/anti-flicker.glsl\nfilter_linear0 = false\n'),
    "response-time": ("motionblur/response-time.glslp", 'shaders = 1\n\nshader0 = shaders/response-time.glsl\nfilter_linear0 = false\n'),
    "mix-frames-smart": ("motionblur/mix_frames_smart.glslp", 'shaders = "1"\n\nshader0 = "shaders/mix_frames_smart.glsl"\nfilter_linear0 = "false"\n'),
    "mix-frames": ("motionblur/mix_frames.glslp", 'shaders = "1"\n\nshader0 = "shaders/mix_frames.glsl"\nfilter_linear0 = "false"\n'),
    # PassFeedback conformance preset: the stock shader, then this repository's fixture shader
    "feedback-persist": ("feedback-persist.glslp",
                         'shaders = 2\nshader0 = stock.glsl\nfilter_linear0 = false\nscale_type0 = source\n'
                         'shader1 = conformance/feedback-persist.glsl\nfilter_linear1 = true\n'),
    # frame history through a pass 0 that READS its size uniforms (this repository's fixture shader), scaled 2x, stock behind it:
    # the history re-draw runs it on the final output with pass 0's stale TextureSize / OutputSize
    "history-size": ("history-size.glslp",
                     'shaders = 2\nshader0 = conformance/history-size.glsl\nfilter_linear0 = true\nscale_type0 = source\nscale0 = 2.0\n'
                     'shader1 = stock.glsl\nfilter_linear1 = true\n'),
    "history-size-single": ("history-size-single.glslp", 'shaders = 1\nshader0 = conformance/history-size.glsl\nfilter_linear0 = false\n'),
    # same keys / values as the reference's crt/crt-hyllian-glow.glslp (its smoke-test default preset)
    "crt-hyllian-glow": ("crt/crt-hyllian-glow.glslp", """shaders = 6

shader0 = shaders/glow/linearize.glsl
filter_linear0 = false
srgb_framebuffer0 = true

shader1 = shaders/hyllian/crt-hyllian-glow/crt-hyllian-glow.glsl
filter_linear1 = false
scale_type1 = viewport
scale1 = 1.0
srgb_framebuffer1 = true
alias1 = CRT_PASS

shader2 = shaders/glow/threshold.glsl
filter_linear2 = false
srgb_framebuffer2 = true

shader3 = shaders/glow/blur_horiz.glsl
mipmap_input3 = true
filter_linear3 = true
scale_type3 = source
scale3 = 0.25
srgb_framebuffer3 = true

shader4 = shaders/glow/blur_vert.glsl
filter_linear4 = true
srgb_framebuffer4 = true

shader5 = shaders/hyllian/crt-hyllian-glow/resolve2.glsl
filter_linear5 = true
"""),
    # same keys / values as the reference's interpolation/sharp-bilinear-2x-prescale.glslp and bilinear.glslp
    "sharp-bilinear-2x": ("interpolation/sharp-bilinear-2x-prescale.glslp",
                          'shaders = 2\nshader0 = ../stock.glsl\nfilter_linear0 = false\nscale_type0 = source\nscale0 = 2.0\n'
                          'shader1 = ../stock.glsl\nfilter_linear1 = true'),
    "bilinear": ("bilinear.glslp", 'shaders = 1\n\nshader0 = stock.glsl\nfilter_linear0 = true\n'),
    # synthesized: mipmap_input on the GL_RGB source frame / on a plain RGBA8 render target (8-bit mip generation and blend)
    **{"mip-source-%s" % sc: ("crt/t-mip-source-%s.glslp" % sc,
                              "shaders = 1\nshader0 = shaders/glow/blur_horiz.glsl\nfilter_linear0 = true\nmipmap_input0 = true\n"
                              "scale_type0 = source\nscale0 = %s\n" % sc) for sc in ("0.4", "0.23")},
    **{"mip-rgba8-%s" % sc: ("crt/t-mip-rgba8-%s.glslp" % sc,
                             "shaders = 2\nshader0 = ../stock.glsl\nfilter_linear0 = false\nscale_type0 = source\nscale0 = 1.0\n"
                             "shader1 = shaders/glow/blur_horiz.glsl\nfilter_linear1 = true\nmipmap_input1 = true\n"
                             "scale_type1 = source\nscale1 = %s\n" % sc) for sc in ("0.37", "0.6")},
    # ... and without filter_linear: GL_NEAREST_MIPMAP_NEAREST
    **{"mipnearest-source-%s" % sc: ("crt/t-mipnearest-source-%s.glslp" % sc,
                                     "shaders = 1\nshader0 = shaders/glow/blur_horiz.glsl\nfilter_linear0 = false\nmipmap_input0 = true\n"
                                     "scale_type0 = source\nscale0 = %s\n" % sc) for sc in ("0.4", "0.17")},
    "mipnearest-rgba8-0.6": ("crt/t-mipnearest-rgba8-0.6.glslp",
                             "shaders = 2\nshader0 = ../stock.glsl\nfilter_linear0 = false\nscale_type0 = source\nscale0 = 1.0\n"
                             "shader1 = shaders/glow/blur_horiz.glsl\nfilter_linear1 = false\nmipmap_input1 = true\n"
                             "scale_type1 = source\nscale1 = 0.6\n"),
    "crt-geom": ("crt/crt-geom.glslp", 'shaders = 1\n\nshader0 = shaders/crt-geom.glsl\nfilter_linear0 = false\n'),
    "crt-easymode": ("crt/crt-easymode.glslp", 'shaders = 1\n\nshader0 = shaders/crt-easymode.glsl\nfilter_linear0 = false\n'),
    "crt-nes-mini": ("crt/crt-nes-mini.glslp", 'shaders = 1\n\nshader0 = shaders/crt-nes-mini.glsl\n'),
    # same keys / values as the reference's scalefx/scalefx.glslp
    "scalefx": ("scalefx/scalefx.glslp", """shaders = 5

shader0 = shaders/scalefx-pass0.glsl
filter_linear0 = false
scale_type0 = source
scale0 = 1.0
float_framebuffer0 = true

shader1 = shaders/scalefx-pass1.glsl
filter_linear1 = false
scale_type1 = source
scale1 = 1.0
float_framebuffer1 = true

shader2 = shaders/scalefx-pass2.glsl
filter_linear2 = false
scale_type2 = source
scale2 = 1.0

shader3 = shaders/scalefx-pass3.glsl
filter_linear3 = false
scale_type3 = source
scale3 = 1.0

shader4 = shaders/scalefx-pass4.glsl
filter_linear4 = false
scale_type4 = source
scale4 = 3.0
"""),
    "epx": ("scalenx/epx.glslp", 'shaders = 1\n\nshader0 = shaders/epx.glsl\nfilter_linear0 = false\nscale_type0 = source\nscale0 = 2.0\n'),
    "lcd1x": ("handheld/lcd1x.glslp", 'shaders = "1"\n\nshader0 = "shaders/lcd1x.glsl"\n\nfilter_linear0 = "false"\nwrap_mode0 = "clamp_to_border"\n'
                                      'mipmap_input0 = "false"\nalias0 = ""\nfloat_framebuffer0 = "false"\nsrgb_framebuffer0 = "false"\n'),
    "lcd3x": ("handheld/lcd3x.glslp", 'shaders = 1\n\nshader0 = shaders/lcd3x.glsl\nfilter_linear0 = false'),
    "bayer": ("dithering/bayer-matrix-dithering.glslp", 'shaders = 1\n\nshader0 = shaders/bayer-matrix-dithering.glsl\nfilter_linear0 = false'),
    "quilez": ("interpolation/quilez.glslp", 'shaders = 1\n\nshader0 = shaders/quilez.glsl\nfilter_linear0 = true'),
    "smootheststep": ("interpolation/smootheststep.glslp", 'shaders = 1\n\nshader0 = shaders/smootheststep.glsl\nfilter_linear0 = true'),
    "sharp-bilinear": ("interpolation/sharp-bilinear.glslp", 'shaders = 1\n\nshader0 = shaders/sharp-bilinear.glsl\nfilter_linear0 = true'),
    "zfast-crt": ("crt/zfast-crt.glslp", 'shaders = 1\n\nshader0 = shaders/zfast_crt.glsl\nfilter_linear0 = true'),
    "stock": ("stock.glslp", 'shaders = "1"\nshader0 = "stock.glsl"\nfilter_linear0 = "false"\n'),
    # Same keys / values as the reference's crt/crt-royale.glslp for the 12 passes, including the
    # three `"true" # comment` booleans that its parser reads as false; only the LUT that the
    # default (slot mask) path samples is declared.
    "crt-royale": ("crt/crt-royale.glslp", """shaders = "12"
textures = "mask_slot_texture_small"
mask_slot_texture_small = "shaders/crt-royale/mask_slot_small_64.png"
mask_slot_texture_small_wrap_mode = "repeat"
mask_slot_texture_small_linear = "true"
mask_slot_texture_small_mipmap = "false"  # trailing comments make this parse as false anyway
shader0 = "shaders/crt-royale/src/crt-royale-first-pass-linearize-crt-gamma-bob-fields.glsl"
alias0 = "ORIG_LINEARIZED"
filter_linear0 = "false"
scale_type0 = "source"
scale0 = "1.0"
srgb_framebuffer0 = "true"
shader1 = "shaders/crt-royale/src/crt-royale-scanlines-vertical-interlacing.glsl"
alias1 = "VERTICAL_SCANLINES"
filter_linear1 = "true"
scale_type_x1 = "source"
scale_x1 = "1.0"
scale_type_y1 = "viewport"
scale_y1 = "1.0"
srgb_framebuffer1 = "true"
shader2 = "shaders/crt-royale/src/crt-royale-bloom-approx.glsl"
alias2 = "BLOOM_APPROX"
filter_linear2 = "true"
scale_type2 = "absolute"
scale_x2 = "320"
scale_y2 = "240"
srgb_framebuffer2 = "true"
shader3 = "../blurs/blur9fast-vertical.glsl"
filter_linear3 = "true"
scale_type3 = "source"
scale3 = "1.0"
srgb_framebuffer3 = "true"
shader4 = "../blurs/blur9fast-horizontal.glsl"
alias4 = "HALATION_BLUR"
filter_linear4 = "true"
scale_type4 = "source"
scale4 = "1.0"
srgb_framebuffer4 = "true"
shader5 = "shaders/crt-royale/src/crt-royale-mask-resize-vertical.glsl"
filter_linear5 = "true"
scale_type_x5 = "absolute"
scale_x5 = "64"
scale_type_y5 = "viewport"
scale_y5 = "0.0625" # viewport fraction
shader6 = "shaders/crt-royale/src/crt-royale-mask-resize-horizontal.glsl"
alias6 = "MASK_RESIZE"
filter_linear6 = "false"
scale_type_x6 = "viewport"
scale_x6 = "0.0625"
scale_type_y6 = "source"
scale_y6 = "1.0"
shader7 = "shaders/crt-royale/src/crt-royale-scanlines-horizontal-apply-mask.glsl"
alias7 = "MASKED_SCANLINES"
filter_linear7 = "true" # parsed as false
scale_type7 = "viewport"
scale7 = "1.0"
srgb_framebuffer7 = "true"
shader8 = "shaders/crt-royale/src/crt-royale-brightpass.glsl"
alias8 = "BRIGHTPASS"
filter_linear8 = "true" # parsed as false
scale_type8 = "viewport"
scale8 = "1.0"
srgb_framebuffer8 = "true"
shader9 = "shaders/crt-royale/src/crt-royale-bloom-vertical.glsl"
filter_linear9 = "true" # parsed as false
scale_type9 = "source"
scale9 = "1.0"
srgb_framebuffer9 = "true"
shader10 = "shaders/crt-royale/src/crt-royale-bloom-horizontal-reconstitute.glsl"
filter_linear10 = "true"
scale_type10 = "source"
scale10 = "1.0"
srgb_framebuffer10 = "true"
shader11 = "shaders/crt-royale/src/crt-royale-geometry-aa-last-pass.glsl"
filter_linear11 = "true"
scale_type11 = "viewport"
mipmap_input11 = "false"
texture_wrap_mode11 = "clamp_to_edge"
"""),    # Same keys / values as the reference's crt/crt-royale-fake-bloom.glslp (9 passes: crt-royale's 0-7 and
    # its last pass, with the PHOSPHOR_BLOOM_FAKE variants of passes 2 and 7; BLOOM_APPROX is 400x300)
    "crt-royale-fake-bloom": ("crt/crt-royale-fake-bloom.glslp", """shaders = "9"
textures = "mask_slot_texture_small"
mask_slot_texture_small = "shaders/crt-royale/mask_slot_small_64.png"
mask_slot_texture_small_wrap_mode = "repeat"
mask_slot_texture_small_linear = "true"
mask_slot_texture_small_mipmap = "false"  # trailing comments make this parse as false anyway
shader0 = "shaders/crt-royale/src/crt-royale-first-pass-linearize-crt-gamma-bob-fields.glsl"
alias0 = "ORIG_LINEARIZED"
filter_linear0 = "false"
scale_type0 = "source"
scale0 = "1.0"
srgb_framebuffer0 = "true"
shader1 = "shaders/crt-royale/src/crt-royale-scanlines-vertical-interlacing.glsl"
alias1 = "VERTICAL_SCANLINES"
filter_linear1 = "true"
scale_type_x1 = "source"
scale_x1 = "1.0"
scale_type_y1 = "viewport"
scale_y1 = "1.0"
srgb_framebuffer1 = "true"
shader2 = "shaders/crt-royale/src/crt-royale-bloom-approx-fake-bloom.glsl"
alias2 = "BLOOM_APPROX"
filter_linear2 = "true"
scale_type2 = "absolute"
scale_x2 = "400"
scale_y2 = "300"
srgb_framebuffer2 = "true"
shader3 = "../blurs/blur9fast-vertical.glsl"
filter_linear3 = "true"
scale_type3 = "source"
scale3 = "1.0"
srgb_framebuffer3 = "true"
shader4 = "../blurs/blur9fast-horizontal.glsl"
alias4 = "HALATION_BLUR"
filter_linear4 = "true"
scale_type4 = "source"
scale4 = "1.0"
srgb_framebuffer4 = "true"
shader5 = "shaders/crt-royale/src/crt-royale-mask-resize-vertical.glsl"
filter_linear5 = "true"
scale_type_x5 = "absolute"
scale_x5 = "64"
scale_type_y5 = "viewport"
scale_y5 = "0.0625" # viewport fraction
shader6 = "shaders/crt-royale/src/crt-royale-mask-resize-horizontal.glsl"
alias6 = "MASK_RESIZE"
filter_linear6 = "false"
scale_type_x6 = "viewport"
scale_x6 = "0.0625"
scale_type_y6 = "source"
scale_y6 = "1.0"
shader7 = "shaders/crt-royale/src/crt-royale-scanlines-horizontal-apply-mask-fake-bloom.glsl"
alias7 = "MASKED_SCANLINES"
filter_linear7 = "true" # This could just as easily be nearest neighbor.
scale_type7 = "viewport"
scale7 = "1.0"
srgb_framebuffer7 = "true"
shader8 = "shaders/crt-royale/src/crt-royale-geometry-aa-last-pass.glsl"
filter_linear8 = "true"
scale_type8 = "viewport"
mipmap_input8 = "true"
texture_wrap_mode8 = "clamp_to_edge"
"""),

}

def _royale_ntsc(pass1, pass2, width):
    """crt/crt-royale-ntsc-*.glslp: two ntsc passes in front of crt-royale's twelve (same keys / values as the
    reference's files; frame_count_mod0 is there too - the reference's parser ignores it)."""
    body = PRESETS["crt-royale"][1]
    body = re.sub(r'^(\w+?)(\d+) = ', lambda m: "%s%d = " % (m.group(1), int(m.group(2)) + 2), body, flags=re.M)
    body = body.replace('shaders = "12"\n', "").replace('mipmap_input13 = "false"', 'mipmap_input13 = "true"')
    head = ('shaders = "14"\nshader0 = "../ntsc/shaders/ntsc-pass1-%s.glsl"\nshader1 = "../ntsc/shaders/ntsc-pass2-%s.glsl"\n'
            'filter_linear0 = false\nfilter_linear1 = false\nscale_type_x0 = absolute\nscale_type_y0 = source\nscale_x0 = %d\n'
            'scale_y0 = 1.0\nframe_count_mod0 = 2\nfloat_framebuffer0 = true\nscale_type1 = source\nscale_x1 = 0.5\nscale_y1 = 1.0\n'
            % (pass1, pass2, width))
    return head + body


import re
# handheld/lcd-grid-v2.glslp and the lcd-grid-v2-<colour>[-motionblur] chains: same passes, keys and parameter values as the
# reference's files (BGR is 1 in all but the plain, the -motionblur and the gbc ones)
def _lcd_grid_v2(colour, motionblur):
    passes = []
    if motionblur:
        passes.append(("../motionblur/shaders/response-time.glsl", "source"))
    passes.append(("shaders/lcd-cgwg/lcd-grid-v2.glsl", "viewport"))
    if colour:
        passes.append(("shaders/color/%s-color.glsl" % colour, "source"))
    t = 'shaders = "%d"\n\n' % len(passes)
    for i, (sh, st) in enumerate(passes):
        t += 'shader%d = "%s"\nfilter_linear%d = "false"\nscale_type%d = "%s"\nscale%d = "1.0"\n\n' % (i, sh, i, i, st, i)
    names = ["RSUBPIX_R", "RSUBPIX_G", "RSUBPIX_B", "GSUBPIX_R", "GSUBPIX_G", "GSUBPIX_B", "BSUBPIX_R", "BSUBPIX_G", "BSUBPIX_B", "gain", "gamma",
             "blacklevel", "ambient", "BGR"]
    vals = [0.75, 0, 0, 0, 0.75, 0, 0, 0, 0.75, 1.5, 2.2, 0, 0, 1.0 if colour not in (None, "gbc") else 0.0]
    t += 'parameters = "%s"\n' % ";".join(names)
    for n, v in zip(names, vals):
        t += '%s = "%f"\n' % (n, v)
    return t


PRESETS["lcd-grid-v2"] = ("handheld/lcd-grid-v2.glslp", _lcd_grid_v2(None, False))
# the shader alone, without the preset files' parameter block (a one-pass chain of this repository): the #pragma defaults apply
PRESETS["lcd-grid-v2-bare"] = ("handheld/lcd-grid-v2-bare.glslp", 'shaders = 1\nshader0 = shaders/lcd-cgwg/lcd-grid-v2.glsl\nfilter_linear0 = false\nscale_type0 = viewport\n')
PRESETS["lcd-grid-v2-motionblur"] = ("handheld/lcd-grid-v2-motionblur.glslp", _lcd_grid_v2(None, True))
for _c in ("gba", "gbc", "nds", "palm", "psp", "vba"):
    PRESETS["lcd-grid-v2-%s-color" % _c] = ("handheld/lcd-grid-v2-%s-color.glslp" % _c, _lcd_grid_v2(_c, False))
    PRESETS["lcd-grid-v2-%s-color-motionblur" % _c] = ("handheld/lcd-grid-v2-%s-color-motionblur.glslp" % _c, _lcd_grid_v2(_c, True))

# handheld/retro-v2.glslp and presets/retro-v2+<console>-color.glslp (same keys / values as the reference's files; the parameter
# blocks of the nds / psp / vba ones name parameters their shaders do not declare - kept, they are inert)
PRESETS["retro-v2"] = ("handheld/retro-v2.glslp", 'shaders = 1\n\nshader0 = shaders/retro-v2.glsl\nfilter_linear0 = false')
for _c, _blk in (("gba", 'parameters = "darken_screen;RETRO_PIXEL_SIZE"\ndarken_screen = "1.000000"\nRETRO_PIXEL_SIZE = "0.840000"\n'),
                 ("gbc", 'parameters = "RETRO_PIXEL_SIZE"\nRETRO_PIXEL_SIZE = "0.840000"\n'),
                 ("nds", 'parameters = "target_gamma;RETRO_PIXEL_SIZE"\ntarget_gamma = "2.000000"\nRETRO_PIXEL_SIZE = "0.840000"\n'),
                 ("psp", 'parameters = "target_gamma;RETRO_PIXEL_SIZE"\ntarget_gamma = "2.200000"\nRETRO_PIXEL_SIZE = "0.840000"\n'),
                 ("vba", 'parameters = "dark_gamma;RETRO_PIXEL_SIZE"\ndark_gamma = "2.900000"\nRETRO_PIXEL_SIZE = "0.840000"\n')):
    PRESETS["retro-v2+%s-color" % _c] = ("presets/retro-v2+%s-color.glslp" % _c,
                                         'shaders = "2"\n\nshader0 = "../handheld/shaders/color/%s-color.glsl"\nshader1 = "../handheld/shaders/retro-v2.glsl"\n\n'
                                         'filter_linear0 = "false"\nscale_type0 = "source"\nscale0 = "1.000000"\n\nfilter_linear1 = "false"\n\n' % _c + _blk)

# handheld/console-border/ngpc-3x.glslp (same keys / values; synthetic border image)
PRESETS["ngpc-3x"] = ("handheld/console-border/ngpc-3x.glslp", """shaders = "2"

shader0 = "../shaders/lcd-cgwg/lcd-grid.glsl"
filter_linear0 = "false"
wrap_mode0 = "clamp_to_border"
scale_type_x0 = "source"
scale_x0 = "4.000000"
scale_type_y0 = "source"
scale_y0 = "4.000000"

shader1 = "shader-files/border.glsl"
filter_linear1 = "true"
wrap_mode1 = "clamp_to_border"

parameters = "box_scale;in_res_x;in_res_y;border_on_top;border_zoom_x;border_zoom_y"
GRID_STRENGTH = "0.150000"
box_scale = "3.0"
in_res_x = "160.0"
in_res_y = "152.0"
border_on_top = "0.000000"
border_zoom_x = "0.70"
border_zoom_y = "0.79"

textures = "BORDER"
BORDER = "resources/ngpc-border-square-4x.png"
BORDER_linear = "true"
BORDER_wrap_mode = "clamp_to_border"
BORDER_mipmap = "false"
""")

# presets/tvout/tvout.glslp, presets/tvout/tvout+ntsc-256px-svideo.glslp and presets/retro-v2+image-adjustment.glslp: same keys / values as the
# reference's files (their parameter blocks name `target_gamma` ... without the shader's `ia_` prefix: inert, kept)
PRESETS["tvout"] = ("presets/tvout/tvout.glslp", '#tvout preset for 240p CRTs\n\nshaders = "2"\nshader0 = "../../crt/shaders/tvout-tweaks.glsl"\nshader1 = "../../misc/image-adjustment.glsl"\n\nscale_type_x0 = "viewport"\nscale_x0 = "1.000000"\nscale_type_y0 = "source"\nscale_y0 = "1.000000"\n\nparameters = "TVOUT_RESOLUTION;TVOUT_COMPOSITE_CONNECTION;TVOUT_TV_COLOR_LEVELS;target_gamma;monitor_gamma;overscan_percent_x;overscan_percent_y;saturation;contrast;luminance;bright_boost;R;G;B"\nTVOUT_RESOLUTION = "320.000000"\nTVOUT_COMPOSITE_CONNECTION = "0.000000"\nTVOUT_TV_COLOR_LEVELS = "1.000000"\ntarget_gamma = "2.400000"\nmonitor_gamma = "2.200000"\noverscan_percent_x = "0.000000"\noverscan_percent_y = "0.000000"\nsaturation = "1.000000"\ncontrast = "1.000000"\nluminance = "1.000000"\nbright_boost = "0.000000"\nR = "1.000000"\nG = "1.000000"\nB = "1.000000"\n')
PRESETS["tvout+ntsc-256px-svideo"] = ("presets/tvout/tvout+ntsc-256px-svideo.glslp", 'shaders = "4"\n\nshader0 = "../../ntsc/shaders/ntsc-pass1-svideo-3phase.glsl"\nfilter_linear0 = "false"\nframe_count_mod0 = "2"\nfloat_framebuffer0 = "true"\nscale_type_x0 = "absolute"\nscale_x0 = "1536"\nscale_type_y0 = "source"\nscale_y0 = "1.000000"\n\nshader1 = "../../ntsc/shaders/ntsc-pass2-3phase.glsl"\nfilter_linear1 = "false"\nfloat_framebuffer1 = "false"\nscale_type_x1 = "source"\nscale_x1 = "0.500000"\nscale_type_y1 = "source"\nscale_y1 = "1.000000"\n\nshader2 = "../../crt/shaders/tvout-tweaks.glsl"\nfilter_linear2 = "false"\nfloat_framebuffer2 = "false"\nscale_type_x2 = "viewport"\nscale_x2 = "1.000000"\nscale_type_y2 = "source"\nscale_y2 = "1.000000"\n\nshader3 = "../../misc/image-adjustment.glsl"\nfloat_framebuffer3 = "false"\n\nparameters = "TVOUT_RESOLUTION;TVOUT_COMPOSITE_CONNECTION;TVOUT_TV_COLOR_LEVELS;target_gamma;monitor_gamma;overscan_percent_x;overscan_percent_y;saturation;contrast;luminance;bright_boost;R;G;B"\nTVOUT_RESOLUTION = "512.000000"\nTVOUT_COMPOSITE_CONNECTION = "0.000000"\nTVOUT_TV_COLOR_LEVELS = "1.000000"\ntarget_gamma = "2.400000"\nmonitor_gamma = "2.200000"\noverscan_percent_x = "0.000000"\noverscan_percent_y = "0.000000"\nsaturation = "1.000000"\ncontrast = "1.000000"\nluminance = "1.000000"\nbright_boost = "0.000000"\nR = "1.000000"\nG = "1.000000"\nB = "1.000000"')
PRESETS["retro-v2+image-adjustment"] = ("presets/retro-v2+image-adjustment.glslp", 'shaders = "2"\n\nshader0 = "../misc/image-adjustment.glsl"\nshader1 = "../handheld/shaders/retro-v2.glsl"\n\nfilter_linear0 = "false"\nscale_type0 = "source"\nscale0 = "1.000000"\n\nfilter_linear1 = "false"\n\nparameters = "target_gamma;monitor_gamma;overscan_percent_x;overscan_percent_y;saturation;contrast;luminance;bright_boost;R;G;B;RETRO_PIXEL_SIZE"\ntarget_gamma = "2.200000"\nmonitor_gamma = "2.20000"\noverscan_percent_x = "0.000000"\noverscan_percent_y = "0.000000"\nsaturation = "1.000000"\ncontrast = "1.000000"\nluminance = "1.000000"\nbright_boost = "0.000000"\nR = "1.000000"\nG = "1.000000"\nB = "1.000000"\nRETRO_PIXEL_SIZE = "0.840000"\n')
PRESETS["tvout+interlacing"] = ("presets/tvout+interlacing/tvout+interlacing.glslp", '#tvout preset for 480p CRTs\n\nshaders = "3"\nshader0 = "../../crt/shaders/tvout-tweaks.glsl"\nshader1 = "../../misc/image-adjustment.glsl"\nshader2 = "../../misc/interlacing.glsl"\n\nscale_type_x0 = "viewport"\nscale_x0 = "1.000000"\nscale_type_y0 = "source"\nscale_y0 = "1.000000"\n\nparameters = "TVOUT_RESOLUTION;TVOUT_COMPOSITE_CONNECTION;TVOUT_TV_COLOR_LEVELS;target_gamma;monitor_gamma;overscan_percent_x;overscan_percent_y;saturation;contrast;luminance;bright_boost;R;G;B"\nTVOUT_RESOLUTION = "320.000000"\nTVOUT_COMPOSITE_CONNECTION = "0.000000"\nTVOUT_TV_COLOR_LEVELS = "1.000000"\ntarget_gamma = "2.400000"\nmonitor_gamma = "2.200000"\noverscan_percent_x = "0.000000"\noverscan_percent_y = "0.000000"\nsaturation = "1.000000"\ncontrast = "1.000000"\nluminance = "1.000000"\nbright_boost = "0.000000"\nR = "1.000000"\nG = "1.000000"\nB = "1.000000"')
PRESETS["side-by-side"] = ("stereoscopic-3d/side-by-side.glslp", 'shaders = 1\n\nshader0 = shaders/side-by-side-simple.glsl\n\nparameters = "eye_sep;y_loc;ana_zoom"\neye_sep = "0.30"\ny_loc = "0.25"\nana_zoom = "0.50"\n')
PRESETS["sbs-warp-mobile-16x9"] = ("stereoscopic-3d/sbs-warp-mobile-16x9.glslp", 'shaders = 1\n\nshader0 = shaders/side-by-side-simple.glsl\n\nparameters = "eye_sep;y_loc;BOTH;ana_zoom;warpY;warpX"\neye_sep = "0.125"\ny_loc = "0.10"\nBOTH = "0.255"\nana_zoom = "0.66"\nwarpY = "0.1"\nwarpX = "0.1"\n')
PRESETS["side-by-side-bare"] = ("stereoscopic-3d/side-by-side-bare.glslp", 'shaders = 1\nshader0 = shaders/side-by-side-simple.glsl\n')
PRESETS["sameboy-lcd"] = ("handheld/sameboy-lcd.glslp", 'shaders = 1\n\nshader0 = shaders/sameboy-lcd.glsl\nfilter_linear0 = false\nscale_type0 = "viewport"\nscale0 = "1.0"\n')
PRESETS["sameboy-lcd-gbc-color-motionblur"] = ("handheld/sameboy-lcd-gbc-color-motionblur.glslp", 'shaders = 3\n\nshader0 = ../motionblur/shaders/response-time.glsl\nfilter_linear0 = false\nscale_type0 = source\nscale0 = 1.0\n\nshader1 = shaders/sameboy-lcd.glsl\nfilter_linear1 = false\nscale_type1 = viewport\nscale1 = "1.0"\n\nshader2 = shaders/color/gbc-color.glsl\nfilter_linear2 = false\nscale_type2 = viewport\n')
PRESETS["crt-consumer"] = ("crt/crt-consumer.glslp", 'shaders = "1"\nshader0 = "shaders/crt-consumer.glsl"\nfilter_linear0 = "true"\nwrap_mode0 = "clamp_to_border"\nmipmap_input0 = "false"\nalias0 = ""\nfloat_framebuffer0 = "false"\nsrgb_framebuffer0 = "false"\n\n')
PRESETS["reverse-aa"] = ("anti-aliasing/reverse-aa.glslp", 'shaders = 1\n\nshader0 = shaders/reverse-aa.glsl\nfilter_linear0 = false\nscale_type0 = source\nscale0 = 2.0\n')
PRESETS["advanced-aa"] = ("anti-aliasing/advanced-aa.glslp", 'shaders = 2\n\nshader0 = shaders/advanced-aa.glsl\nfilter_linear0 = false\nscale_type0 = source\nscale_x0 = 2.0\nscale_y0 = 2.0\n\nshader1 = ../stock.glsl\nfilter_linear1 = true\n')
PRESETS["crt-lottes"] = ("crt/crt-lottes.glslp", 'shaders = 1\n\nshader0 = shaders/crt-lottes.glsl\nfilter_linear0 = false\n')
PRESETS["fakelottes"] = ("crt/fakelottes.glslp", 'shaders = 1\n\nshader0 = shaders/fakelottes.glsl\nfilter_linear0 = true\n')
PRESETS["jinc2-sharper"] = ("windowed/jinc2-sharper.glslp", 'shaders = 1\n\nshader0 = shaders/jinc2-sharper.glsl\nfilter_linear0 = false\n')
PRESETS["tvout-jinc-sharpen"] = ("presets/tvout/tvout-jinc-sharpen.glslp", '#try to sharpen blended output with Jinc2 set with very low Window Sinc parameter\n\nshaders = "3"\nshader0 = "../../crt/shaders/tvout-tweaks.glsl"\nshader1 = "../../misc/image-adjustment.glsl"\nshader2 = "../../windowed/shaders/jinc2-sharper.glsl"\nshader3 = "../../misc/interlacing.glsl"\n\nscale_type_x0 = "source"\nscale_x0 = "2.000000"\nscale_type_y0 = "source"\nscale_y0 = "1.000000"\n\n\nscale_type_x2 = "viewport"\nscale_x2 = "1.000000"\nscale_type_y2 = "source"\nscale_y2 = "1.000000"\n\n\n\nparameters = "TVOUT_RESOLUTION;TVOUT_COMPOSITE_CONNECTION;TVOUT_TV_COLOR_LEVELS;target_gamma;monitor_gamma;overscan_percent_x;overscan_percent_y;saturation;contrast;luminance;bright_boost;R;G;B;JINC2_WINDOW_SINC;JINC2_SINC;JINC2_AR_STRENGTH"\nTVOUT_RESOLUTION = "160.000000"\nTVOUT_COMPOSITE_CONNECTION = "0.000000"\nTVOUT_TV_COLOR_LEVELS = "0.000000"\ntarget_gamma = "2.400000"\nmonitor_gamma = "2.200000"\noverscan_percent_x = "0.000000"\noverscan_percent_y = "0.000000"\nsaturation = "1.000000"\ncontrast = "1.000000"\nluminance = "1.000000"\nbright_boost = "0.000000"\nR = "1.000000"\nG = "1.000000"\nB = "1.000000"\nJINC2_WINDOW_SINC = "0.010000"\nJINC2_SINC = "0.9"\nJINC2_AR_STRENGTH = "0.900000"\n')
PRESETS["interlacing-bare"] = ("misc/interlacing-bare.glslp", 'shaders = 1\nshader0 = misc/interlacing.glsl\nfilter_linear0 = false\n')
# image-adjustment alone, no parameter block (a one-pass chain of this repository): every parameter can move
PRESETS["image-adjustment-bare"] = ("misc/image-adjustment-bare.glslp", 'shaders = 1\nshader0 = misc/image-adjustment.glsl\nfilter_linear0 = false\n')
PRESETS["tvout-tweaks-bare"] = ("crt/tvout-tweaks-bare.glslp", 'shaders = 1\nshader0 = shaders/tvout-tweaks.glsl\nfilter_linear0 = false\n')

# ntsc/ntsc-256px-svideo-gauss-scanline.glslp (same keys / values)
PRESETS["ntsc-256px-svideo-gauss-scanline"] = ("ntsc/ntsc-256px-svideo-gauss-scanline.glslp", """shaders = 4
shader0 = shaders/ntsc-pass1-svideo-3phase.glsl
shader1 = shaders/ntsc-pass2-3phase.glsl
shader2 = shaders/ntsc-gauss-pass.glsl
shader3 = shaders/ntsc-stock.glsl

filter_linear0 = false
filter_linear1 = false
filter_linear2 = false
filter_linear3 = true 

scale_type_x0 = absolute
scale_type_y0 = source
scale_x0 = 1024
scale_y0 = 1.0
frame_count_mod0 = 2
float_framebuffer0 = true

scale_type1 = source
scale_x1 = 0.5
scale_y1 = 1.0

scale_type_x2 = source
scale_type_y2 = viewport
scale2 = 1.0

""")

# crt/crt-potato-cool.glslp (same keys / values; synthetic mask image)
PRESETS["crt-potato-cool"] = ("crt/crt-potato-cool.glslp", 'shaders = 1\n\nshader0 = shaders/crt-potato/shader-files/crt-potato.glsl\nfilter_linear0 = false\nscale_type0 = viewport\n'
                              'scale0 = 1.0\nalias0 = "PASS0"\n\ntextures = MASK\nMASK = shaders/crt-potato/resources/crt-potato-thin.png\nMASK_linear = false\nMASK_wrap_mode = "repeat"\n')

# handheld/sameboy-dmg-response-time.glslp (same keys / values; its two shaders are byte-identical copies of motionblur/response-time and gb-palette)
PRESETS["sameboy-dmg-response-time"] = ("handheld/sameboy-dmg-response-time.glslp", 'shaders = "2"\n\nshader0 = "shaders/sameboy-palettes/response-time.glsl"\nfilter_linear0 = "true"\nwrap_mode0 = "clamp_to_border"\nmipmap_input0 = "false"\nalias0 = ""\nfloat_framebuffer0 = "false"\nsrgb_framebuffer0 = "false"\nscale_type_x0 = "source"\nscale_x0 = "1.000000"\nscale_type_y0 = "source"\nscale_y0 = "1.000000"\nparameters = "response_time"\nresponse_time = "0.333000"\n\nshader1 = "shaders/sameboy-palettes/gb-palette.glsl"\nfilter_linear1 = "true"\nwrap_mode1 = "clamp_to_border"\nmipmap_input1 = "false"\nalias1 = ""\nfloat_framebuffer1 = "false"\nsrgb_framebuffer1 = "false"\nscale_type_x1 = "source"\nscale_x1 = "1.000000"\nscale_type_y1 = "source"\nscale_y1 = "1.000000"\ntextures = "COLOR_PALETTE"\nCOLOR_PALETTE = "shaders/sameboy-palettes/resources/DMG.png"\nCOLOR_PALETTE_linear = "false"\nCOLOR_PALETTE_wrap_mode = "clamp_to_border"\nCOLOR_PALETTE_mipmap = "false"\n')
# handheld/gb-palette-dmg.glslp (same keys / values; synthetic 4-band palette image)
PRESETS["gb-palette-dmg"] = ("handheld/gb-palette-dmg.glslp", 'shaders = 1\nshader0 = shaders/gb-palette/gb-palette.glsl\n\nscale_type0 = source\nfilter_linear0 = false\n\n'
                             'textures = COLOR_PALETTE\nCOLOR_PALETTE = shaders/gb-palette/resources/palette-dmg.png\nCOLOR_PALETTE_linear = false\n')

# reshade/lut.glslp and reshade/gba.glslp (same keys / values; the LUT images are small synthetic grades, tests/golden/lut_color*_synthetic.png)
PRESETS["reshade-lut"] = ("reshade/lut.glslp", 'shaders = 1\n\nshader0 = shaders/LUT/LUT.glsl\n\ntextures = SamplerLUT\n\nSamplerLUT = shaders/LUT/16.png\nSamplerLUT_linear = true\n')
PRESETS["reshade-gba"] = ("reshade/gba.glslp", 'shaders = 1\n\nshader0 = shaders/LUT/LUT.glsl\nfilter_linear0 = "false"\nscale_type_x0 = "source"\nscale_x0 = "1.000000"\n'
                          'scale_type_y0 = "source"\nscale_y0 = "1.000000"\n\ntextures = SamplerLUT\n\nSamplerLUT = shaders/LUT/GBA.png\nSamplerLUT_linear = true\n\n'
                          'parameters = "LUT_Size"\nLUT_Size = "32.000000"\n')

# borders/sgb/sgb-crt-geom-1x.glslp and borders/gameboy-player/gameboy-player.glslp (same keys / values; synthetic border image, see above)
PRESETS["sgb-crt-geom-1x"] = ("borders/sgb/sgb-crt-geom-1x.glslp", """shaders = 2

textures = "BORDER"
BORDER = "sgb.png"
BORDER_linear = true

# Pass0: Apply Game Boy Player border
shader0 = "../resources/imgborder-sgb.glsl"
scale_type_x0 = "absolute"
scale_x0 = "256"
scale_type_y0 = "absolute"
scale_y0 = "224"

shader1 = "../../crt/shaders/crt-geom.glsl"

parameters = "box_scale;in_res_x;in_res_y;border_on_top"
box_scale = 1.0
in_res_x = 160.0
in_res_y = 144.0
border_on_top = 0.0
""")
PRESETS["gameboy-player"] = ("borders/gameboy-player/gameboy-player.glslp", """shaders = "1"
shader0 = "../resources/imgborder-gameboy-player.glsl"

scale_type_x0 = "absolute"
scale_x0 = "608"
scale_type_y0 = "absolute"
scale_y0 = "448"

parameters = "box_scale;location;in_res_x;in_res_y"
box_scale = "2.000000"
location = "0.500000"
in_res_x = "240.000000"
in_res_y = "160.000000"

textures = "BORDER"
BORDER = "gameboy-player.png"
""")
# the sgb shader alone, without a parameter block (a one-pass chain of this repository): every parameter can move
PRESETS["imgborder-sgb-bare"] = ("borders/sgb/imgborder-sgb-bare.glslp", 'shaders = 1\nshader0 = "../resources/imgborder-sgb.glsl"\ntextures = "BORDER"\nBORDER = "sgb.png"\nBORDER_linear = true\n')

# handheld/lcd-grid.glslp, and handheld/console-border/gba-3x.glslp (motionblur-simple, gba-color, lcd-grid, border overlay): same
# keys / values as the reference's files
PRESETS["lcd-grid"] = ("handheld/lcd-grid.glslp", 'shaders = 1\n\nshader0 = shaders/lcd-cgwg/lcd-grid.glsl\nfilter_linear0 = false')
PRESETS["gba-3x"] = ("handheld/console-border/gba-3x.glslp", """shaders = "4"

shader0 = "../../motionblur/shaders/motionblur-simple.glsl"
scale_type0 = source
scale0 = 1.0
filter_linear0 = false

shader1 = "../shaders/color/gba-color.glsl"
filter_linear1 = false

shader2 = "../shaders/lcd-cgwg/lcd-grid.glsl"
filter_linear2 = "false"
wrap_mode2 = "clamp_to_border"
scale_type_x2 = "source"
scale_x2 = "3.000000"
scale_type_y2 = "source"
scale_y2 = "3.000000"

shader3 = "shader-files/gb-pass-5.glsl"
filter_linear3 = "true"
wrap_mode3 = "clamp_to_border"

parameters = "SCALE;OUT_X;OUT_Y;GRID_STRENGTH;mixfactor"
GRID_STRENGTH = "0.150000"
SCALE = "1.0"
OUT_X = "2400.0"
OUT_Y = "1200.0"
mixfactor = "0.50"

textures = "BORDER"
BORDER = "resources/gba-border-square-4x.png"
BORDER_linear = "true"
BORDER_wrap_mode = "clamp_to_border"
BORDER_mipmap = "false"
""")

# handheld/agb001.glslp and agb001-gba-color-motionblur.glslp: same keys / values as the reference's files
PRESETS["agb001"] = ("handheld/agb001.glslp", 'shaders = 2\n\nshader0 = shaders/mgba/agb001.glsl\nfilter_linear0 = false\nscale_type0 = source\nscale0 = 4.0\n\n'
                     'shader1 = ../stock.glsl\nfilter_linear1 = true\nscale_type1 = viewport\n')
PRESETS["agb001-gba-color-motionblur"] = ("handheld/agb001-gba-color-motionblur.glslp",
                                          'shaders = 3\n\nshader0 = ../motionblur/shaders/response-time.glsl\nfilter_linear0 = false\nscale_type0 = source\nscale0 = 1.0\n\n'
                                          'shader1 = shaders/mgba/agb001.glsl\nfilter_linear1 = false\nscale_type1 = source\nscale1 = 4.0\n\n'
                                          'shader2 = shaders/color/gba-color.glsl\nfilter_linear2 = true\nscale_type2 = viewport\n')

# handheld/console-border/: two of the 75 presets whose last pass lays a border image over the frame (gb-pass-5.glsl) - same passes,
# keys and values as the reference's gba-lcd-grid-v2-3x.glslp and gbc-retro-v2-2x.glslp; the border PNG here is a small synthetic
# image (tests/golden/lut_border_synthetic.png), the reference's are 1-2 MB artwork
PRESETS["gba-lcd-grid-v2-3x"] = ("handheld/console-border/gba-lcd-grid-v2-3x.glslp", """shaders = 4
shader0 = ../../motionblur/shaders/response-time.glsl
shader1 = ../shaders/lcd-cgwg/lcd-grid-v2.glsl
shader2 = ../shaders/color/gba-color.glsl
shader3 = shader-files/gb-pass-5.glsl

scale_type0 = source
scale0 = 1

scale_type1 = source
scale1 = 3

scale_type2 = source
scale2 = 1

filter_linear0 = false
filter_linear1 = false
filter_linear2 = false
filter_linear3 = true

textures = BORDER

BORDER = resources/gba-border-square-4x.png
BORDER_linear = true

parameters = "SCALE;OUT_X;OUT_Y;RSUBPIX_R;RSUBPIX_G;RSUBPIX_B;GSUBPIX_R;GSUBPIX_G;GSUBPIX_B;BSUBPIX_R;BSUBPIX_G;BSUBPIX_B;gain;gamma;blacklevel;ambient;BGR"
SCALE = "1.0"
OUT_X = "2400.0"
OUT_Y = "1200.0"
RSUBPIX_R = "0.750000"
RSUBPIX_G = "0.000000"
RSUBPIX_B = "0.000000"
GSUBPIX_R = "0.000000"
GSUBPIX_G = "0.750000"
GSUBPIX_B = "0.000000"
BSUBPIX_R = "0.000000"
BSUBPIX_G = "0.000000"
BSUBPIX_B = "0.750000"
gain = "1.500000"
gamma = "2.200000"
blacklevel = "0.000000"
ambient = "0.000000"
BGR = "1.000000"
""")
PRESETS["gbc-retro-v2-2x"] = ("handheld/console-border/gbc-retro-v2-2x.glslp", """shaders = 4
shader0 = ../../motionblur/shaders/response-time.glsl
shader1 = ../shaders/color/gbc-color.glsl
shader2 = ../shaders/retro-v2.glsl
shader3 = shader-files/gb-pass-5.glsl

scale_type0 = source
scale0 = 1

scale_type1 = source
scale1 = 1

scale_type2 = source
scale2 = 2

filter_linear0 = false
filter_linear1 = false
filter_linear2 = false
filter_linear3 = true

textures = BORDER

BORDER = resources/color-border-square-4x.png
BORDER_linear = true

parameters = "SCALE;OUT_X;OUT_Y;RETRO_PIXEL_SIZE"
SCALE = "1.0"
OUT_X = "1600.0"
OUT_Y = "800.0"
RETRO_PIXEL_SIZE = "0.55"
""")

# crt-royale with an RGBA32F last target: the last pass's floats as computed, for the curved-geometry / tex2Daa form
PRESETS["crt-royale-f32-last"] = ("crt/crt-royale-f32-last.glslp", PRESETS["crt-royale"][1] + 'float_framebuffer11 = "true"\n')
PRESETS["crt-royale-ntsc-256px-svideo"] = ("crt/crt-royale-ntsc-256px-svideo.glslp", _royale_ntsc("svideo-3phase", "3phase", 1536))
PRESETS["crt-royale-ntsc-320px-composite"] = ("crt/crt-royale-ntsc-320px-composite.glslp", _royale_ntsc("composite-2phase", "2phase", 1280))

# files copied next to a preset: name -> (preset key, relative path below the preset dir, source under tests/golden)
ASSETS = {"mask_slot_small_64.png": ("crt-royale", "shaders/crt-royale/mask_slot_small_64.png", "lut_mask_slot_small_64.png"),
          "mask_slot_small_64.png#fake-bloom": ("crt-royale-fake-bloom", "shaders/crt-royale/mask_slot_small_64.png", "lut_mask_slot_small_64.png"),
          "mask_slot_small_64.png#ntsc-256": ("crt-royale-ntsc-256px-svideo", "shaders/crt-royale/mask_slot_small_64.png", "lut_mask_slot_small_64.png"),
          "gba-border": ("gba-lcd-grid-v2-3x", "resources/gba-border-square-4x.png", "lut_border_synthetic.png"),
          "gba-border#gba-3x": ("gba-3x", "resources/gba-border-square-4x.png", "lut_border_synthetic.png"),
          "lut16": ("reshade-lut", "shaders/LUT/16.png", "lut_color16_synthetic.png"),
          "lut32": ("reshade-gba", "shaders/LUT/GBA.png", "lut_color32_synthetic.png"),
          "ngpc-border": ("ngpc-3x", "resources/ngpc-border-square-4x.png", "lut_border_synthetic.png"),
          "gb-palette": ("gb-palette-dmg", "shaders/gb-palette/resources/palette-dmg.png", "lut_palette_synthetic.png"),
          "potato-mask": ("crt-potato-cool", "shaders/crt-potato/resources/crt-potato-thin.png", "lut_potato_mask_synthetic.png"),
          "sameboy-palette": ("sameboy-dmg-response-time", "shaders/sameboy-palettes/resources/DMG.png", "lut_palette_synthetic.png"),
          "sgb-border": ("sgb-crt-geom-1x", "sgb.png", "lut_border_synthetic.png"),
          "gbp-border": ("gameboy-player", "gameboy-player.png", "lut_border_synthetic.png"),
          "color-border": ("gbc-retro-v2-2x", "resources/color-border-square-4x.png", "lut_border_synthetic.png"),
          "mask_slot_small_64.png#ntsc-320": ("crt-royale-ntsc-320px-composite", "shaders/crt-royale/mask_slot_small_64.png", "lut_mask_slot_small_64.png")}

ROYALE_LAST_PARAMS = [
    ("crt_gamma", 2.5), ("lcd_gamma", 2.2), ("levels_contrast", 1.0), ("halation_weight", 0.0),
    ("diffusion_weight", 0.075), ("bloom_underestimate_levels", 0.8), ("bloom_excess", 0.0), ("beam_min_sigma", 0.02),
    ("beam_max_sigma", 0.3), ("beam_spot_power", 0.33), ("beam_min_shape", 2.0), ("beam_max_shape", 4.0),
    ("beam_shape_power", 0.25), ("beam_horiz_filter", 0.0), ("beam_horiz_sigma", 0.35),
    ("beam_horiz_linear_rgb_weight", 1.0), ("convergence_offset_x_r", 0.0), ("convergence_offset_x_g", 0.0),
    ("convergence_offset_x_b", 0.0), ("convergence_offset_y_r", 0.0), ("convergence_offset_y_g", 0.0),
    ("convergence_offset_y_b", 0.0), ("mask_type", 1.0), ("mask_sample_mode_desired", 0.0),
    ("mask_specify_num_triads", 0.0), ("mask_triad_size_desired", 3.0), ("mask_num_triads_desired", 480.0),
    ("aa_subpixel_r_offset_y_runtime", 0.0), ("aa_cubic_c", 0.5), ("aa_gauss_sigma", 0.5), ("geom_mode_runtime", 0.0),
    ("geom_radius", 2.0), ("geom_view_dist", 2.0), ("geom_tilt_angle_x", 0.0), ("geom_tilt_angle_y", 0.0),
    ("geom_aspect_ratio_x", 432.0), ("geom_aspect_ratio_y", 329.0), ("geom_overscan_x", 1.0), ("geom_overscan_y", 1.0),
    ("border_size", 0.015), ("border_darkness", 2.0), ("border_compress", 2.5), ("interlace_bff", 0.0),
    ("interlace_1080i", 0.0)]
_R = "crt/shaders/crt-royale/src/crt-royale-"

# shader identity -> oracle pass function, parameter order (name, default), extra samplers
SHADERS = {
    "stock.glsl": {"oracle": "stock", "params": [], "samplers": []},
    "scanlines/shaders/scanline.glsl": {
        "oracle": "scanline",
        "params": [("SCANLINE_BASE_BRIGHTNESS", 0.95), ("SCANLINE_SINE_COMP_A", 0.0), ("SCANLINE_SINE_COMP_B", 0.25),
                   ("size", 1.0)],
        "samplers": []},
    "crt/shaders/crt-pi.glsl": {
        "oracle": "crt_pi",
        "params": [("CURVATURE_X", 0.10), ("CURVATURE_Y", 0.15), ("MASK_BRIGHTNESS", 0.70), ("SCANLINE_WEIGHT", 6.0),
                   ("SCANLINE_GAP_BRIGHTNESS", 0.12), ("BLOOM_FACTOR", 1.5), ("INPUT_GAMMA", 2.4), ("OUTPUT_GAMMA", 2.2)],
        "samplers": []},
    # size_independent: reads no size uniform, so the history re-draw (which keeps pass 0's stale size
    # uniforms, reference ShaderEngine.cpp:1805-1834) is well defined for any geometry
    "conformance/feedback-persist.glsl": {"oracle": "feedback_persist", "params": [("PERSIST", 0.8)],
                                          "samplers": ["PassFeedback0", "PassFeedback1"]},
    # stale_size_uniforms: reads them, and its oracle pass / kernel take the stale values of the history re-draw separately
    "conformance/history-size.glsl": {"oracle": "history_size", "params": [("HS_MIX", 0.3)], "samplers": ["PrevTexture", "Prev1Texture"],
                                      "stale_size_uniforms": True},
    "motionblur/shaders/mix_frames.glsl": {"oracle": "mix_frames", "params": [], "samplers": ["PrevTexture"],
                                           "size_independent": True},
    "motionblur/shaders/motionblur-simple.glsl": {"oracle": "motionblur_simple", "params": [], "size_independent": True,
                                                  "samplers": ["Prev6Texture", "Prev5Texture", "Prev4Texture", "Prev3Texture", "Prev2Texture",
                                                               "Prev1Texture", "PrevTexture"]},
    "motionblur/shaders/braid-rewind.glsl": {"oracle": "braid_rewind", "params": [], "size_independent": True,
                                             "samplers": ["Prev6Texture", "Prev5Texture", "Prev4Texture", "Prev3Texture", "Prev2Texture",
                                                          "Prev1Texture", "PrevTexture"]},
    "motionblur/shaders/response-time.glsl": {"oracle": "response_time", "params": [("response_time", 0.333)], "size_independent": True,
                                              "samplers": ["PrevTexture", "Prev1Texture", "Prev2Texture", "Prev3Texture", "Prev4Texture",
                                                           "Prev5Texture", "Prev6Texture"]},
    "handheld/shaders/color/gba-color.glsl": {"oracle": "gba_color", "params": [('darken_screen', 1.0)], "samplers": [], "size_independent": True},
    "handheld/shaders/color/gbc-color.glsl": {"oracle": "gbc_color", "params": [('lighten_screen', 1.0)], "samplers": [], "size_independent": True},
    "handheld/shaders/color/gbc-gambatte-color.glsl": {"oracle": "gbc_gambatte_color", "params": [], "samplers": [], "size_independent": True},
    "handheld/shaders/color/nds-color.glsl": {"oracle": "nds_color", "params": [], "samplers": [], "size_independent": True},
    "handheld/shaders/color/palm-color.glsl": {"oracle": "palm_color", "params": [], "samplers": [], "size_independent": True},
    "handheld/shaders/color/psp-color.glsl": {"oracle": "psp_color", "params": [], "samplers": [], "size_independent": True},
    "handheld/shaders/color/vba-color.glsl": {"oracle": "vba_color", "params": [('darken_screen', 1.0)], "samplers": [], "size_independent": True},
    "borders/resources/imgborder-sgb.glsl": {"oracle": "imgborder", "samplers": ["BORDER"], "params": [("box_scale", 1.0), ("location_x", 0.5), ("location_y", 0.5), ("in_res_x", 160.0), ("in_res_y", 144.0), ("border_on_top", 0.0), ("border_zoom_x", 1.0), ("border_zoom_y", 1.0), ("OS_MASK_TOP", 0.0), ("OS_MASK_BOTTOM", 0.0), ("OS_MASK_LEFT", 0.0), ("OS_MASK_RIGHT", 0.0)]},
    "borders/resources/imgborder-gameboy-player.glsl": {"oracle": "imgborder", "samplers": ["BORDER"], "params": [("box_scale", 2.0), ("location_x", 0.5), ("location_y", 0.5), ("in_res_x", 240.0), ("in_res_y", 160.0), ("border_on_top", 0.0), ("border_zoom_x", 1.0), ("border_zoom_y", 1.0), ("OS_MASK_TOP", 0.0), ("OS_MASK_BOTTOM", 0.0), ("OS_MASK_LEFT", 0.0), ("OS_MASK_RIGHT", 0.0)]},
    "borders/resources/imgborder-sgba.glsl": {"oracle": "imgborder", "samplers": ["BORDER"], "params": [("box_scale", 1.0), ("location_x", 0.5), ("location_y", 0.5), ("in_res_x", 240.0), ("in_res_y", 160.0), ("border_on_top", 0.0), ("border_zoom_x", 1.0), ("border_zoom_y", 1.0), ("OS_MASK_TOP", 0.0), ("OS_MASK_BOTTOM", 0.0), ("OS_MASK_LEFT", 0.0), ("OS_MASK_RIGHT", 0.0)]},
    "handheld/console-border/shader-files/border.glsl": {"oracle": "console_border", "samplers": ["BORDER"],
                                                         "params": [("box_scale", 4.0), ("location_x", 0.5), ("location_y", 0.5), ("in_res_x", 320.0),
                                                                    ("in_res_y", 240.0), ("border_on_top", 1.0), ("border_zoom_x", 1.0), ("border_zoom_y", 1.0)]},
    "handheld/shaders/sameboy-palettes/gb-palette.glsl": {"oracle": "gb_palette", "samplers": ["COLOR_PALETTE"], "params": [], "size_independent": True},
    "handheld/shaders/sameboy-palettes/response-time.glsl": {"oracle": "response_time", "params": [("response_time", 0.333)], "size_independent": True,
                                                             "samplers": ["PrevTexture", "Prev1Texture", "Prev2Texture", "Prev3Texture", "Prev4Texture",
                                                                          "Prev5Texture", "Prev6Texture"]},
    "handheld/shaders/gb-palette/gb-palette.glsl": {"oracle": "gb_palette", "samplers": ["COLOR_PALETTE"], "params": [], "size_independent": True},
    "crt/shaders/crt-potato/shader-files/crt-potato.glsl": {"oracle": "crt_potato", "samplers": ["MASK"], "params": []},
    "stereoscopic-3d/shaders/side-by-side-simple.glsl": {"oracle": "side_by_side", "samplers": [], "params": [('eye_sep', 0.30000001192092896), ('y_loc', 0.25), ('BOTH', 0.5099999904632568), ('ana_zoom', 0.75), ('WIDTH', 3.049999952316284), ('HEIGHT', 2.0), ('warpX', 0.10000000149011612), ('warpY', 0.10000000149011612), ('pulfrich', 0.0)]},
    "handheld/shaders/sameboy-lcd.glsl": {"oracle": "sameboy_lcd", "samplers": [], "params": [('COLOR_LOW', 0.800000011920929), ('COLOR_HIGH', 1.0), ('SCANLINE_DEPTH', 0.10000000149011612)]},
    "crt/shaders/crt-consumer.glsl": {"oracle": "crt_consumer", "samplers": [], "params": [('blurx', 0.25), ('blury', -0.15000000596046448), ('warpx', 0.029999999329447746), ('warpy', 0.03999999910593033), ('corner', 0.009999999776482582), ('smoothness', 400.0), ('scanlow', 6.0), ('scanhigh', 8.0), ('beamlow', 1.350000023841858), ('beamhigh', 1.0499999523162842), ('brightboost1', 1.100000023841858), ('brightboost2', 1.0499999523162842), ('Shadowmask', 7.0), ('masksize', 1.0), ('MaskDark', 0.5), ('MaskLight', 1.5), ('slotmask', 0.0), ('slotwidth', 2.0), ('double_slot', 1.0), ('slotms', 1.0), ('GAMMA_IN', 2.5), ('GAMMA_OUT', 2.200000047683716), ('glow', 0.05000000074505806), ('Size', 1.0), ('sat', 1.100000023841858), ('contrast', 1.0), ('nois', 0.0), ('WP', 0.0), ('inter', 1.0), ('vignette', 1.0), ('vpower', 0.20000000298023224), ('vstr', 40.0), ('alloff', 0.0)]},
    "anti-aliasing/shaders/reverse-aa.glsl": {"oracle": "reverse_aa", "samplers": [], "params": [('REVERSEAA_SHARPNESS', 2.0)]},
    "anti-aliasing/shaders/advanced-aa.glsl": {"oracle": "advanced_aa", "samplers": [], "params": [("AA_RESOLUTION_X", 0.0), ("AA_RESOLUTION_Y", 0.0)]},
    "crt/shaders/crt-lottes.glsl": {"oracle": "crt_lottes", "samplers": [], "params": [('hardScan', -8.0), ('hardPix', -3.0), ('warpX', 0.03099999949336052), ('warpY', 0.04100000113248825), ('maskDark', 0.5), ('maskLight', 1.5), ('scaleInLinearGamma', 1.0), ('shadowMask', 3.0), ('brightBoost', 1.0), ('hardBloomPix', -1.5), ('hardBloomScan', -2.0), ('bloomAmount', 0.15000000596046448), ('shape', 2.0)]},
    "crt/shaders/fakelottes.glsl": {"oracle": "fakelottes", "samplers": [], "params": [('shadowMask', 1.0), ('SCANLINE_SINE_COMP_B', 0.4000000059604645), ('warpX', 0.03099999949336052), ('warpY', 0.04100000113248825), ('maskDark', 0.5), ('maskLight', 1.5), ('crt_gamma', 2.5), ('monitor_gamma', 2.200000047683716), ('SCANLINE_SINE_COMP_A', 0.0), ('SCANLINE_BASE_BRIGHTNESS', 0.949999988079071)]},
    "windowed/shaders/jinc2-sharper.glsl": {"oracle": "jinc2_sharper", "samplers": [], "params": []},
    "misc/interlacing.glsl": {"oracle": "interlacing", "samplers": [], "params": [("percent", 0.0), ("enable_480i", 1.0), ("top_field_first", 0.0)]},
    "crt/shaders/tvout-tweaks.glsl": {"oracle": "tvout_tweaks", "samplers": [], "params": [('TVOUT_RESOLUTION', 256.0), ('TVOUT_COMPOSITE_CONNECTION', 0.0), ('TVOUT_TV_COLOR_LEVELS', 0.0), ('TVOUT_RESOLUTION_Y', 256.0), ('TVOUT_RESOLUTION_I', 83.19999694824219), ('TVOUT_RESOLUTION_Q', 25.600000381469727)]},
    "misc/image-adjustment.glsl": {"oracle": "image_adjustment", "samplers": [], "params": [('ia_target_gamma', 2.2), ('ia_monitor_gamma', 2.2), ('ia_overscan_percent_x', 0.0), ('ia_overscan_percent_y', 0.0), ('ia_saturation', 1.0), ('ia_contrast', 1.0), ('ia_luminance', 1.0), ('ia_black_level', 0.0), ('ia_bright_boost', 0.0), ('ia_R', 1.0), ('ia_G', 1.0), ('ia_B', 1.0), ('ia_ZOOM', 1.0), ('ia_XPOS', 0.0), ('ia_YPOS', 0.0), ('ia_TOPMASK', 0.0), ('ia_BOTMASK', 0.0), ('ia_LMASK', 0.0), ('ia_RMASK', 0.0), ('ia_GRAIN_STR', 0.0), ('ia_SHARPEN', 0.0), ('ia_FLIP_HORZ', 0.0), ('ia_FLIP_VERT', 0.0)]},
    "ntsc/shaders/ntsc-gauss-pass.glsl": {"oracle": "ntsc_gauss", "samplers": [], "params": [("NTSC_CRT_GAMMA", 2.5), ("NTSC_DISPLAY_GAMMA", 2.1)]},
    "ntsc/shaders/ntsc-stock.glsl": {"oracle": "stock", "params": [], "samplers": [], "size_independent": True},
    "reshade/shaders/LUT/LUT.glsl": {"oracle": "lut", "samplers": ["SamplerLUT"], "params": [("LUT_Size", 16.0)], "size_independent": True},
    "handheld/console-border/shader-files/gb-pass-5.glsl": {"oracle": "gb_pass_5", "samplers": ["BORDER"],
                                                            "params": [("SCALE", 0.6667), ("OUT_X", 1600.0), ("OUT_Y", 800.0)]},
    "handheld/shaders/mgba/agb001.glsl": {"oracle": "agb001", "samplers": [], "params": []},
    "handheld/shaders/retro-v2.glsl": {"oracle": "retro_v2", "samplers": [], "params": [("RETRO_PIXEL_SIZE", 0.84)]},
    "handheld/shaders/lcd-cgwg/lcd-grid.glsl": {"oracle": "lcd_grid", "samplers": [], "params": [("GRID_STRENGTH", 0.05), ("gamma", 2.2)]},
    "handheld/shaders/lcd-cgwg/lcd-grid-v2.glsl": {"oracle": "lcd_grid_v2", "samplers": [],
                                                   "params": [("RSUBPIX_R", 1.0), ("RSUBPIX_G", 0.0), ("RSUBPIX_B", 0.0), ("GSUBPIX_R", 0.0), ("GSUBPIX_G", 1.0),
                                                              ("GSUBPIX_B", 0.0), ("BSUBPIX_R", 0.0), ("BSUBPIX_G", 0.0), ("BSUBPIX_B", 1.0), ("gain", 1.0),
                                                              ("gamma", 3.0), ("outgamma", 2.2), ("blacklevel", 0.05), ("ambient", 0.0), ("BGR", 0.0)]},
    "stereoscopic-3d/shaders/shutter-3d.glsl": {"oracle": "shutter_3d", "size_independent": True, "samplers": ["PrevTexture"],
                                                "params": [("ZOOM", 1.0), ("vert_pos", 0.0), ("horz_pos", 0.0), ("separation", 0.0), ("flicker", 0.0),
                                                           ("height_mod", 1.0), ("swap_eye", 0.0)]},
    "misc/anti-flicker.glsl": {"oracle": "anti_flicker", "size_independent": True, "samplers": ["PrevTexture", "Prev1Texture"],
                               "params": [("lum_diff_thresh", 0.5)]},
    "motionblur/shaders/mix_frames_smart.glsl": {"oracle": "mix_frames_smart", "params": [("DEFLICKER_EMPHASIS", 0.0)], "size_independent": True,
                                                 "samplers": ["PrevTexture", "Prev1Texture", "Prev2Texture", "Prev3Texture", "Prev4Texture"]},
    **{"ntsc/shaders/ntsc-pass1-%s.glsl" % n: {"oracle": "ntsc_pass1_" + n.replace("-", "_"), "params": [], "samplers": []}
       for n in ("svideo-3phase", "composite-3phase", "svideo-2phase", "composite-2phase")},
    **{"ntsc/shaders/ntsc-pass2-%s.glsl" % n: {"oracle": "ntsc_pass2_" + n.replace("-", "_"), "params": [], "samplers": []}
       for n in ("3phase-gamma", "3phase-linear", "3phase", "2phase-gamma", "2phase-linear", "2phase")},
    "xbr/shaders/xbr-lv3.glsl": {
        "oracle": "xbr_lv3",
        "params": [("XBR_Y_WEIGHT", 48.0), ("XBR_EQ_THRESHOLD", 10.0), ("XBR_EQ_THRESHOLD2", 2.0),
                   ("XBR_LV2_COEFFICIENT", 2.0), ("corner_type", 3.0)],
        "samplers": []},
    "crt/shaders/crt-easymode.glsl": {
        "oracle": "crt_easymode",
        "params": [("SHARPNESS_H", 0.5), ("SHARPNESS_V", 1.0), ("MASK_STRENGTH", 0.3), ("MASK_DOT_WIDTH", 1.0), ("MASK_DOT_HEIGHT", 1.0),
                   ("MASK_STAGGER", 0.0), ("MASK_SIZE", 1.0), ("SCANLINE_STRENGTH", 1.0), ("SCANLINE_BEAM_WIDTH_MIN", 1.5),
                   ("SCANLINE_BEAM_WIDTH_MAX", 1.5), ("SCANLINE_BRIGHT_MIN", 0.35), ("SCANLINE_BRIGHT_MAX", 0.65),
                   ("SCANLINE_CUTOFF", 400.0), ("GAMMA_INPUT", 2.0), ("GAMMA_OUTPUT", 1.8), ("BRIGHT_BOOST", 1.2), ("DILATION", 1.0)],
        "samplers": []},
    "crt/shaders/crt-geom.glsl": {
        "oracle": "crt_geom",
        "params": [("CRTgamma", 2.4), ("monitorgamma", 2.2), ("d", 1.6), ("CURVATURE", 1.0), ("R", 2.0), ("cornersize", 0.03),
                   ("cornersmooth", 1000.0), ("x_tilt", 0.0), ("y_tilt", 0.0), ("overscan_x", 100.0), ("overscan_y", 100.0),
                   ("DOTMASK", 0.3), ("SHARPER", 1.0), ("scanline_weight", 0.3), ("lum", 0.0), ("interlace_detect", 1.0),
                   ("SATURATION", 1.0)],
        "samplers": []},
    "crt/shaders/crt-nes-mini.glsl": {"oracle": "crt_nes_mini",
                                      "params": [("SCANTHICK", 2.0), ("INTENSITY", 0.15), ("BRIGHTBOOST", 0.15)], "samplers": []},
    "scalenx/shaders/epx.glsl": {"oracle": "epx", "params": [], "samplers": []},
    "scalefx/shaders/scalefx-pass0.glsl": {"oracle": "scalefx0", "params": [], "samplers": []},
    "scalefx/shaders/scalefx-pass1.glsl": {"oracle": "scalefx1", "params": [("SFX_CLR", 0.5), ("SFX_SAA", 1.0)], "samplers": []},
    "scalefx/shaders/scalefx-pass2.glsl": {"oracle": "scalefx2", "params": [], "samplers": ["PassPrev2Texture"]},
    "scalefx/shaders/scalefx-pass3.glsl": {"oracle": "scalefx3", "params": [("SFX_SCN", 1.0)], "samplers": []},
    "scalefx/shaders/scalefx-pass4.glsl": {"oracle": "scalefx4", "params": [], "samplers": ["PassPrev5Texture"]},
    "handheld/shaders/lcd1x.glsl": {"oracle": "lcd1x", "params": [("BRIGHTEN_SCANLINES", 16.0), ("BRIGHTEN_LCD", 4.0)], "samplers": []},
    "handheld/shaders/lcd3x.glsl": {"oracle": "lcd3x", "params": [("brighten_scanlines", 16.0), ("brighten_lcd", 4.0)], "samplers": []},
    "dithering/shaders/bayer-matrix-dithering.glsl": {"oracle": "bayer", "params": [("animate", 0.0), ("dither_size", 0.0)], "samplers": []},
    "interpolation/shaders/quilez.glsl": {"oracle": "quilez", "params": [], "samplers": []},
    "interpolation/shaders/smootheststep.glsl": {"oracle": "smootheststep", "params": [], "samplers": []},
    "interpolation/shaders/sharp-bilinear.glsl": {"oracle": "sharp_bilinear",
                                                  "params": [("SHARP_BILINEAR_PRE_SCALE", 4.0), ("AUTO_PRESCALE", 1.0)], "samplers": []},
    "crt/shaders/zfast_crt.glsl": {
        "oracle": "zfast_crt",
        "params": [("BLURSCALEX", 0.30), ("LOWLUMSCAN", 6.0), ("HILUMSCAN", 8.0), ("BRIGHTBOOST", 1.25), ("MASK_DARK", 0.25),
                   ("MASK_FADE", 0.8)],
        "samplers": []},
    "crt/shaders/glow/linearize.glsl": {"oracle": "glow_linearize", "params": [("INPUT_GAMMA", 2.4)], "samplers": []},
    "crt/shaders/hyllian/crt-hyllian-glow/crt-hyllian-glow.glsl": {
        "oracle": "crt_hyllian_glow",
        "params": [("BEAM_PROFILE", 0.0), ("BEAM_MIN_WIDTH", 0.86), ("BEAM_MAX_WIDTH", 1.0), ("SCANLINES_STRENGTH", 0.58),
                   ("COLOR_BOOST", 1.25), ("HFILTER_SHARPNESS", 1.0), ("CRT_ANTI_RINGING", 1.0), ("InputGamma", 2.4),
                   ("OutputGamma", 2.2), ("VSCANLINES", 0.0)],
        "samplers": []},
    "crt/shaders/glow/threshold.glsl": {"oracle": "glow_threshold", "params": [("GLOW_WHITEPOINT", 1.0), ("GLOW_ROLLOFF", 3.0)],
                                         "samplers": []},
    "crt/shaders/glow/blur_horiz.glsl": {"oracle": "glow_blur_h", "params": [], "samplers": []},
    "crt/shaders/glow/blur_vert.glsl": {"oracle": "glow_blur_v", "params": [], "samplers": []},
    "crt/shaders/hyllian/crt-hyllian-glow/resolve2.glsl": {
        "oracle": "hyllian_resolve2",
        "params": [("BLOOM_STRENGTH", 0.45), ("OUTPUT_GAMMA", 2.2), ("PHOSPHOR_LAYOUT", 4.0), ("MASK_INTENSITY", 0.5)],
        "samplers": ["PassPrev4Texture"]},
    "xbr/shaders/xbr-lv2.glsl": {
        "oracle": "xbr_lv2",
        "params": [("XBR_SCALE", 3.0), ("XBR_Y_WEIGHT", 48.0), ("XBR_EQ_THRESHOLD", 15.0), ("XBR_LV1_COEFFICIENT", 0.5),
                   ("XBR_LV2_COEFFICIENT", 2.0), ("small_details", 0.0)],
        "samplers": []},
    _R + "first-pass-linearize-crt-gamma-bob-fields.glsl": {"oracle": "royale_first", "params": [], "samplers": []},
    _R + "scanlines-vertical-interlacing.glsl": {"oracle": "royale_scan_v", "params": [], "samplers": []},
    _R + "bloom-approx.glsl": {"oracle": "royale_bloom_approx", "params": [], "samplers": ["PassPrev2Texture"]},
    "blurs/blur9fast-vertical.glsl": {"oracle": "blur9_v", "params": [], "samplers": []},
    "blurs/blur9fast-horizontal.glsl": {"oracle": "blur9_h", "params": [], "samplers": []},
    _R + "mask-resize-vertical.glsl": {"oracle": "royale_mask_v", "params": [], "samplers": ["mask_slot_texture_small"]},
    _R + "mask-resize-horizontal.glsl": {"oracle": "royale_mask_h", "params": [], "samplers": []},
    _R + "scanlines-horizontal-apply-mask.glsl": {"oracle": "royale_scan_h", "params": [],
                                                   "samplers": ["PassPrev6Texture", "PassPrev3Texture"]},
    _R + "bloom-approx-fake-bloom.glsl": {"oracle": "royale_bloom_approx", "params": [], "samplers": ["PassPrev2Texture"]},
    _R + "scanlines-horizontal-apply-mask-fake-bloom.glsl": {"oracle": "royale_scan_h_fake", "params": [],
                                                              "samplers": ["PassPrev6Texture", "PassPrev5Texture", "PassPrev3Texture"]},
    _R + "brightpass.glsl": {"oracle": "royale_brightpass", "params": [], "samplers": ["PassPrev4Texture"]},
    _R + "bloom-vertical.glsl": {"oracle": "royale_bloom_v", "params": [], "samplers": []},
    _R + "bloom-horizontal-reconstitute.glsl": {"oracle": "royale_bloom_h", "params": [],
                                                 "samplers": ["PassPrev3Texture", "PassPrev2Texture", "PassPrev6Texture"]},
    _R + "geometry-aa-last-pass.glsl": {"oracle": "royale_last", "params": ROYALE_LAST_PARAMS, "samplers": []},
}


def write_tree(root):
    out = {}
    for name, (rel, text) in PRESETS.items():
        p = os.path.join(root, "shaders_glsl", rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(text)
        out[name] = p
    import shutil
    here = os.path.dirname(os.path.abspath(__file__))
    for _, (key, rel, src) in ASSETS.items():
        dst = os.path.join(os.path.dirname(out[key]), rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copy(os.path.join(here, "golden", src), dst)
    return out


def identity(shader_path):
    """Registry key of a resolved shader path: the part below shaders_glsl/, else the longest
    known identity that the path ends with (or that ends with the path's last two components)."""
    k = shader_path.rfind("shaders_glsl/")
    if k >= 0:
        return shader_path[k + len("shaders_glsl/"):]
    tail = "/".join(shader_path.split("/")[-2:])
    for ident in SHADERS:
        if shader_path.endswith("/" + ident) or ident.endswith("/" + tail) or ident == tail:
            return ident
    return shader_path
