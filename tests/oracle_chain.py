"""Runs a whole preset through the oracle (TEST INFRASTRUCTURE).

Own restatement, in Python, of the pass-graph rules the product implements in C++
(sizes: reference ShaderEngine.cpp:856-894 and :1881-1910; target formats :2882-2890; sampler
state persisting on the consumed texture :1008-1036; PassPrev / alias / LUT binding
:1163-1415), so that the engine's plumbing is checked by something other than itself.
"""
import math

import numpy as np

import chain_specs
from oracle_lib import Tex, run_pass


def _cround(x):  # std::round: halves away from zero
    return int(math.floor(x + 0.5)) if x >= 0 else -int(math.floor(-x + 0.5))


def calc_scale(src, typ, scale, vp):
    f32 = np.float32
    if typ == "" or typ == "source":
        s = f32(scale) if scale != 0 else f32(1)
        return _cround(float(f32(src) * s))
    if typ == "viewport":
        s = f32(scale) if scale != 0 else f32(1)
        return _cround(float(f32(vp) * s))
    if typ == "absolute":
        return _cround(float(f32(scale)))
    return src


def pass_sizes(passes, w, h, vw, vh):
    out = []
    cw, ch = w, h
    for i, p in enumerate(passes):
        tx, ty, sx, sy = p["stx"], p["sty"], p["sx"], p["sy"]
        last = i == len(passes) - 1
        if last and tx != "viewport" and (tx == "" or (tx == "source" and sx == 1.0)):
            tx, sx = "viewport", 1.0
        if last and ty != "viewport" and (ty == "" or (ty == "source" and sy == 1.0)):
            ty, sy = "viewport", 1.0
        ow, oh = calc_scale(cw, tx, sx, vw), calc_scale(ch, ty, sy, vh)
        out.append((ow, oh))
        cw, ch = ow, oh
    return out


def run_chain(passes, rgb, vw, vh, frame_count=1, luts=None, custom=None, global_params=None, flags=0,
              given=None):
    """passes: list of dicts as produced by the preset dump (shader, filter_linear, wrap,
    alias, float_fb, srgb_fb, stx, sx, sty, sy).  rgb: (h, w, 3) uint8 source frame.
    luts: name -> (rgba array, linear, wrap).  given: optional list of per-pass arrays to feed forward
    instead of the oracle's own outputs (isolates each pass when checking against golden data).
    Returns the list of per-pass outputs."""
    h, w, _ = rgb.shape
    src = np.concatenate([rgb, np.full((h, w, 1), 255, np.uint8)], -1)
    sizes = pass_sizes(passes, w, h, vw, vh)
    fmts = ["f32" if p["float_fb"] else ("srgb8" if p["srgb_fb"] else "rgba8") for p in passes]
    outs = []

    def tex_of_pass(k):
        nxt = passes[k + 1] if k + 1 < len(passes) else {"filter_linear": True, "wrap": "clamp_to_edge"}
        return Tex(given[k] if given is not None else outs[k], fmts[k], nxt["filter_linear"], nxt["wrap"])

    source_tex = Tex(src, "rgbx8", passes[0]["filter_linear"], passes[0]["wrap"])
    cur = source_tex
    for i, p in enumerate(passes):
        spec = chain_specs.SHADERS[chain_specs.identity(p["shader"])]
        extra = []
        for name in spec["samplers"]:
            t = None
            if name.startswith("PassPrev") and name.endswith("Texture"):
                n = int(name[8:-7])
                t = tex_of_pass(i - n) if 1 <= n <= i else source_tex
            elif name == "OrigTexture":
                t = source_tex
            else:
                for k in range(i):
                    if passes[k]["alias"] and passes[k]["alias"] == name:
                        t = tex_of_pass(k)
                if t is None and luts and name in luts:
                    arr, linear, wrap = luts[name]
                    t = Tex(arr, "rgba8", linear, wrap)
            extra.append(t if t is not None else cur)
        params = []
        for name, default in spec["params"]:
            v = default
            if custom and name in custom:
                v = custom[name]
            if global_params and name in global_params:
                v = global_params[name]
            params.append(v)
        ow, oh = sizes[i]
        o = run_pass(spec["oracle"], cur, ow, oh, out_fmt=fmts[i], params=params, frame_count=frame_count,
                     extra=extra, src_w=w, src_h=h, chain=sizes, pass_index=i, vp=(vw, vh), flags=flags)
        outs.append(o)
        cur = tex_of_pass(i)
    return outs
