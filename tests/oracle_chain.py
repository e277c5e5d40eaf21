"""Runs a whole preset through the oracle (TEST INFRASTRUCTURE).

Own restatement, in Python, of the pass-graph rules the product implements in C++
(sizes: reference ShaderEngine.cpp:856-894 and :1881-1910; target formats :2882-2890; sampler
state persisting on the consumed texture :1008-1036; PassPrev / alias / LUT binding
:1163-1415), so that the engine's plumbing is checked by something other than itself.
"""
import math

import numpy as np

import chain_specs
import oracle_lib
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


# ShaderEngine.cpp:2260-2294: uniforms the reference overwrites with fixed values on every draw
HARD_CODED = {"BLURSCALEX": 0.30, "LOWLUMSCAN": 6.0, "HILUMSCAN": 8.0, "BRIGHTBOOST": 1.25, "MASK_DARK": 0.25, "MASK_FADE": 0.8}


class ChainState:
    """What the reference keeps between frames (ShaderEngine.h:140-143 and the GL objects' own state):
    the frame-history ring (newest first, at most 7), the texture bound to every texture unit, and
    the sampler uniforms of pass 0's program (a uniform keeps its value until set again)."""

    def __init__(self):
        self.history = []          # list of (h, w, 4) uint8 arrays
        self.units = {}            # unit -> Tex
        self.pass0_units = {}      # sampler name -> unit, as last set on pass 0's program
        self.frame_count = 0
        # PassFeedback ping-pong (cpp:1285-1347, :1710-1718): per pass the partner texture (array or
        # None), its sampler state, and whether any program asked for it
        self.feedback = {}         # pass index -> array
        self.feedback_state = {}   # pass index -> (linear, wrap)
        self.feedback_enabled = set()


def _history_tex(arr):
    return Tex(arr, "rgba8", True, "clamp_to_edge")   # creation state, never changed (cpp:1757-1761)


def run_chain(passes, rgb, vw, vh, frame_count=1, luts=None, custom=None, global_params=None, flags=0,
              given=None, state=None, force_f32=False, f16_targets=False):
    """passes: list of dicts as produced by the preset dump (shader, filter_linear, wrap,
    alias, float_fb, srgb_fb, stx, sx, sty, sy).  rgb: (h, w, 3) uint8 source frame.
    luts: name -> (rgba array, linear, wrap).  given: optional list of per-pass arrays to feed forward
    instead of the oracle's own outputs (isolates each pass when checking against golden data).
    state: a ChainState carried from frame to frame (frame history); None = stateless.
    force_f32: every pass target is RGBA32F (the float-precision goldens, not the reference's formats).
    f16_targets: float targets are stored as binary16 (the engine's opt-in setFloatTargetFp16): each float pass
    output is rounded to nearest-even binary16 and the passes after it read the widened values.
    Returns the list of per-pass outputs (float targets as float32 holding the rounded values when f16_targets)."""
    if global_params is None:
        global_params = getattr(passes, "globals", None)
    h, w, _ = rgb.shape
    src = np.concatenate([rgb, np.full((h, w, 1), 255, np.uint8)], -1)
    sizes = pass_sizes(passes, w, h, vw, vh)
    fmts = ["f32" if (p["float_fb"] or force_f32) else ("srgb8" if p["srgb_fb"] else "rgba8") for p in passes]
    outs = []
    units = state.units if state is not None else {}

    mip_cache = {}

    def tex_of_pass(k):
        nxt = passes[k + 1] if k + 1 < len(passes) else {"filter_linear": True, "wrap": "clamp_to_edge"}
        # mipmap_input of the consuming pass (ShaderEngine.cpp:1022-1033): the chain is generated when that pass
        # binds its input and stays on the texture (as does the min filter) for later PassPrev reads
        mip = bool(nxt.get("mipmap"))   # LINEAR_MIPMAP_LINEAR with filter_linear, NEAREST_MIPMAP_NEAREST without
        if k not in mip_cache:
            mip_cache[k] = Tex(given[k] if given is not None else outs[k], fmts[k], nxt["filter_linear"], nxt["wrap"], mipmap=mip)
        return mip_cache[k]

    # mipmap_input0: the reference generates the chain on the SOURCE texture too (ShaderEngine.cpp:1019-1031)
    source_tex = Tex(src, "rgbx8", passes[0]["filter_linear"], passes[0]["wrap"], mipmap=bool(passes[0].get("mipmap")))
    cur = source_tex
    pass0_call = None
    for i, p in enumerate(passes):
        spec = chain_specs.SHADERS[chain_specs.identity(p["shader"])]
        declared = spec["samplers"]
        if state is not None and i in state.feedback and state.feedback[i].shape[:2] != (sizes[i][1], sizes[i][0]):
            # the pass's size changed: the reference recreates its framebuffer, deletes the feedback partner and clears
            # feedbackEnabled (cpp:918-933); the partner comes back, empty, when a program next asks for it
            del state.feedback[i]
            state.feedback_state.pop(i, None)
            state.feedback_enabled.discard(i)
        bound = {}                      # sampler name -> unit set in this draw
        unit = 1

        def bind(name, tex):
            nonlocal unit
            units[unit] = tex
            bound[name] = unit
            unit += 1

        # the reference's binding order (ShaderEngine.cpp:1095-1415): history / previous passes,
        # PassPrev<N> beyond pass 0, aliases, (PassFeedback: no registered shader), OrigTexture, LUTs
        if i == 0:
            hist = state.history if state is not None else []
            for k in range(7):
                for name in (("PrevTexture" if k == 0 else "Prev%dTexture" % k), "PassPrev%dTexture" % k):
                    if name in declared:
                        if k < len(hist):
                            bind(name, _history_tex(hist[k]))
                        break
        else:
            for pp in range(i):
                for name in ("PassPrev%dTexture" % (i - pp), "PrevTexture" if pp == 0 else "Prev%dTexture" % pp):
                    if name in declared:
                        bind(name, tex_of_pass(pp))
                        break
            for n in range(i + 1, i + 13):
                if "PassPrev%dTexture" % n in declared:
                    bind("PassPrev%dTexture" % n, source_tex)
            for pp in range(i):
                al = passes[pp]["alias"]
                if al and al in declared:
                    bind(al, tex_of_pass(pp))
        lost_draw = False
        for fp in range(i + 1):     # PassFeedback<fp>, cpp:1285-1347
            name = next((n for n in ("PassFeedback%d" % fp, "PassFeedback%dTexture" % fp) if n in declared), None)
            if name is None:
                continue
            if state is None:
                raise ValueError("PassFeedback needs a ChainState")
            state.feedback_enabled.add(fp)
            if fp not in state.feedback:
                # first sight: the partner texture is created now (zero-filled, creation sampler state).
                # createFramebuffer leaves framebuffer 0 bound (cpp:2931), so THIS pass's draw misses its
                # target, which keeps its clear colour (0,0,0,0)
                fw, fh = sizes[fp]
                state.feedback[fp] = np.zeros((fh, fw, 4), np.float32 if fmts[fp] == "f32" else np.uint8)
                state.feedback_state[fp] = (True, "clamp_to_edge")
                lost_draw = True
            lin, wrap = state.feedback_state[fp]
            bind(name, Tex(state.feedback[fp], fmts[fp], lin, wrap))
        if "OrigTexture" in declared:
            bind("OrigTexture", source_tex)
        for name in sorted(luts or {}):
            arr, linear, wrap = luts[name]
            bind(name, Tex(arr, "rgba8", linear, wrap))
        if i == 0 and state is not None:
            state.pass0_units.update(bound)
        sampler_units = dict(state.pass0_units) if (i == 0 and state is not None) else bound
        extra = [units[sampler_units[name]] if sampler_units.get(name, 0) else cur for name in declared]
        params = []
        for name, default in spec["params"]:
            v = default
            if custom and name in custom:
                v = custom[name]
            if name in HARD_CODED:          # written after the pragma parameters, whatever they were
                v = HARD_CODED[name]
            if global_params and name in global_params:
                v = global_params[name]
            params.append(v)
        ow, oh = sizes[i]
        call = dict(params=params, frame_count=frame_count, src_w=w, src_h=h, chain=sizes, pass_index=i, vp=(vw, vh),
                    flags=flags)
        if i == 0:
            pass0_call = (spec, call)
        if lost_draw:
            o = np.zeros((oh, ow, 4), np.float32 if fmts[i] == "f32" else np.uint8)
        else:
            o = run_pass(spec["oracle"], cur, ow, oh, out_fmt=fmts[i], extra=extra, **call)
        if f16_targets and fmts[i] == "f32":
            o = o.astype(np.float16).astype(np.float32)
        outs.append(o)
        cur = tex_of_pass(i)
    if state is not None and state.feedback_enabled:
        # ping-pong swap (cpp:1710-1718): what was written becomes next frame's "previous"; the texture
        # object keeps the sampler state its consumer set on it
        for fp in sorted(state.feedback_enabled):
            if fp in state.feedback:
                nxt = passes[fp + 1] if fp + 1 < len(passes) else {"filter_linear": True, "wrap": "clamp_to_edge"}
                state.feedback[fp] = outs[fp] if given is None else given[fp]
                state.feedback_state[fp] = (nxt["filter_linear"], nxt["wrap"])
        if any(n.startswith("Prev") or n.startswith("PassPrev0") for n in chain_specs.SHADERS[chain_specs.identity(passes[0]["shader"])]["samplers"]):
            raise NotImplementedError("frame history together with PassFeedback")
        return outs
    if state is not None:
        # history push (cpp:1735-1865): the final output drawn through pass 0's program - its sampler
        # uniforms as they stand, unit 0 = the final output, every other unit as the frame left it -
        # into an RGBA8 texture of the output size
        spec, call = pass0_call
        same = len(passes) == 1 and sizes[-1] == (w, h)
        if not (spec.get("size_independent") or spec.get("stale_size_uniforms") or same):
            raise NotImplementedError("history re-draw with stale size uniforms")
        if spec.get("stale_size_uniforms") and not same:
            # the program's size uniforms still hold what pass 0's own draw set: its input (the source frame) and its output
            call = dict(call, uni=(w, h, sizes[0][0], sizes[0][1]))
        final = cur
        extra = [units[state.pass0_units[n]] if state.pass0_units.get(n, 0) else final for n in spec["samplers"]]
        ow, oh = sizes[-1]
        if len(state.history) == 7:
            # a full ring recycles its OLDEST texture as the target of this very draw (cpp:1762-1768) and clears it to
            # (0, 0, 0, 1) first (:1797-1798) - while the frame's binding still has it on a sampler unit (Prev6Texture):
            # every fragment reads its own, just cleared, texel (llvmpipe renders straight into the texture's memory)
            oldest = state.history[6]
            if oldest.shape[:2] != (oh, ow):
                raise NotImplementedError("history ring recycling a texture of another size (re-allocated: undefined content)")
            cleared = np.zeros_like(oldest)
            cleared[..., 3] = 255
            extra = [Tex(cleared, t.fmt, t.c.linear, {v: k for k, v in oracle_lib.WRAP.items()}.get(t.c.wrap, "clamp_to_edge"))
                     if getattr(t, "arr", None) is oldest else t for t in extra]
        hist = run_pass(spec["oracle"], final, ow, oh, out_fmt="rgba8", extra=extra, **call)
        state.history.insert(0, hist)
        del state.history[7:]
        state.frame_count += 1
    return outs
