"""Opt-in binary16 storage of float_framebuffer targets (rc_engine_set_float_target_fp16; BASELINE's "fp16
intermediates").  Contract, stated in include/rc_shaderchain.h and DESIGN.md:
  * arithmetic stays float; only the storage of RGBA32F targets changes: store = round to nearest even, fetch = exact
    widening.  So the fp16 chain IS the fp32 chain with those targets rounded - checked bit for bit against the
    oracle run that way;
  * against the default (fp32, bit-exact) path the final 8-bit output moves by at most ONE step (ntsc presets).
The default stays fp32."""
import os

import numpy as np
import pytest

from oracle_chain import run_chain

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FP16_TOLERANCE_STEPS = 1      # of the final RGBA8 output, ntsc presets


@pytest.mark.parametrize("case,key", [("ntsc_svideo_96x64_to_256x192", "ntsc-256px-svideo"), ("ntsc_svideo_120x50_to_301x117", "ntsc-256px-svideo"),
                                      ("ntsc_320px_composite_72x40_to_320x120", "ntsc-320px"),
                                      ("ntsc_256px_composite_80x48_to_200x144", "ntsc-256px")])
def test_fp16_targets_equal_the_rounded_fp32_chain(case, key, preset_tree, rc_lib):
    from gpu_util import make_engine, run_engine
    from retrocapture_amd import engine as eng
    import chain_specs
    if key not in chain_specs.PRESETS:
        pytest.skip("preset %s not in chain_specs" % key)
    g = np.load(os.path.join(GOLD, case + ".npz"))
    vw, vh = [int(v) for v in g["viewport"]]
    passes = eng.preset_dump(preset_tree[key])["passes"]
    frames = int(g["frames"])
    e = make_engine(preset_tree[key], vw, vh)
    for _ in range(frames):
        ref = run_engine(e, g["input_rgb"])[0]          # default: RGBA32F intermediate
    assert e.passInfo(0)["format"] == "f32"
    e.shutdown()
    e = make_engine(preset_tree[key], vw, vh)
    e.setFloatTargetFp16(True)
    for _ in range(frames):
        got = run_engine(e, g["input_rgb"])[0]
    assert e.passInfo(0)["format"] == "f16"
    p0 = e.readPass(0, 0)
    assert p0.dtype == np.float16
    want = run_chain(passes, g["input_rgb"], vw, vh, frame_count=frames, f16_targets=True)
    assert np.array_equal(p0.view(np.uint16), want[0].astype(np.float16).view(np.uint16)), "pass 0 is not the rounded fp32 pass"
    assert np.array_equal(got, want[-1]), "the fp16 chain is not the fp32 chain with rounded float targets"
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= FP16_TOLERANCE_STEPS, "fp16 storage moved the output by %d steps" % d.max()
    # ... and the default path is still the bit-exact one
    assert np.array_equal(ref, g["pass%d" % (len(passes) - 1)])
    # general kernel forms read / write the 8-byte texels too
    e.setGeneralKernelsOnly(True)
    assert np.array_equal(run_engine(e, g["input_rgb"])[0], got) if frames == 1 else True
    e.shutdown()


def test_fp16_targets_full_size(preset_tree, rc_lib):
    """BASELINE config 3 (ntsc-256px-svideo, 1920x1080) with the 1024x1080 intermediate as binary16: 8.8 MB instead of
    17.7 MB per frame; output within one step of the fp32 path on noise."""
    from gpu_util import make_engine, run_engine
    frames = np.random.default_rng(8).integers(0, 256, (2, 1080, 1920, 3), dtype=np.uint8)
    e = make_engine(preset_tree["ntsc-256px-svideo"], 1920, 1080)
    ref = run_engine(e, frames)
    e.shutdown()
    e = make_engine(preset_tree["ntsc-256px-svideo"], 1920, 1080)    # a fresh engine: the chroma phase depends on FrameCount
    e.setFloatTargetFp16(True)
    got = run_engine(e, frames)
    info = e.passInfo(0)
    assert info["format"] == "f16" and (info["width"], info["height"]) == (1024, 1080)
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= FP16_TOLERANCE_STEPS and float((d == 0).mean()) > 0.9
    e.shutdown()
