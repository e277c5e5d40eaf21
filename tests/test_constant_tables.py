"""Filter tables typed into the oracle and the kernels: identical to each other, and - where the reference tree is at hand
(the build container; never on the GPU box) - to the numbers in the reference's shader files.  A mistyped digit in a
constant hides behind an 8-bit store (round 1 carried one in the 2-phase luma filter: 1 ulp in 6 % of the float pixels)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/shaders/shaders_glsl/ntsc/shaders"


def _c_table(text, name, n):
    m = re.search(r"%s\[%d\] = \{(.*?)\};" % (name, n), text, re.S)
    assert m, name
    return [np.float32(float(x.strip().rstrip("f"))) for x in m.group(1).replace("\n", " ").split(",") if x.strip()]


TABLES = [("k_luma3", 25), ("k_chroma3", 25), ("k_luma2", 33), ("k_chroma2", 33)]


def test_ntsc_tables_agree_between_oracle_and_kernels():
    oracle = open(os.path.join(ROOT, "oracle", "rc_passes_ntsc_xbr.c")).read()
    kernels = open(os.path.join(ROOT, "retrocapture_amd", "csrc", "kernels", "pass_ntsc.hip")).read()
    for name, n in TABLES:
        assert _c_table(oracle, name, n) == _c_table(kernels, name, n), name


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
def test_ntsc_tables_match_the_reference_shaders():
    oracle = open(os.path.join(ROOT, "oracle", "rc_passes_ntsc_xbr.c")).read()
    for fn, taps, names in (("ntsc-pass2-3phase-gamma.glsl", 24, ("k_luma3", "k_chroma3")), ("ntsc-pass2-2phase-gamma.glsl", 32, ("k_luma2", "k_chroma2"))):
        src = open(os.path.join(REF, fn)).read()
        blk = src[src.index("#define TAPS %d" % taps):]
        for gl, name in zip(("luma_filter", "chroma_filter"), names):
            m = re.search(r"const float %s\[TAPS \+ 1\] = float\[TAPS \+ 1\]\((.*?)\);" % gl, blk, re.S)
            vals = [np.float32(float(x.strip())) for x in m.group(1).replace("\n", " ").split(",") if x.strip()]
            assert vals == _c_table(oracle, name, taps + 1), (fn, gl)
