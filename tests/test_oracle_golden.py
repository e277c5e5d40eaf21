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
    "scanline_320x240": "scanline",
    "scanline_64x48_to_160x100": "scanline",
    "crt_pi_96x64_to_192x128": "crt-pi",
    "crt_pi_80x60_to_250x190": "crt-pi",
}

# exact-match floor per preset (fraction of bytes identical to llvmpipe) and max |diff|
BAR = {"scanline": (1.0, 0), "crt-pi": (1.0, 0)}


def preset_passes(tmp_path, key):
    """Parse our hand-written preset with the product's parser (host only, no GPU)."""
    from retrocapture_amd import engine
    tree = chain_specs.write_tree(str(tmp_path))
    return engine.preset_dump(tree[key])["passes"]


@pytest.mark.parametrize("case", sorted(CASES))
def test_oracle_matches_llvmpipe(case, tmp_path, rc_lib):
    g = np.load(os.path.join(GOLD, case + ".npz"))
    key = CASES[case]
    passes = preset_passes(tmp_path, key)
    vw, vh = [int(v) for v in g["viewport"]]
    outs = run_chain(passes, g["input_rgb"], vw, vh, frame_count=int(g["frames"]))
    assert len(outs) == int(g["n_passes"])
    floor, maxdiff = BAR[key]
    for i, o in enumerate(outs):
        ref = g["pass%d" % i]
        assert o.shape == ref.shape, (i, o.shape, ref.shape)
        if ref.dtype == np.uint8:
            d = np.abs(o.astype(np.int32) - ref.astype(np.int32))
            exact = float((d == 0).mean())
            assert d.max() <= maxdiff and exact >= floor, "pass %d: exact %.5f max %d" % (i, exact, d.max())
        else:
            assert np.array_equal(o.view(np.uint32), ref.view(np.uint32)) or np.allclose(o, ref, rtol=1e-6, atol=1e-7)


def test_every_golden_file_has_a_case():
    names = {os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz"))} - {"llvmpipe_tables"}
    assert names <= set(CASES) | EXTRA_GOLDEN


EXTRA_GOLDEN = set()
