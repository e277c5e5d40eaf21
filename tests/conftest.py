import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

REFERENCE = "/root/reference"
HAVE_REFERENCE = os.path.isdir(os.path.join(REFERENCE, "shaders", "shaders_glsl"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    skip_ref = pytest.mark.skip(reason="/root/reference not present on this machine")
    for item in items:
        if "reference" in item.keywords and not HAVE_REFERENCE:
            item.add_marker(skip_ref)


@pytest.fixture(scope="session")
def rc_lib():
    """The product's C-ABI library; built in-tree if missing (hipcc cross-compiles)."""
    from retrocapture_amd import engine
    if not os.path.exists(engine.library_path()):
        import __graft_entry__
        __graft_entry__.build()
    return engine.load_library()


@pytest.fixture()
def preset_tree(tmp_path):
    """Writes hand-written presets under <tmp>/shaders_glsl/... (no .glsl files: the GPU box
    has no reference assets, the engine is told to use its registry's parameter tables)."""
    import chain_specs
    return chain_specs.write_tree(str(tmp_path))
