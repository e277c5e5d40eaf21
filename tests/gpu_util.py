"""Helpers for the -m gpu tests: device buffers come from torch (plumbing only)."""
import numpy as np


def to_device_rgba(rgb_frames):
    """(n, h, w, 3) or (h, w, 3) uint8 -> torch uint8 CUDA tensor (n, h, w, 4), alpha 255."""
    import torch
    a = np.asarray(rgb_frames)
    if a.ndim == 3:
        a = a[None]
    n, h, w, _ = a.shape
    rgba = np.concatenate([a, np.full((n, h, w, 1), 255, np.uint8)], -1)
    return torch.from_numpy(np.ascontiguousarray(rgba)).cuda()


def make_engine(preset_path, vw, vh, chunk=None):
    from retrocapture_amd import ShaderEngine
    e = ShaderEngine()
    assert e.init(0), "ShaderEngine.init failed: no HIP device?"
    e.setAllowMissingSources(True)
    e.setAsyncTableBuilds(False)   # the specialised forms from the first frame on (tests compare forms; test_async_table_builds turns it on)
    if chunk:
        e.setChunkFrames(chunk)
    st = e.loadPresetStatus(preset_path)
    assert st == 0, "loadPreset status %d" % st
    e.setViewport(vw, vh)
    return e


def run_engine(e, rgb_frames):
    """Applies the chain to the frames; returns list (per pass) of host arrays of frame 0 and
    the (n, oh, ow, 4) final output."""
    import torch
    d = to_device_rgba(rgb_frames)
    n, h, w, _ = d.shape
    ptr, ow, oh = e.applyShaderBatch(d, n, w, h) if n > 1 else e.applyShader(d, w, h)
    e.sync()
    last = e.passCount() - 1
    final = np.stack([e.readPass(last, k) for k in range(n)])
    assert final.shape[1:3] == (oh, ow)
    return final
