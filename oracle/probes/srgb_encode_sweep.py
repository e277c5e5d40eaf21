#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (oracle/probes): exhaustive check of the oracle's sRGB8 encode against Mesa llvmpipe.

Every float with bits 0 .. 0x3f800000 (all of [0, 1]) and a sample of values outside that range (negative, > 1,
infinities, NaNs, denormals) is rendered through an sRGB8 render target with GL_FRAMEBUFFER_SRGB enabled
(oracle/_ref/glprobe, identity fragment shader, the reference's state: ShaderEngine.cpp:944-952) and the stored
byte is compared with oracle/rc_sampler.c o_store_srgb8.  Needs /root/reference-free tools only, but the GL of the
build container: run here, not on the GPU box.

    python3 oracle/probes/srgb_encode_sweep.py            # ~3 minutes on 8 cores
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GLPROBE = os.path.join(ROOT, "oracle", "_ref", "glprobe")
W, H = 4096, 2048


def gl_encode(vals, body):
    n = vals.size
    px = (n + 2) // 3
    pad = np.zeros(W * H * 4, dtype=np.float32).reshape(-1, 4)
    flat = np.zeros(px * 3, dtype=np.float32)
    flat[:n] = vals
    pad[:px, :3] = flat.reshape(-1, 3)
    pad[:, 3] = 1.0
    out = subprocess.run([GLPROBE, body, str(W), str(H), "srgb8"], input=pad.tobytes(), capture_output=True, check=True).stdout
    return np.frombuffer(out, dtype=np.uint8).reshape(-1, 4)[:px, :3].reshape(-1)[:n]


def main():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    lib.o_store_srgb8_array.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    tmp = tempfile.TemporaryDirectory()
    body = os.path.join(tmp.name, "id.glsl")
    open(body, "w").write("vec4 f(vec4 v){ return v; }\n")
    per = W * H * 3
    total_bad = 0
    first_bad = []

    def check(bits):
        nonlocal total_bad
        vals = bits.view(np.float32)
        got = gl_encode(vals, body)
        want = np.empty(vals.size, dtype=np.uint8)
        lib.o_store_srgb8_array(vals.ctypes.data, want.ctypes.data, vals.size)
        bad = np.nonzero(got != want)[0]
        total_bad += bad.size
        for i in bad[:5]:
            first_bad.append((hex(int(bits[i])), int(got[i]), int(want[i])))

    end = 0x3f800000 + 1
    for start in range(0, end, per):
        check(np.arange(start, min(end, start + per), dtype=np.uint32))
        print("\r%5.1f %%  mismatches %d" % (100.0 * min(end, start + per) / end, total_bad), end="", file=sys.stderr)
    rng = np.random.default_rng(5)
    outside = np.concatenate([
        rng.integers(0x3f800001, 0x7f800000, 1 << 20, dtype=np.uint32),              # > 1
        rng.integers(0x80000000, 0xff800000, 1 << 20, dtype=np.uint32),              # negative
        np.array([0x7f800000, 0xff800000, 0x7fc00000, 0xffc00000, 0x7f800001, 0x80000000], dtype=np.uint32),
        rng.integers(0x7f800001, 0x80000000, 1 << 12, dtype=np.uint32)])             # NaNs
    check(outside)
    print("\nfloats compared: %d in [0,1] + %d outside; mismatches: %d" % (end, outside.size, total_bad))
    for b in first_bad[:20]:
        print("  bits %s: GL %d, oracle %d" % b)
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
