"""TEST INFRASTRUCTURE (oracle/): empirical extraction of llvmpipe's (Mesa 23.2.1) texel decode,
UNORM8 / sRGB8 store rounding and bilinear filter arithmetic, via oracle/_ref/glprobe.
Writes tests/golden/llvmpipe_tables.npz (sRGB decode table, sRGB encode thresholds) which the C
oracle and the HIP kernels restate as constant tables."""
import numpy as np, os, sys
sys.path.insert(0, os.path.dirname(__file__))
from llvmpipe_math_model import run, F, fmaf
import subprocess

PROBE = os.path.join(os.path.dirname(__file__), "..", "_ref", "glprobe")

def run_tex(body, inp, W, H, tex, TW, TH, fmt, filt, wrap, target="f32"):
    open("/tmp/_b.glsl", "w").write(body)
    open("/tmp/_t.bin", "wb").write(tex.tobytes())
    r = subprocess.run([PROBE, "/tmp/_b.glsl", str(W), str(H), target, "/tmp/_t.bin", str(TW), str(TH), fmt, filt, wrap],
                       input=inp.tobytes(), capture_output=True)
    if r.returncode: raise RuntimeError(r.stderr.decode())
    return np.frombuffer(r.stdout, dtype=F if target == "f32" else np.uint8).reshape(-1, 4)

def thresholds(target):
    """For each k in 1..255 find the smallest float x (bit pattern) whose stored byte is >= k."""
    lo = np.zeros(255, np.int64)                       # bits of 0.0 -> byte 0  (< k)
    hi = np.full(255, np.float32(1.0).view(np.int32), np.int64)   # 1.0 -> 255 (>= k)
    ks = np.arange(1, 256)
    while np.any(hi - lo > 1):
        mid = (lo + hi) // 2
        x = mid.astype(np.int32).view(F)
        inp = np.zeros((256, 4), F); inp[:255, 0] = x
        o = run("vec4 f(vec4 v){ return vec4(v.x, v.x, v.x, v.x); }", inp, 256, 1, target)
        b = o[:255, 0].astype(np.int64)
        ge = b >= ks
        hi = np.where(ge, mid, hi); lo = np.where(ge, lo, mid)
    return hi.astype(np.int32).view(F)

if __name__ == "__main__":
    t_u8 = thresholds("u8")
    # candidate: RNE(x*255)
    print("unorm8 thresholds[:4]", t_u8[:4], "expected (k-0.5)/255:", [(k - 0.5) / 255 for k in range(1, 5)])
    xs = np.concatenate([t_u8, np.nextafter(t_u8, F(0)), np.random.default_rng(0).random(100000, dtype=F)])
    def store_u8_model(x):
        x = np.clip(x, F(0), F(1))
        return np.rint((x * F(255.0)).astype(F)).astype(np.int32)          # RNE of fl(x*255)
    def store_u8_model2(x):
        x = np.clip(x, F(0), F(1))
        return (((x * F(255.0 / 256.0)).astype(F) + F(32768.0)).astype(F).view(np.int32)) & 0xff
    n = (len(xs) + 255) // 256 * 256
    inp = np.zeros((n, 4), F); inp[:len(xs), 0] = xs
    o = run("vec4 f(vec4 v){ return v.xxxx; }", inp, 256, n // 256, "u8")[:len(xs), 0]
    print("u8 model rint(x*255) mismatches:", int((store_u8_model(xs) != o).sum()), " model2:", int((store_u8_model2(xs) != o).sum()))
    # alpha of srgb target is linear
    t_s = thresholds("srgb8")
    print("srgb thresholds[:4]", t_s[:4], t_s[-2:])
    # sRGB decode table + unorm decode
    tex = np.zeros((256, 4), np.uint8); tex[:, 0] = tex[:, 1] = tex[:, 2] = tex[:, 3] = np.arange(256)
    inp = np.zeros((256, 4), F)
    dec_s = run_tex("vec4 f(vec4 v){ return texelFetch(S, ivec2(gl_FragCoord.xy), 0); }", inp, 256, 1, tex, 256, 1, "srgb8", "nearest", "edge")
    dec_u = run_tex("vec4 f(vec4 v){ return texelFetch(S, ivec2(gl_FragCoord.xy), 0); }", inp, 256, 1, tex, 256, 1, "rgba8", "nearest", "edge")
    k = np.arange(256)
    print("unorm decode == k/255f :", bool(np.all(dec_u[:, 0] == (k.astype(F) / F(255)))), " == k*(1/255f):", bool(np.all(dec_u[:, 0] == (k.astype(F) * (F(1) / F(255))).astype(F))))
    print("srgb alpha decode linear:", bool(np.all(dec_s[:, 3] == dec_u[:, 3])))
    c = k / 255.0
    ref = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
    print("srgb decode == f32(double formula):", int((dec_s[:, 0] != ref.astype(F)).sum()), "mismatches")
    rgb = np.zeros((256, 3), np.uint8); rgb[:, 0] = k; rgb[:, 1] = 255 - k; rgb[:, 2] = 7
    dec_rgb = run_tex("vec4 f(vec4 v){ return texelFetch(S, ivec2(gl_FragCoord.xy), 0); }", inp, 256, 1, rgb, 256, 1, "rgb8", "nearest", "edge")
    print("GL_RGB alpha:", np.unique(dec_rgb[:, 3]), "r ok:", bool(np.all(dec_rgb[:, 0] == dec_u[:, 0])))
    np.savez(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "llvmpipe_tables.npz"),
             srgb_decode=dec_s[:, 0].copy(), srgb_encode_thresholds=t_s, unorm8_thresholds=t_u8)
