"""TEST INFRASTRUCTURE (oracle/): numpy model of Mesa 23.2 gallivm float math
(lp_build_exp2 / lp_build_log2 / lp_build_pow / lp_build_sin_or_cos), compared bit-for-bit
against the real llvmpipe through oracle/_ref/glprobe. Determines whether the polynomial
steps are fused (FMA) or not, which the C oracle and the HIP kernels then restate."""
import numpy as np, subprocess, sys, os

F = np.float32
PROBE = os.path.join(os.path.dirname(__file__), "..", "_ref", "glprobe")

def mad(a, b, c, fma):
    if fma:
        return (a.astype(np.float64) * b.astype(np.float64) + np.float64(c)).astype(F)
    return (a * b).astype(F) + F(c)

EXP2_C = [1.000000000000000000000, 0.693153073200168932794, 0.240153617044375388211,
          0.0558263180532956664775, 0.00898934009049466391101, 0.00187757667519147912699]
# LOG_POLY_DEGREE 4 (found empirically: the degree-5 table does not match Mesa 23.2.1)
LOG2_C = [2.88539009343309178325, 0.961791550404184197881, 0.577440339438736392009,
          0.403343858251329912514, 0.406718052498846252698]

def poly(x, coeffs, fma):
    x2 = (x * x).astype(F)
    even = odd = None
    for i in reversed(range(len(coeffs))):
        c = F(coeffs[i])
        if i % 2 == 0:
            even = mad(x2, even, c, fma) if even is not None else np.full_like(x, c)
        else:
            odd = mad(x2, odd, c, fma) if odd is not None else np.full_like(x, c)
    return mad(odd, x, even, fma) if not fma else (odd.astype(np.float64) * x.astype(np.float64) + even.astype(np.float64)).astype(F)

def exp2(x, fma):
    x = np.minimum(F(128.0), x)
    x = np.maximum(F(-126.99999), x)
    ip = np.floor(x).astype(F)
    fp = (x - ip).astype(F)
    e = ((ip.astype(np.int32) + 127) << 23).astype(np.int32).view(F)
    return (e * poly(fp, EXP2_C, fma)).astype(F)

def log2(x, fma):
    i = x.view(np.int32)
    logexp = (((i & 0x7f800000) >> 23) - 127).astype(F)
    mant = ((i & 0x007fffff) | 0x3f800000).view(F)
    y = ((mant - F(1)) / (mant + F(1))).astype(F)
    z = (y * y).astype(F)
    p = poly(z, LOG2_C, fma)
    if fma:
        return (y.astype(np.float64) * p.astype(np.float64) + logexp.astype(np.float64)).astype(F)
    return (y * p).astype(F) + logexp

def pow_(x, y, fma):
    return exp2((log2(x, fma) * F(y)).astype(F), fma)

def run(body, inp, W, H, target="f32"):
    open("/tmp/_b.glsl", "w").write(body)
    r = subprocess.run([PROBE, "/tmp/_b.glsl", str(W), str(H), target], input=inp.tobytes(), capture_output=True)
    if r.returncode: raise RuntimeError(r.stderr.decode())
    if target == "f32":
        return np.frombuffer(r.stdout, dtype=F).reshape(-1, 4)
    return np.frombuffer(r.stdout, dtype=np.uint8).reshape(-1, 4)

if __name__ == "__main__":
    W, H = 1024, 256
    rng = np.random.default_rng(1)
    n = W * H
    inp = np.stack([rng.random(n, dtype=F), (rng.random(n, dtype=F) * 40 - 20).astype(F),
                    (rng.random(n, dtype=F) * 100 + F(1e-3)).astype(F), rng.random(n, dtype=F)], -1).astype(F)
    o = run("vec4 f(vec4 v){ return vec4(pow(v.x, 2.4), exp2(v.y), log2(v.z), pow(v.w, 1.0/2.2)); }", inp, W, H)
    for fma in (False, True):
        m = [pow_(inp[:, 0], 2.4, fma), exp2(inp[:, 1], fma), log2(inp[:, 2], fma), pow_(inp[:, 3], F(1.0) / F(2.2), fma)]
        print("fma" if fma else "nofma", [float((m[k].view(np.int32) != o[:, k].view(np.int32)).mean()) for k in range(4)])


def fmaf(a, b, c):
    return (np.asarray(a, F).astype(np.float64) * np.asarray(b, F).astype(np.float64) + np.asarray(c, F).astype(np.float64)).astype(F)

def sincos(x, want_cos):
    """sse_mathfun-style sin/cos as in gallivm lp_build_sin_or_cos."""
    xi = x.view(np.int32)
    xa = (xi & 0x7fffffff).view(F)
    sign = xi & np.int32(-2147483648)
    y = (xa * F(1.27323954473516)).astype(F)
    j = y.astype(np.int32)            # truncation
    j = (j + 1) & ~1
    y2 = j.astype(F)
    j2 = j - 2 if want_cos else j
    if want_cos:
        swap = ((~j2) & 4) << 29
    else:
        swap = (j & 4) << 29
    polymask = (j2 & 2) == 0
    x3 = fmaf(y2, F(-0.78515625), xa)
    x3 = fmaf(y2, F(-2.4187564849853515625e-4), x3)
    x3 = fmaf(y2, F(-3.77489497744594108e-8), x3)
    z = (x3 * x3).astype(F)
    yc = fmaf(np.full_like(z, F(2.443315711809948E-005)), z, F(-1.388731625493765E-003))
    yc = fmaf(yc, z, F(4.166664568298827E-002))
    yc = (yc * z).astype(F); yc = (yc * z).astype(F)
    yc = (yc - (z * F(0.5)).astype(F)).astype(F)
    yc = (yc + F(1)).astype(F)
    ys = fmaf(np.full_like(z, F(-1.9515295891E-4)), z, F(8.3321608736E-3))
    ys = fmaf(ys, z, F(-1.6666654611E-1))
    ys = (ys * z).astype(F)
    ys = fmaf(ys, x3, x3)
    r = np.where(polymask, ys, yc)
    sb = swap if want_cos else (sign ^ swap)
    return (r.view(np.int32) ^ sb.astype(np.int32)).view(F)

def probe2():
    W, H = 1024, 64
    rng = np.random.default_rng(2)
    n = W * H
    a = (rng.random(n, dtype=F) * 200 - 100).astype(F)
    b = (rng.random(n, dtype=F) * 8 - 4).astype(F)
    c = (rng.random(n, dtype=F) * 10 + F(0.01)).astype(F)
    d = rng.random(n, dtype=F)
    inp = np.stack([a, b, c, d], -1).astype(F)
    def cmp(name, got, want):
        dd = got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64)
        print(f"{name:28s} mismatch {float((dd != 0).mean()):.6f} max_ulp {int(np.abs(dd).max())}")
    o = run("vec4 f(vec4 v){ return vec4(sin(v.x), cos(v.x), sin(v.y), cos(v.y)); }", inp, W, H)
    cmp("sin big", sincos(a, False), o[:, 0]); cmp("cos big", sincos(a, True), o[:, 1])
    cmp("sin small", sincos(b, False), o[:, 2]); cmp("cos small", sincos(b, True), o[:, 3])
    o = run("vec4 f(vec4 v){ return vec4(exp(v.y), sqrt(v.z), inversesqrt(v.z), 1.0/v.z); }", inp, W, H)
    cmp("exp", exp2((b * F(1.4426950408889634)).astype(F), True), o[:, 0])
    cmp("sqrt", np.sqrt(c).astype(F), o[:, 1])
    cmp("rsq=1/sqrt", (F(1) / np.sqrt(c).astype(F)).astype(F), o[:, 2])
    cmp("rcp", (F(1) / c).astype(F), o[:, 3])
    o = run("vec4 f(vec4 v){ return vec4(v.x/v.z, fract(v.x), mod(v.x, v.z), log(v.z)); }", inp, W, H)
    cmp("div", (a / c).astype(F), o[:, 0])
    cmp("fract x-floor", (a - np.floor(a)).astype(F), o[:, 1])
    cmp("mod x-y*floor(x/y)", (a - (c * np.floor((a / c).astype(F)).astype(F)).astype(F)).astype(F), o[:, 2])
    cmp("log = log2*ln2", (log2(c, True) * F(0.69314718055994529)).astype(F), o[:, 3])
    o = run("vec4 f(vec4 v){ return vec4(mix(v.x, v.y, v.w), smoothstep(v.y, v.z, v.w*4.0), mix(v.x, v.y, 0.25), clamp(v.y,0.0,1.0)); }", inp, W, H)
    cmp("mix a+t*(b-a)", (a + (d * (b - a).astype(F)).astype(F)).astype(F), o[:, 0])
    cmp("mix a*(1-t)+b*t", ((a * (F(1) - d).astype(F)).astype(F) + (b * d).astype(F)).astype(F), o[:, 0])
    cmp("mix fma(t,b-a... )", fmaf(d, (b - a).astype(F), a), o[:, 0])
    t = np.clip((((d * F(4)).astype(F) - b).astype(F) / (c - b).astype(F)).astype(F), F(0), F(1))
    cmp("smoothstep t*t*(3-2t)", ((t * t).astype(F) * (F(3) - (F(2) * t).astype(F)).astype(F)).astype(F), o[:, 1])
    cmp("smoothstep t*(t*(3-2t))", (t * (t * (F(3) - (F(2) * t).astype(F)).astype(F)).astype(F)).astype(F), o[:, 1])
    cmp("mix const a+t*(b-a)", (a + (F(0.25) * (b - a).astype(F)).astype(F)).astype(F), o[:, 2])
    cmp("mix const a*(1-t)+b*t", ((a * F(0.75)).astype(F) + (b * F(0.25)).astype(F)).astype(F), o[:, 2])

if __name__ == "__main__":
    probe2()
