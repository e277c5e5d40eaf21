"""crt-royale pass 1 at 1:1 geometry runs from an expansion table with a MEASURED bound
(retrocapture_amd/csrc/kernels/pass_royale_scan.hip): when the tables of a geometry are built, the device evaluates the
exact beam function at every float colour a node can be selected for and every row distance of the geometry, and records
the largest difference to the polynomial  fma(c, fma(c, a2, a1), fma(W, dist, a0))  the table kernel evaluates (a0..a2: the
second-order expansion around the node written in the colour c itself; W: dK/ddist with the bound's 11-bit code in its low bits).
Checked here against the oracle's exact float evaluation of the beam function (which the GPU parity tests hold bit-equal
to the device's): on the CPU the host-built coefficients and the zero node's analytic bound; on the GPU the measured bound
itself - it must hold for random colours of every node, and for whole nodes enumerated float by float it must be EXACTLY
the largest error the oracle finds (the enumeration is complete and evaluates what the kernel evaluates)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib

F = np.float32
LOG_NODES, LOG_MAX, MAX_DELTA, MAX_DIST = 192, F(2.0 ** -8), F(2.6e-4), 2.0 ** -14
CONV = np.array([0.2, 0.4, 0.6], F)
OFF = 1.0 / 3.0
# row distances of the kind k_scan_geometry finds at 1080p (0 and a few multiples of 2^-16, both signs)
DISTS = np.array([0.0, 2.0 ** -15, -(2.0 ** -15), 3.0 * 2.0 ** -16, -(2.0 ** -14)], F)


def beam(dist, color, ph=1.0):
    lib = oracle_lib.lib()
    lib.o_royale_beam_array.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_size_t]
    dist = np.ascontiguousarray(dist, F)
    color = np.ascontiguousarray(color, F)
    out = np.empty_like(dist)
    lib.o_royale_beam_array(dist.ctypes.data, color.ctypes.data, F(ph), out.ctypes.data, dist.size)
    return out


def dd_of(j, ch, dist):
    """distance argument of scanline j, channel ch (rc_passes_royale.c o_pass_royale_scan_v, dist_round = 0)"""
    dist = np.asarray(dist, F)
    if j == 0:
        return dist - CONV[ch]
    if j == 1:
        return np.abs((F(1) + CONV[ch]) - dist)
    a, b = dist + (F(1) - CONV[ch]), (F(2) + CONV[ch]) - dist
    return a + F(0) * (b - a)


_fmaf = C.CDLL("libm.so.6").fmaf
_fmaf.restype = C.c_float
_fmaf.argtypes = [C.c_float, C.c_float, C.c_float]


def fma32(a, b, c):
    """float32 fused multiply-add, element by element through libm (exact: what the device's v_fma_f32 returns)"""
    a, b, c = np.broadcast_arrays(np.asarray(a, F), np.asarray(b, F), np.asarray(c, F))
    return np.array([_fmaf(x, y, z) for x, y, z in zip(a.ravel().tolist(), b.ravel().tolist(), c.ravel().tolist())], F).reshape(a.shape)


def node_range(n, c0):
    """the float colours (as uint32 bit patterns, inclusive) the table kernel selects node n for (scan_node_range)"""
    if n < LOG_NODES:
        lo = 0x2F800000 + (n << 20)
        return lo, lo + (1 << 20) - 1
    lo = max(F(c0) - MAX_DELTA, LOG_MAX)
    hi = min(F(c0) + MAX_DELTA, F(1))
    return int(F(lo).view(np.uint32)), int(F(hi).view(np.uint32))


def test_host_tables(rc_lib):
    from retrocapture_amd import engine
    A, zero, node = engine.royale_scan_tables(OFF)
    n_nodes = A.shape[1]
    assert n_nodes == LOG_NODES + 256 and node.shape == (9, n_nodes) and (zero == 0).all()
    assert (node == node[0]).all() and (node[0, LOG_NODES] == 0) and np.all(np.diff(node[0, :LOG_NODES]) > 0) and np.all(np.diff(node[0, LOG_NODES:]) > 0)
    # first derivative in colour against a central difference of the oracle (double precision differences of float values)
    for jc in (0, 4, 8):
        j, ch = divmod(jc, 3)
        idx = np.arange(LOG_NODES + 40, n_nodes, 7)
        c0 = node[jc][idx].astype(np.float64)
        h = 1e-3 * c0
        d0 = np.broadcast_to(dd_of(j, ch, np.zeros(1, F)), idx.shape).copy()
        kp, km = beam(d0, (c0 + h).astype(F)).astype(np.float64), beam(d0, (c0 - h).astype(F)).astype(np.float64)
        num = (kp - km) / ((c0 + h).astype(F).astype(np.float64) - (c0 - h).astype(F).astype(np.float64))
        assert np.allclose(A[jc][idx, 1], num, rtol=2e-2, atol=1e-4)


def test_zero_and_tiny_colours(rc_lib):
    """Colours below 2^-32 use the zero node (K = 0): its bound 2.5e-8 is analytic - 0 <= K <= 81 colour (beta <= 4,
    1 / alpha <= 35.4, gamma_impl >= 0.88) - checked here on a sweep of such colours, every role and distance."""
    rng = np.random.default_rng(3)
    bits = np.concatenate([rng.integers(0x00800000, 0x2F800000, 200000, dtype=np.uint32), np.array([0, 1, 0x2F7FFFFF], np.uint32)])
    c = bits.view(F)
    for jc in range(9):
        j, ch = divmod(jc, 3)
        for dist in DISTS:
            k = beam(np.broadcast_to(dd_of(j, ch, np.array([dist], F)), c.shape).copy(), c)
            assert (k >= 0).all() and (k <= 81.0 * c.astype(np.float64) + 1e-30).all() and float(k.max()) < 1.9e-8


@pytest.mark.gpu
def test_measured_bound_holds_for_every_node(rc_lib):
    from retrocapture_amd import engine
    _, _, node = engine.royale_scan_tables(OFF)
    A, bound = engine.royale_scan_bounds(OFF, DISTS)
    n_nodes = A.shape[1]
    assert ((bound[:, LOG_NODES] >= F(2.5e-8)) & (bound[:, LOG_NODES] <= F(2.6e-8))).all()
    rng = np.random.default_rng(17)
    per = 600
    worst = 0.0
    for jc in range(9):
        j, ch = divmod(jc, 3)
        c0 = node[jc]
        # every node the kernel can select: all log buckets, and the bytes whose range reaches above 2^-8 (darker bytes always
        # select a log bucket; their records carry the largest bound code)
        sel = (c0 > 0) & ((np.arange(n_nodes) < LOG_NODES) | (c0 + MAX_DELTA >= LOG_MAX))
        assert (bound[jc][(c0 > 0) & ~sel] > 3e-3).all()
        idx = np.nonzero(sel)[0]
        lo, hi = np.array([node_range(int(n), c0[n]) for n in idx]).T
        bits = (lo[:, None] + (rng.random((idx.size, per)) * (hi - lo + 1)[:, None]).astype(np.int64)).astype(np.uint32)
        c = bits.view(F)
        dist = DISTS[rng.integers(0, DISTS.size, c.shape)]
        exact = beam(dd_of(j, ch, dist.reshape(-1)), c.reshape(-1)).reshape(c.shape).astype(np.float64)
        a = A[jc][idx].astype(np.float64)   # (W as stored: the kernel multiplies dist by it with the bound's code bits in)
        cd = c.astype(np.float64)
        # the kernel's three fmas in double (products of floats are exact there; the two roundings can move the model by one
        # float ulp, which the tolerance below allows)
        model = (cd * (cd * a[:, 2:3] + a[:, 1:2]).astype(F) + (a[:, 3:4] * dist + a[:, 0:1]).astype(F)).astype(F).astype(np.float64)
        err = np.abs(exact - model)
        ulp = np.spacing(np.abs(model).astype(F)).astype(np.float64)
        assert (err <= bound[jc][idx][:, None] + 1.01 * ulp).all(), "role %d: the expansion leaves its measured bound" % jc
        worst = max(worst, float((err / bound[jc][idx][:, None]).max()))
    assert worst > 0.5, "the bounds are far from tight (%.2f): not what an exhaustive measurement gives" % worst


@pytest.mark.gpu
@pytest.mark.parametrize("jc,n", [(0, LOG_NODES + 255), (4, LOG_NODES + 200), (8, LOG_NODES + 129), (2, LOG_NODES + 90)])
def test_measured_bound_is_the_exhaustive_maximum(jc, n, rc_lib):
    """A whole byte node, float by float, against the oracle: the device's bound is the largest error there is (times its
    rounding allowance 1.000001, + 1e-12, rounded up to the table's 11-bit code)."""
    from retrocapture_amd import engine
    _, _, node = engine.royale_scan_tables(OFF)
    A, bound = engine.royale_scan_bounds(OFF, DISTS)
    j, ch = divmod(jc, 3)
    c0 = node[jc][n]
    lo, hi = node_range(n, c0)
    c = np.arange(lo, hi + 1, dtype=np.uint32).view(F)
    a = A[jc][n]
    inner = fma32(c, a[2], a[1])
    # W: dK/ddist in the upper bits, the bound's code below - unknown while the bound is being measured, so the device takes
    # the worse of the two extreme codes (the polynomial is monotone in W)
    w_ends = [((a[3:4].view(np.uint32) & np.uint32(0xFFFFF800)) | np.uint32(code)).view(F)[0] for code in (0, 0x7FF)]
    worst = 0.0
    for dist in DISTS:
        exact = beam(np.broadcast_to(dd_of(j, ch, np.array([dist], F)), c.shape).copy(), c).astype(np.float64)
        for w in w_ends:
            base = fma32(w, dist, a[0])
            model = fma32(c, inner, np.broadcast_to(base, c.shape)).astype(np.float64)
            worst = max(worst, float(np.abs(exact - model).max()))
    want = F(F(np.nextafter(F(worst), F(np.inf)) if F(worst) < worst else F(worst)) * F(1.000001) + F(1e-12))
    # the table keeps the bound as an 11-bit code rounded up (6 mantissa bits): at most 1/64 above the measured value
    assert float(want) <= float(bound[jc][n]) <= float(want) * (1.0 + 1.0 / 64.0) * 1.0001, (bound[jc][n], want, worst)
