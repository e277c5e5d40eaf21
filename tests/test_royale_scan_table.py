"""crt-royale pass 1 at 1:1 geometry runs from an expansion table with a remainder bound
(retrocapture_amd/csrc/kernels/pass_royale_scan.hip).  CPU check of that bound: for every table node and
random colour / distance perturbations inside the range the kernel uses it for, the oracle's exact float
evaluation of the beam function must lie within `bound` of  T + dK/dc d + (d2K/dc2 / 2) d^2 + dK/ddist dist,
with T the oracle's exact value at the node (the kernel takes T from its own general-form code on the device,
which the GPU tests hold bit-equal to the oracle)."""
import ctypes as C

import numpy as np

import oracle_lib

F = np.float32
LOG_NODES, LOG_MAX, MAX_DELTA, MAX_DIST = 192, F(2.0 ** -8), 2.6e-4, 2.0 ** -14
CONV = np.array([0.2, 0.4, 0.6], F)


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


def test_expansion_stays_within_its_bound(rc_lib):
    from retrocapture_amd import engine
    A, bound, node = engine.royale_scan_tables(1.0 / 3.0)
    n_nodes = A.shape[1]
    assert n_nodes == LOG_NODES + 256 and node.shape == (9, n_nodes)
    rng = np.random.default_rng(17)
    worst = 0.0
    per = 1500
    for jc in range(9):
        j, ch = divmod(jc, 3)
        c0 = node[jc].astype(np.float64)
        T = beam(np.broadcast_to(dd_of(j, ch, np.zeros(1, F)), (n_nodes,)).copy(), node[jc]).astype(np.float64)
        used = (bound[jc] < 1e20) & (c0 > 0)
        idx = np.nonzero(used)[0]
        # colour range of a node: half a bucket for the log-spaced nodes, MAX_DELTA (above LOG_MAX only) for byte nodes
        half = np.where(idx < LOG_NODES, 2.0 ** np.floor(np.log2(c0[idx])) / 16.0, MAX_DELTA)
        d = rng.uniform(-1, 1, (idx.size, per)) * rng.choice([1.0, 1.0, 0.3, 0.01, 0.0], (idx.size, per)) * half[:, None]
        c = (c0[idx][:, None] + d).astype(F)
        c = np.where(idx[:, None] < LOG_NODES, c, np.maximum(c, LOG_MAX))
        keep = np.abs(c.astype(np.float64) - c0[idx][:, None]) <= half[:, None] * 1.0001
        dist = (rng.uniform(-1, 1, (idx.size, per)) * rng.choice([1.0, 1.0, 0.5, 0.0], (idx.size, per)) * MAX_DIST).astype(F)
        exact = beam(dd_of(j, ch, dist.reshape(-1)), c.reshape(-1)).reshape(c.shape).astype(np.float64)
        dl = c.astype(np.float64) - c0[idx][:, None]
        a = A[jc][idx].astype(np.float64)
        model = T[idx][:, None] + dl * (a[:, 1:2] + a[:, 2:3] * dl) + a[:, 3:4] * dist.astype(np.float64)
        ratio = np.where(keep, np.abs(exact - model) / bound[jc][idx][:, None].astype(np.float64), 0.0)
        worst = max(worst, float(ratio.max()))
    # the bound carries a factor 2 on the sampled remainder and 3 on the measured rounding noise
    assert worst <= 0.75, "expansion error reaches %.2f of its bound" % worst


def test_zero_and_tiny_colours(rc_lib):
    """Colours below 2^-32 use the zero node (K = 0, bound 1.5e-8): the exact K there is below 1e-8."""
    from retrocapture_amd import engine
    A, bound, node = engine.royale_scan_tables(1.0 / 3.0)
    z = LOG_NODES
    assert (node[:, z] == 0).all() and (A[:, z] == 0).all() and (bound[:, z] >= 1.4e-8).all()
    c = np.array([0.0, 1e-30, 2.0 ** -33, 2.3e-10], F)
    for jc in range(9):
        j, ch = divmod(jc, 3)
        k = beam(np.broadcast_to(dd_of(j, ch, np.zeros(1, F)), c.shape).copy(), c)
        assert k[0] == 0 and (k <= 1e-8).all()
