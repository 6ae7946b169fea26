"""not gpu: the product's per-cell / per-face arithmetic (athenak_amd/csrc/akmi_numerics.hpp), compiled for the CPU with
g++ (tests/host_shim/: a stand-in for the HIP runtime header + a C-ABI wrapper), against the oracle's single-state
functions (oracle/akref.h) -- bit for bit, on a million random faces per solver: order-one states, nearly equal sides,
states spread over twelve decades, exact zeros / signed zeros / vanishing fields, super-fast flows of both signs.

This is the one place where the numerics header meets the oracle without a GPU in between.  The header is NOT a
restatement of the reference line by line (it computes one side of a face where the reference computes both, shares
sub-expressions between the PPM variants, ...), so agreement here is a statement about the rearrangement, which the
GPU suite then repeats through the kernels."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "host_shim")
SO = os.path.join(SHIM, "libnumerics_host.so")


@pytest.fixture(scope="module")
def hn():
    src = os.path.join(SHIM, "numerics_host.cpp")
    hdr = os.path.join(ROOT, "athenak_amd", "csrc", "akmi_numerics.hpp")
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        # -ffp-contract=off: products and sums rounded separately, as in the device build (and in the oracle's)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I", SHIM, "-I",
                               os.path.join(ROOT, "athenak_amd", "csrc"), src, "-o", SO])
    return C.CDLL(SO)


@pytest.fixture(scope="module")
def ref():
    from oracle import akref
    return akref.lib()


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _states(n, nv, seed):
    """n left/right states of nv variables (d, u, v, w, e[, by, bz]) + the face field, in four flavours"""
    rng = np.random.default_rng(seed)
    L = rng.uniform(-1.0, 1.0, (n, nv))
    R = rng.uniform(-1.0, 1.0, (n, nv))
    bn = rng.uniform(-1.0, 1.0, n)
    q = n//4
    for W in (L, R):                                   # densities / energies positive
        W[:, 0] = 0.5 + rng.uniform(0, 1, n)
        W[:, 4] = 0.5 + rng.uniform(0, 1, n)
    # nearly equal sides
    R[q:2*q] = L[q:2*q]*(1.0 + 1e-4*rng.uniform(-0.5, 0.5, (q, nv)))
    # twelve decades
    sl = slice(2*q, 3*q)
    for W in (L, R):
        W[sl] = np.sign(W[sl])*10.0**rng.uniform(-6, 6, (q, nv))
        W[sl, 0] = np.abs(W[sl, 0])
        W[sl, 4] = np.abs(W[sl, 4])
    bn[sl] = np.sign(bn[sl])*10.0**rng.uniform(-6, 6, q)
    # zeros, signed zeros, vanishing fields, super-fast flows
    sl = slice(3*q, n)
    m = n - 3*q
    L[sl, 1] += rng.choice([-6.0, 0.0, 6.0], m)
    R[sl, 1] = L[sl, 1] + 0.1*rng.uniform(-0.5, 0.5, m)
    z = rng.integers(0, 4, m)
    for W in (L, R):
        for k in range(2, nv):
            if k == 4:
                continue
            W[sl, k] = np.where(z == 0, 0.0, np.where(z == 1, -0.0, W[sl, k]))
    bn[sl] = np.where(rng.integers(0, 3, m) == 0, 0.0, bn[sl])
    return np.ascontiguousarray(L), np.ascontiguousarray(R), np.ascontiguousarray(bn)


def _same(a, b):
    return np.array_equal(a.view(np.uint64), b.view(np.uint64)) or bool(np.all((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))))


N = 1_000_000


@pytest.mark.parametrize("kind,name", [(1, "plm"), (2, "ppm4"), (3, "ppmx"), (4, "wenoz"), (5, "teno")])
def test_reconstructions_match_the_oracle_bitwise(hn, ref, kind, name):
    rng = np.random.default_rng(kind)
    st = rng.uniform(-1.0, 1.0, (N, 5)) + rng.integers(0, 2, (N, 1))
    st[::5, 3] = st[::5, 2]                            # flat pairs and triples: the limiters' equalities
    st[::9, 1] = st[::9, 2]
    st[::9, 3] = st[::9, 2]
    st[::13] *= 10.0**rng.uniform(-6, 6, (len(st[::13]), 1))
    st = np.ascontiguousarray(st)
    up, down = np.empty(N), np.empty(N)
    hn.hn_recon_n(kind, C.c_long(N), _p(st), _p(up), _p(down))
    f = getattr(ref, "akref_" + name)
    f.restype = None
    M = 200_000                                        # the oracle entry is one call per cell
    a, b = C.c_double(), C.c_double()
    for i in range(0, N, N//M):
        s = st[i]
        if kind == 1:
            f(C.c_double(s[1]), C.c_double(s[2]), C.c_double(s[3]), C.byref(a), C.byref(b))
        else:
            f(C.c_double(s[0]), C.c_double(s[1]), C.c_double(s[2]), C.c_double(s[3]), C.c_double(s[4]), C.byref(a), C.byref(b))
        assert _same(np.array([a.value, b.value]), np.array([up[i], down[i]])), (name, i, s, a.value, b.value, up[i], down[i])


@pytest.mark.parametrize("kind,name", [(0, "llf_hyd"), (1, "hlle_hyd"), (2, "hllc"), (4, "roe_hyd")])
def test_hydro_solvers_match_the_oracle_bitwise(hn, ref, kind, name):
    L, R, _ = _states(N, 5, 10 + kind)
    out = np.empty((N, 5))
    hn.hn_riemann_hyd_n(kind, C.c_double(1.4), C.c_long(N), _p(L), _p(R), _p(out))
    f = getattr(ref, "akref_" + name)
    f.restype = None
    o = np.empty(5)
    for i in range(0, N, 5):
        f(C.c_double(1.4), _p(L[i]), _p(R[i]), _p(o))
        assert _same(o, out[i]), (name, i, L[i], R[i], o, out[i])


@pytest.mark.parametrize("kind,name", [(0, "llf_mhd"), (1, "hlle_mhd"), (3, "hlld"), (13, "hlld")],
                         ids=["llf", "hlle", "hlld", "hlld-earlyouts"])
def test_mhd_solvers_match_the_oracle_bitwise(hn, ref, kind, name):
    gamma = 5.0/3.0
    L, R, bn = _states(N, 7, 20 + kind)
    out = np.empty((N, 7))
    hn.hn_riemann_mhd_n(kind, C.c_double(gamma), C.c_long(N), _p(L), _p(R), _p(bn), _p(out))
    f = getattr(ref, "akref_" + name)
    f.restype = None
    o = np.empty(7)
    for i in range(0, N, 4):
        f(C.c_double(gamma), _p(L[i]), _p(R[i]), C.c_double(bn[i]), _p(o))
        # both return (d, mx, my, mz, E, F(by), F(bz)); the callers store ey = -F(by), ez = F(bz) (hlld_mhd.hpp:346-347)
        assert _same(o, out[i]), (name, i, L[i], R[i], bn[i], o, out[i])
