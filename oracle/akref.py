"""ctypes binding of the CPU ORACLE (oracle/libakref.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/akref.h).  The product package `athenak_amd` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

RECON = {"dc": 0, "plm": 1, "ppm4": 2, "ppmx": 3, "wenoz": 4, "teno": 5}
RSOLVER = {"llf": 0, "hlle": 1, "hllc": 2, "hlld": 3, "roe": 4, "advect": 5}
BC = {"block": -1, "periodic": 0, "outflow": 1, "reflect": 2, "user": 3, "inflow": 4, "diode": 5,
      "vacuum": 6}
PGEN = {"linear_wave": 0, "shock_tube": 1, "orszag_tang": 2, "blast": 3}


class Pack(C.Structure):
    """struct akmi_pack (include/akmi.h)"""
    _fields_ = [("nmb", C.c_int), ("nvar", C.c_int), ("nx1", C.c_int), ("nx2", C.c_int),
                ("nx3", C.c_int), ("ng", C.c_int), ("dx", C.c_void_p),
                ("gamma", C.c_double), ("dfloor", C.c_double), ("pfloor", C.c_double),
                ("tfloor", C.c_double), ("sfloor", C.c_double), ("sigma_max", C.c_double),
                ("iso_cs", C.c_double), ("is_ideal", C.c_int)]


class Params(C.Structure):
    """struct akref_params (oracle/akref.h)"""
    _fields_ = [("nx1", C.c_int), ("nx2", C.c_int), ("nx3", C.c_int),
                ("mb_nx1", C.c_int), ("mb_nx2", C.c_int), ("mb_nx3", C.c_int),
                ("ng", C.c_int),
                ("x1min", C.c_double), ("x1max", C.c_double), ("x2min", C.c_double),
                ("x2max", C.c_double), ("x3min", C.c_double), ("x3max", C.c_double),
                ("bcs", C.c_int*6),
                ("nstages", C.c_int), ("cfl", C.c_double), ("tlim", C.c_double),
                ("nlim", C.c_int),
                ("is_mhd", C.c_int), ("recon", C.c_int), ("rsolver", C.c_int),
                ("gamma", C.c_double), ("dfloor", C.c_double), ("pfloor", C.c_double),
                ("tfloor", C.c_double), ("sfloor", C.c_double), ("sigma_max", C.c_double),
                ("is_ideal", C.c_int), ("iso_cs", C.c_double), ("nscalars", C.c_int),
                ("fofc", C.c_int),
                ("kinematic", C.c_int),
                ("eta_ad", C.c_double),
                ("nu_iso", C.c_double), ("alpha_iso", C.c_double), ("eta_ohm", C.c_double),
                ("pgen", C.c_int),
                ("wave_flag", C.c_int), ("along_x1", C.c_int), ("along_x2", C.c_int),
                ("along_x3", C.c_int),
                ("amp", C.c_double), ("dens", C.c_double), ("pgas", C.c_double),
                ("vx0", C.c_double), ("vy0", C.c_double), ("vz0", C.c_double),
                ("bx0", C.c_double), ("by0", C.c_double), ("bz0", C.c_double),
                ("shock_dir", C.c_int), ("xshock", C.c_double),
                ("wl", C.c_double*8), ("wr", C.c_double*8),
                ("pi_amb", C.c_double), ("di_amb", C.c_double), ("prat", C.c_double),
                ("drat", C.c_double), ("b_amb", C.c_double), ("inner_radius", C.c_double),
                ("outer_radius", C.c_double),
                ("split_kernels", C.c_int),
                ("smr_nmb", C.c_int), ("smr_root_level", C.c_int),
                ("smr_lloc", C.c_void_p), ("smr_nghbr", C.c_void_p), ("prolong_prims", C.c_int)]


def build(force=False):
    """Compile oracle/libakref.so with the committed Makefile."""
    so = os.path.join(_HERE, "libakref.so")
    srcs = [os.path.join(_HERE, f) for f in ("akref_kernels.c", "akref_sim.c", "akref_smr.c", "akref.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "akmi.h"))
    if force or not os.path.exists(so) or any(
            os.path.getmtime(s) > os.path.getmtime(so) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libakref.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libakref.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.akref_create.restype = C.c_void_p
        L.akref_create.argtypes = [C.POINTER(Params)]
        L.akref_array.restype = C.c_void_p
        L.akref_array.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_longlong)]
        for f in ("akref_time", "akref_dt", "akref_tlim"):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("akref_ncycle", "akref_nmb", "akref_nfofc", "akref_step", "akref_run"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("akref_initialize", "akref_reinitialize", "akref_destroy"):
            getattr(L, f).restype = None
            getattr(L, f).argtypes = [C.c_void_p]
        L.akref_linear_wave_errors.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.akref_divb.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.akref_totals.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.akref_pack.argtypes = [C.c_void_p, C.POINTER(Pack)]
        L.akref_bvals_cc_segsize.restype = C.c_longlong
        L.akref_bvals_fc_segsize.restype = C.c_longlong
        # the checker is serial unless a caller (bench.py's cpu_baseline) asks for threads: without
        # this call libgomp starts one spinning thread per core in every test process
        L.akref_set_threads(1)
        _LIB = L
    return _LIB


def default_params(**kw):
    p = Params()
    lib().akref_params_default(C.byref(p))
    set_params(p, **kw)
    return p


def set_params(p, **kw):
    for k, v in kw.items():
        if k == "bcs":
            for i, b in enumerate(v):
                p.bcs[i] = BC[b] if isinstance(b, str) else int(b)
        elif k in ("wl", "wr"):
            for i, x in enumerate(v):
                getattr(p, k)[i] = float(x)
        elif k == "recon" and isinstance(v, str):
            p.recon = RECON[v]
        elif k == "rsolver" and isinstance(v, str):
            p.rsolver = RSOLVER[v]
        elif k == "pgen" and isinstance(v, str):
            p.pgen = PGEN[v]
        else:
            if not hasattr(p, k):
                raise KeyError(k)
            setattr(p, k, v)
    return p


def make_pack(nmb, nx1, nx2, nx3, ng, dx, gamma, nvar=5, dfloor=None, pfloor=None, tfloor=None,
              sfloor=None, sigma_max=None):
    """Build an akmi_pack for host (numpy) arrays.  dx: float64 array [nmb,3] kept alive by
    the returned tuple."""
    flt_min = float(np.finfo(np.float32).tiny)
    flt_max = float(np.finfo(np.float32).max)
    dx = np.ascontiguousarray(dx, dtype=np.float64).reshape(nmb, 3)
    pk = Pack(nmb, nvar, nx1, nx2, nx3, ng, dx.ctypes.data, gamma,
              flt_min if dfloor is None else dfloor, flt_min if pfloor is None else pfloor,
              flt_min if tfloor is None else tfloor, flt_min if sfloor is None else sfloor,
              flt_max if sigma_max is None else sigma_max, 1.0, 1)
    return pk, dx


def ptr(a):
    """C pointer to a contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


class Sim:
    """Whole-run oracle: mesh + pgen + RK driver on one process."""

    def __init__(self, smr_lloc=None, smr_nghbr=None, smr_root_level=0, **kw):
        self.params = default_params(**kw)
        if smr_lloc is not None:
            # statically refined mesh: Z-ordered leaves {lx1,lx2,lx3,level} and their neighbour table
            # {gid, level, dest} per NeighborIndex slot, built by the caller (kept alive here)
            self._lloc = np.ascontiguousarray(smr_lloc, dtype=np.int32).reshape(-1, 4)
            self._nghbr = np.ascontiguousarray(smr_nghbr, dtype=np.int32).reshape(-1, 56, 3)
            assert len(self._lloc) == len(self._nghbr)
            self.params.smr_nmb = len(self._lloc)
            self.params.smr_root_level = int(smr_root_level)
            self.params.smr_lloc = self._lloc.ctypes.data
            self.params.smr_nghbr = self._nghbr.ctypes.data
        self.L = lib()
        self.h = self.L.akref_create(C.byref(self.params))
        if not self.h:
            raise ValueError("akref_create failed (mesh not divisible by meshblock?)")
        self.h = C.c_void_p(self.h)

    def close(self):
        if self.h:
            self.L.akref_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def initialize(self):
        self.L.akref_initialize(self.h)

    def reinitialize(self):
        self.L.akref_reinitialize(self.h)

    def step(self):
        return self.L.akref_step(self.h)

    def run(self):
        return self.L.akref_run(self.h)

    time = property(lambda s: s.L.akref_time(s.h))
    dt = property(lambda s: s.L.akref_dt(s.h))
    tlim = property(lambda s: s.L.akref_tlim(s.h))
    ncycle = property(lambda s: s.L.akref_ncycle(s.h))
    nmb = property(lambda s: s.L.akref_nmb(s.h))
    nfofc = property(lambda s: s.L.akref_nfofc(s.h))

    def pack(self):
        pk = Pack()
        self.L.akref_pack(self.h, C.byref(pk))
        return pk

    def dims(self):
        p = self.params
        ng = p.ng
        n1 = p.mb_nx1 + 2*ng
        n2 = p.mb_nx2 + 2*ng if p.nx2 > 1 else 1
        n3 = p.mb_nx3 + 2*ng if p.nx3 > 1 else 1
        return n3, n2, n1

    def array(self, name):
        """numpy VIEW of an oracle array (shape per include/akmi.h)."""
        cnt = C.c_longlong(0)
        p = self.L.akref_array(self.h, name.encode(), C.byref(cnt))
        if not p or cnt.value == 0:
            raise KeyError(name)
        isint = name in ("nghbr", "bcs", "lloc", "counters")
        ct = C.c_int if isint else C.c_double
        buf = (ct*cnt.value).from_address(p)
        a = np.frombuffer(buf, dtype=np.int32 if isint else np.float64)
        n3, n2, n1 = self.dims()
        nmb = self.nmb
        fs = 1 if self.params.is_mhd else 0
        nv = (5 if self.params.is_ideal else 4) + self.params.nscalars
        shapes = {
            "u0": (nmb, nv, n3, n2, n1), "w0": (nmb, nv, n3, n2, n1), "u1": (nmb, nv, n3, n2, n1),
            "bcc0": (nmb, 3, n3, n2, n1),
            "b0x1f": (nmb, n3, n2, n1+1), "b0x2f": (nmb, n3, n2+1, n1), "b0x3f": (nmb, n3+1, n2, n1),
            "b1x1f": (nmb, n3, n2, n1+1), "b1x2f": (nmb, n3, n2+1, n1), "b1x3f": (nmb, n3+1, n2, n1),
            "flx1": (nmb, nv, n3, n2, n1+fs), "flx2": (nmb, nv, n3, n2+fs, n1),
            "flx3": (nmb, nv, n3+fs, n2, n1),
            "e1": (nmb, n3+1, n2+1, n1), "e2": (nmb, n3+1, n2, n1+1), "e3": (nmb, n3, n2+1, n1+1),
            "dx": (nmb, 3), "xminmax": (nmb, 6), "nghbr": (nmb, 27), "bcs": (nmb, 6),
            "lloc": (nmb, 3), "counters": (3,),
        }
        for e in ("e3x1", "e2x1", "e1x2", "e3x2", "e2x3", "e1x3"):
            shapes[e] = (nmb, n3, n2, n1)
        return a.reshape(shapes[name])

    def linear_wave_errors(self):
        out = (C.c_double*12)()
        n = self.L.akref_linear_wave_errors(self.h, out)
        return np.array(out[:n])

    def divb(self):
        out = (C.c_double*2)()
        self.L.akref_divb(self.h, out)
        return out[0], out[1]

    def totals(self):
        out = (C.c_double*5)()
        self.L.akref_totals(self.h, out)
        return np.array(out[:])
