"""NativeSimulation: the C++ host driver (csrc/akmi_host.cpp) behind the same Python surface
as main.Simulation.  Mesh, MeshBlockPack, TaskList, Hydro/MHD tasks and the Driver loop run
in C++ (akmi_sim_*); Python only evaluates the closed-form initial conditions into the native
device arrays (aliased as torch tensors through __cuda_array_interface__)."""
import ctypes as C

import torch

from . import capi
from .mesh import Mesh
from .pgen import ProblemGenerator


class _DevAlias:
    """exposes a raw device pointer to torch (zero-copy)"""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (ptr, False),
                                         "version": 3, "strides": None}


class _Face:
    pass


class _Eos:
    def __init__(self, pin, blk):
        ideal = pin.GetString(blk, "eos") == "ideal"
        self.eos_data = type("EOS_Data", (), {
            "gamma": pin.GetReal(blk, "gamma") if ideal else 0.0, "is_ideal": ideal,
            "iso_cs": 0.0 if ideal else pin.GetReal(blk, "iso_sound_speed")})()


class _PhysAlias:
    """the attributes ProblemGenerator needs, backed by the native arrays"""


class NativeSimulation:
    def __init__(self, pin, initialize=True):
        self.pin = pin
        self.L = capi.lib()
        stream = capi._stream()
        h = self.L.akmi_sim_create(pin.Dump().encode(), stream)
        if not h:       # a run-time failure (device allocation, HIP); deck errors exit with "### FATAL ERROR" as the reference does
            raise capi.AkmiError("akmi_sim_create failed: %s" % self.L.akmi_last_error().decode())
        self.h = C.c_void_p(h)
        # a Python Mesh of the same deck gives the problem generators their coordinates; with a
        # communicator (akmi_comm_init_*) both sides cut the block list the same way
        # (tests/test_host_plan.py) and the arrays are those of this rank's pack
        self.pmesh = Mesh(pin, my_rank=self.L.akmi_comm_rank(), nranks=self.L.akmi_comm_nranks())
        assert self.pmesh.pmb_pack.gids == self.L.akmi_sim_gids(self.h)
        is_mhd = pin.DoesBlockExist("mhd")
        blk = "mhd" if is_mhd else "hydro"
        n3, n2, n1 = self.pmesh.mb_indcs.ncells
        nmb = self.pmesh.pmb_pack.nmb_thispack
        assert nmb == self.L.akmi_sim_nmb_thisrank(self.h)
        ph = _PhysAlias()
        ph.peos = _Eos(pin, blk)
        ph.nfluid = 5 if ph.peos.eos_data.is_ideal else 4
        ph.nscalars = pin.GetOrAddInteger(blk, "nscalars", 0)
        ph.nvars = nv = ph.nfluid + ph.nscalars
        self._shape = (is_mhd, nmb, nv, n3, n2, n1)
        self._phys = ph
        self._refresh()
        if is_mhd:
            self.pmesh.pmb_pack.pmhd = ph
        else:
            self.pmesh.pmb_pack.phydro = ph
        self.pmesh.pgen = ProblemGenerator(pin, self.pmesh)
        if initialize:
            self.Initialize()

    def _refresh(self):
        """(re)alias the native arrays: the C++ host swaps its two registers (u0/u1, b0/b1) after an
        out-of-place first stage, so the pointers behind the names change from call to call"""
        is_mhd, nmb, nv, n3, n2, n1 = self._shape
        ph = self._phys
        ph.u0 = self._alias("u0", (nmb, nv, n3, n2, n1))
        ph.w0 = self._alias("w0", (nmb, nv, n3, n2, n1))
        ph.u1 = self._alias("u1", (nmb, nv, n3, n2, n1))
        if is_mhd:
            ph.bcc0 = self._alias("bcc0", (nmb, 3, n3, n2, n1))
            for reg in ("b0", "b1"):
                f = _Face()
                f.x1f = self._alias(reg + "x1f", (nmb, n3, n2, n1 + 1))
                f.x2f = self._alias(reg + "x2f", (nmb, n3, n2 + 1, n1))
                f.x3f = self._alias(reg + "x3f", (nmb, n3 + 1, n2, n1))
                setattr(ph, reg, f)

    def _alias(self, name, shape):
        cnt = C.c_longlong(0)
        p = self.L.akmi_sim_array(self.h, name.encode(), C.byref(cnt))
        n = 1
        for s in shape:
            n *= s
        assert p and cnt.value == n, (name, cnt.value, n)
        return torch.as_tensor(_DevAlias(p, shape), device="cuda")

    @property
    def phys(self):
        return self._phys

    def Initialize(self):
        torch.cuda.synchronize()
        capi.check(self.L.akmi_sim_initialize(self.h, C.c_double(self.pin.GetReal("time", "tlim"))),
                   "sim_initialize")

    def Execute(self, max_cycles=None):
        n = capi.check(self.L.akmi_sim_execute(self.h, -1 if max_cycles is None else int(max_cycles)), "sim_execute")
        self._refresh()
        self.pmesh.time = self.time
        self.pmesh.dt = self.dt
        self.pmesh.ncycle = self.ncycle
        return n

    time = property(lambda s: s.L.akmi_sim_time(s.h))
    dt = property(lambda s: s.L.akmi_sim_dt(s.h))
    tlim = property(lambda s: s.L.akmi_sim_tlim(s.h))
    ncycle = property(lambda s: s.L.akmi_sim_ncycle(s.h))

    def close(self):
        if self.h:
            self.L.akmi_sim_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- ranks ------------------------------------------------------------------------------------------
_KEEP = []      # callbacks handed to C must outlive the communicator


def init_comm_from_torch_distributed():
    """Give the C++ host a communicator for the ranks of the initialised torch.distributed job.
    Backend nccl (= RCCL, one GPU per rank): rank 0 creates the RCCL id, a broadcast hands it to
    every rank, and from there on the C++ host calls RCCL itself (ncclSend/ncclRecv/ncclAllReduce).
    Any other backend: the C++ host stages its messages through pinned host memory and this
    module moves them with torch.distributed point-to-point operations (tests: gloo, two ranks
    sharing the only GPU of the box, which RCCL refuses)."""
    import numpy as np
    import torch.distributed as dist
    L = capi.lib()
    rank, world = dist.get_rank(), dist.get_world_size()
    if dist.get_backend() == "nccl":
        idb = C.create_string_buffer(128)
        if rank == 0:
            capi.check(L.akmi_comm_unique_id(idb), "comm_unique_id")
        t = torch.frombuffer(bytearray(idb.raw), dtype=torch.uint8).cuda()
        dist.broadcast(t, 0)
        capi.check(L.akmi_comm_init_rccl(rank, world, bytes(t.cpu().numpy().tobytes())), "comm_init_rccl")
        return "rccl"

    EX = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p),
                     C.POINTER(C.c_longlong), C.POINTER(C.c_void_p), C.POINTER(C.c_longlong))
    AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)

    def view(ptr, n):
        return torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(n,)))

    def exchange(user, npeer, peers, sp, sc, rp, rc):
        try:
            ops = []
            for q in range(npeer):
                if rc[q] > 0:
                    ops.append(dist.P2POp(dist.irecv, view(rp[q], rc[q]), peers[q]))
            for q in range(npeer):
                if sc[q] > 0:
                    ops.append(dist.P2POp(dist.isend, view(sp[q], sc[q]), peers[q]))
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
            return 0
        except Exception:      # never unwind through C
            import traceback
            traceback.print_exc()
            return 1

    def allreduce_min(user, vals, n):
        try:
            t = view(vals, n)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    ex, ar = EX(exchange), AR(allreduce_min)
    _KEEP[:] = [ex, ar]
    capi.check(L.akmi_comm_init_callbacks(rank, world, ex, ar, None), "comm_init_callbacks")
    return "callbacks"


def finalize_comm():
    capi.lib().akmi_comm_finalize()
    _KEEP[:] = []
