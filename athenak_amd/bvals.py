"""MeshBoundaryValues: same-level ghost-zone exchange for cell- and face-centred variables.

Mirrors the roles of MeshBoundaryValuesCC/FC (src/bvals/bvals.hpp:215-267): PackAndSend*
fills same-rank neighbours directly (src/bvals/bvals_cc.cpp:122-135) and ships one
rank-packed message per peer rank per variable class (src/bvals/bvals.cpp:134-310,
bvals_cc.cpp:247-258); RecvAndUnpack* completes the receives and unpacks.  MPI_Isend/Irecv
become torch.distributed P2P ops (backend "nccl" = RCCL over xGMI; "gloo" in CPU tests); the
MPI_Test polling of the reference (bvals_cc.cpp:287-303) becomes a stream-ordered wait.

Segment order inside a peer message is fixed by (receiver gid, receiver direction), known
to both sides from the block tables alone, which replaces the reference's one-shot header
exchange (src/bvals/bvals.cpp:248-270).
"""
import ctypes as C

import numpy as np
import torch

from . import capi
from .tasklist import TaskStatus


class HipBvalsKernels:
    """pack / unpack / local-fill / BC kernels through the C ABI (device tensors only)."""

    def __init__(self):
        self.L = capi.lib()

    def cc_segsize(self, pack, d):
        return int(self.L.akmi_bvals_cc_segsize(C.byref(pack), d))

    def fc_segsize(self, pack, d):
        return int(self.L.akmi_bvals_fc_segsize(C.byref(pack), d))

    def cc_local(self, pack, nvar, nghbr, u):
        capi.check(self.L.akmi_bvals_cc_local(C.byref(pack), nvar, capi._p(nghbr), capi._p(u),
                                              capi._stream()), "bvals_cc_local")

    def cc_local_bcs(self, pack, nvar, nghbr, bcs, u, u_in=None, dt3_reset=None):
        """the gather and the physical boundary functions in one launch (packs without off-rank neighbours)"""
        capi.check(self.L.akmi_bvals_cc_local_bcs(C.byref(pack), nvar, capi._p(nghbr), capi._p(bcs), capi._p(u_in),
                                                  capi._p(u), capi._p(dt3_reset), capi._stream()), "bvals_cc_local_bcs")

    def fc_local_bcs(self, pack, nghbr, bcs, b1, b2, b3, b_in=None):
        capi.check(self.L.akmi_bvals_fc_local_bcs(C.byref(pack), capi._p(nghbr), capi._p(bcs), capi._p(b_in), capi._p(b1),
                                                  capi._p(b2), capi._p(b3), capi._stream()), "bvals_fc_local_bcs")

    def cc_pack(self, pack, nvar, nsend, tab, off, u, buf):
        capi.check(self.L.akmi_bvals_cc_pack(C.byref(pack), nvar, nsend, capi._p(tab), capi._p(off),
                                             capi._p(u), capi._p(buf), capi._stream()), "bvals_cc_pack")

    def cc_unpack(self, pack, nvar, nghbr, seg_off, buf, u):
        capi.check(self.L.akmi_bvals_cc_unpack(C.byref(pack), nvar, capi._p(nghbr), capi._p(seg_off),
                                               capi._p(buf), capi._p(u), capi._stream()), "bvals_cc_unpack")

    def fc_local(self, pack, nghbr, b1, b2, b3):
        capi.check(self.L.akmi_bvals_fc_local(C.byref(pack), capi._p(nghbr), capi._p(b1), capi._p(b2),
                                              capi._p(b3), capi._stream()), "bvals_fc_local")

    def fc_pack(self, pack, nsend, tab, off, b1, b2, b3, buf):
        capi.check(self.L.akmi_bvals_fc_pack(C.byref(pack), nsend, capi._p(tab), capi._p(off),
                                             capi._p(b1), capi._p(b2), capi._p(b3), capi._p(buf),
                                             capi._stream()), "bvals_fc_pack")

    def fc_unpack(self, pack, nghbr, seg_off, buf, b1, b2, b3):
        capi.check(self.L.akmi_bvals_fc_unpack(C.byref(pack), capi._p(nghbr), capi._p(seg_off),
                                               capi._p(buf), capi._p(b1), capi._p(b2), capi._p(b3),
                                               capi._stream()), "bvals_fc_unpack")

    def _dirs(self, bcs):
        """bit d set: some MeshBlock has a physical boundary across direction d (anything but block / periodic); the
        other directions are not launched (akmi_*_bcs_dirs).  Read from the flag table once per table."""
        mask = getattr(bcs, "_akmi_bc_dirs", None)      # kept ON the table (an address can be reused by the allocator for
        if mask is None:                                # the next mesh's table; the tensor object cannot)
            f = bcs.detach().cpu().numpy().reshape(-1, 6)
            phys = (f != capi.BC["block"]) & (f != capi.BC["periodic"])
            mask = sum(1 << q for q in range(3) if phys[:, 2*q:2*q + 2].any())
            bcs._akmi_bc_dirs = mask
        return mask

    def hydro_bcs(self, pack, nvar, bcs, u, u_in=None):
        capi.check(self.L.akmi_hydro_bcs_dirs(C.byref(pack), nvar, capi._p(bcs), self._dirs(bcs), capi._p(u_in),
                                              capi._p(u), capi._stream()), "hydro_bcs")

    def bfield_bcs(self, pack, bcs, b1, b2, b3, b_in=None):
        capi.check(self.L.akmi_bfield_bcs_dirs(C.byref(pack), capi._p(bcs), self._dirs(bcs), capi._p(b_in), capi._p(b1),
                                               capi._p(b2), capi._p(b3), capi._stream()), "bfield_bcs")


class HaloProfile:
    """Where a multi-rank stage of the Python host spends its exchange (the mirror of akmi_comm_profile of the C++ host):
    device time of the pack and unpack kernels (event pairs), the exposed wait for the receives -- a stall of the compute
    stream with RCCL (event pair round the wait), host time spent blocked in wait() with a host-staged transport -- and the
    dt all-reduce of a cycle.  Switched on by bench.py for its timed loop; None otherwise."""

    def __init__(self):
        self.pairs = {"pack": [], "wait": [], "unpack": []}
        self.wait_host_s = 0.0
        self.dt_reduce_s = 0.0
        self.dt_calls = 0
        self.bytes = 0
        self.posts = 0
        self.peers = 0

    def mark(self, cat):
        import torch
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.pairs[cat].append((a, b))
        a.record()
        return b

    def summary(self, nstages, ncycles):
        import torch
        torch.cuda.synchronize()
        ms = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in self.pairs.items()}
        ns, nc = max(nstages, 1), max(ncycles, 1)
        return {"pack_ms": round(ms["pack"]/ns, 5), "exposed_wait_ms": round((ms["wait"] + self.wait_host_s*1e3)/ns, 5),
                "unpack_ms": round(ms["unpack"]/ns, 5), "dt_reduce_ms": round(self.dt_reduce_s*1e3/nc, 5),
                "bytes_sent_per_stage": int(self.bytes/ns), "peers": int(self.peers), "rccl_ranks": None,
                "posts_per_stage": round(self.posts/ns, 3),
                "event_pairs": {k: len(v) for k, v in self.pairs.items()},
                "what": "Python host, rank 0, timed loop: HIP event pairs round the pack / unpack kernels and round the wait "
                        "for the receives (RCCL: the stall of the compute stream; host-staged transport: plus the host time "
                        "blocked in wait()); dt all-reduce: host time of the call per cycle"}


HALO_PROF = None        # a HaloProfile while bench.py times its loop


class _Channel:
    """Send/recv plan of one variable class (CC or FC)."""

    def __init__(self):
        self.nsend = 0
        self.send_tab = self.send_off = self.seg_off = None
        self.sendbuf = self.recvbuf = None
        self.h_send = self.h_recv = None     # pinned host mirrors (host-staged transport only)
        self.send_slices = {}   # peer -> (start, stop) in sendbuf
        self.recv_slices = {}
        self.works = []


class MeshBoundaryValues:
    """Ghost-zone exchange + physical BCs for one MeshBlockPack."""

    def __init__(self, ppack, kernels=None, device=None):
        device = device or capi.DEVICE
        self.pmy_pack = ppack
        self.k = kernels if kernels is not None else HipBvalsKernels()
        self.device = device
        pm = ppack.pmesh
        pmb = ppack.pmb
        self.my_rank, self.nranks = pm.my_rank, pm.nranks
        self.pack_c = None      # akmi_pack set by physics (needs device dx)
        nmb = ppack.nmb_thispack
        gids = ppack.gids
        # device neighbour table: >=0 local index, -1 none, <=-2 remote slot
        tab = -np.ones((nmb, 27), dtype=np.int32)
        recv_items = {}   # peer -> list of (my gid, o)
        send_items = {}   # peer -> list of (receiver gid, receiver o, my local m, d)
        # multilevel meshes: the exchange is MeshBoundaryValuesSMR's (bvals_smr.py); this object
        # then only applies the physical boundary conditions
        for m in range(nmb if not pm.multilevel else 0):
            for d in range(27):
                g, r = int(pmb.nghbr_gid[m, d]), int(pmb.nghbr_rank[m, d])
                if g < 0:
                    continue
                if r == self.my_rank:
                    tab[m, d] = g - gids
                else:
                    recv_items.setdefault(r, []).append((gids + m, d))
                    send_items.setdefault(r, []).append((g, 26 - d, m, d))
        self.peers = sorted(set(recv_items) | set(send_items))
        self._recv_items = {r: sorted(v) for r, v in recv_items.items()}
        self._send_items = {r: sorted(v) for r, v in send_items.items()}
        # remote slots numbered in (peer, gid, o) order
        self._slots = []
        for r in self.peers:
            for (g, o) in self._recv_items.get(r, []):
                tab[g - gids, o] = -(len(self._slots) + 2)
                self._slots.append((r, g, o))
        self.nghbr_host = tab
        self.nghbr = torch.from_numpy(tab.copy()).to(device)
        self.bcs = torch.from_numpy(np.ascontiguousarray(pmb.mb_bcs)).to(device)
        self.cc = None
        self.fc = None
        # single-rank uniform meshes with physical boundaries: the gather applies the boundary functions as well
        # (akmi_bvals_*_local_bcs); HydroBCs / BFieldBCs then have nothing left to do.  AKMI_FOLD_BCS=0: separate kernels.
        import os
        self.fold_bcs = (not self.peers and not pm.multilevel and not pm.strictly_periodic
                         and os.environ.get("AKMI_FOLD_BCS", "1") != "0" and hasattr(getattr(self.k, "L", None), "akmi_bvals_cc_local_bcs"))
        self._u_bcs_done = self._b_bcs_done = False

    # ------------------------------------------------------------------------------
    def _plan(self, segsize):
        ch = _Channel()
        if not self.peers:
            return ch
        gids = self.pmy_pack.gids
        # receive side
        seg_off, off = [], 0
        for r in self.peers:
            start = off
            for (g, o) in self._recv_items.get(r, []):
                seg_off.append(off)
                off += segsize(o)
            ch.recv_slices[r] = (start, off)
        nrecv = off
        # send side
        tab, soff, off = [], [], 0
        for r in self.peers:
            start = off
            for (g, o, m, d) in self._send_items.get(r, []):
                tab.append((m, d))
                soff.append(off)
                off += segsize(26 - d)   # == receiver's region for o = 26-d
            ch.send_slices[r] = (start, off)
        ch.nsend = len(tab)
        dev = self.device
        ch.send_tab = torch.tensor(tab, dtype=torch.int32, device=dev).reshape(-1, 2).contiguous()
        ch.send_off = torch.tensor(soff, dtype=torch.int64, device=dev)
        ch.seg_off = torch.tensor(seg_off, dtype=torch.int64, device=dev)
        ch.sendbuf = torch.empty(max(off, 1), dtype=torch.float64, device=dev)
        ch.recvbuf = torch.empty(max(nrecv, 1), dtype=torch.float64, device=dev)
        return ch

    def set_pack(self, pack_c, nvar):
        self.pack_c = pack_c
        self.nvar = nvar
        # inflow states (bvals.cpp:323-326), [variable][BoundaryFace]; a problem generator fills them
        self.u_in = torch.zeros((nvar, 6), dtype=torch.float64, device=self.device)
        self.b_in = torch.zeros((3, 6), dtype=torch.float64, device=self.device)
        self.cc = self._plan(lambda d: nvar*self.k.cc_segsize(pack_c, d))
        self.fc = self._plan(lambda d: self.k.fc_segsize(pack_c, d))

    # ------------------------------------------------------------------------------
    def _staged(self):
        """device buffers + a transport that cannot move device memory (anything but RCCL):
        stage the messages through pinned host buffers, like the reference built without
        GPU-aware MPI"""
        import torch.distributed as dist
        return torch.device(self.device).type == "cuda" and dist.get_backend() != "nccl"

    def _post(self, ch):
        import torch.distributed as dist
        sbuf, rbuf = ch.sendbuf, ch.recvbuf
        if self._staged():
            if ch.h_send is None:
                ch.h_send = torch.empty(ch.sendbuf.shape, dtype=torch.float64, pin_memory=True)
                ch.h_recv = torch.empty(ch.recvbuf.shape, dtype=torch.float64, pin_memory=True)
            ch.h_send.copy_(ch.sendbuf, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            sbuf, rbuf = ch.h_send, ch.h_recv
        ops = []
        for r in self.peers:
            a, b = ch.recv_slices[r]
            if b > a:
                ops.append(dist.P2POp(dist.irecv, rbuf[a:b], r))
        for r in self.peers:
            a, b = ch.send_slices[r]
            if b > a:
                ops.append(dist.P2POp(dist.isend, sbuf[a:b], r))
                if HALO_PROF is not None:
                    HALO_PROF.bytes += 8*(b - a)
        if HALO_PROF is not None:
            HALO_PROF.posts += 1
            HALO_PROF.peers = max(HALO_PROF.peers, len(self.peers))
        ch.works = dist.batch_isend_irecv(ops) if ops else []

    def _wait(self, ch):
        prof = HALO_PROF
        if prof is not None:
            import time as _t
            e1, t0 = prof.mark("wait"), _t.perf_counter()
        for w in ch.works:
            w.wait()
        ch.works = []
        if prof is not None:
            if ch.h_recv is not None:           # host-staged: wait() blocked the host, nothing stalled on the stream
                prof.wait_host_s += _t.perf_counter() - t0
            e1.record()
        if ch.h_recv is not None:
            ch.recvbuf.copy_(ch.h_recv, non_blocking=True)

    # ---- cell-centred ------------------------------------------------------------
    def PackAndSendCC(self, u, dt3_reset=None):
        """same-rank ghosts are final after this call; remote data is in flight.  Returns with _u_bcs_done set when the
        physical boundary functions were applied by the same launch (dt3_reset: that launch also resets the CFL minima)."""
        if self.fold_bcs:
            self.k.cc_local_bcs(self.pack_c, self.nvar, self.nghbr, self.bcs, u, self.u_in, dt3_reset)
            self._u_bcs_done = True
            return TaskStatus.complete
        self.k.cc_local(self.pack_c, self.nvar, self.nghbr, u)
        if self.peers:
            e1 = HALO_PROF.mark("pack") if HALO_PROF is not None else None
            self.k.cc_pack(self.pack_c, self.nvar, self.cc.nsend, self.cc.send_tab, self.cc.send_off,
                           u, self.cc.sendbuf)
            if e1 is not None:
                e1.record()
            self._post(self.cc)
        return TaskStatus.complete

    def RecvAndUnpackCC(self, u):
        if self.peers:
            self._wait(self.cc)
            e1 = HALO_PROF.mark("unpack") if HALO_PROF is not None else None
            self.k.cc_unpack(self.pack_c, self.nvar, self.nghbr, self.cc.seg_off, self.cc.recvbuf, u)
            if e1 is not None:
                e1.record()
        return TaskStatus.complete

    # ---- face-centred ------------------------------------------------------------
    def PackAndSendFC(self, b):
        if self.fold_bcs:
            self.k.fc_local_bcs(self.pack_c, self.nghbr, self.bcs, b.x1f, b.x2f, b.x3f, self.b_in)
            self._b_bcs_done = True
            return TaskStatus.complete
        self.k.fc_local(self.pack_c, self.nghbr, b.x1f, b.x2f, b.x3f)
        if self.peers:
            e1 = HALO_PROF.mark("pack") if HALO_PROF is not None else None
            self.k.fc_pack(self.pack_c, self.fc.nsend, self.fc.send_tab, self.fc.send_off,
                           b.x1f, b.x2f, b.x3f, self.fc.sendbuf)
            if e1 is not None:
                e1.record()
            self._post(self.fc)
        return TaskStatus.complete

    def RecvAndUnpackFC(self, b):
        if self.peers:
            self._wait(self.fc)
            e1 = HALO_PROF.mark("unpack") if HALO_PROF is not None else None
            self.k.fc_unpack(self.pack_c, self.nghbr, self.fc.seg_off, self.fc.recvbuf,
                             b.x1f, b.x2f, b.x3f)
            if e1 is not None:
                e1.record()
        return TaskStatus.complete

    # ---- physical boundaries -----------------------------------------------------
    def HydroBCs(self, u):
        if self._u_bcs_done:                 # applied by the gather of PackAndSendCC
            self._u_bcs_done = False
            return
        self.k.hydro_bcs(self.pack_c, self.nvar, self.bcs, u, self.u_in)

    def BFieldBCs(self, b):
        if self._b_bcs_done:
            self._b_bcs_done = False
            return
        self.k.bfield_bcs(self.pack_c, self.bcs, b.x1f, b.x2f, b.x3f, self.b_in)
