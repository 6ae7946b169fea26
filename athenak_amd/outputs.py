"""Outputs: the data formats on the far side of the hot path, written so that the reference's
own readers (vis/python/athena_read.py tab()/hst()/error_dat(), bin_convert.read_binary())
parse them unchanged.

Mirrors, for the file types a hydro/MHD run of this path uses:
  Outputs                src/outputs/outputs.cpp:47-304      (<outputN> blocks -> pout_list)
  BaseTypeOutput         src/outputs/basetype_output.cpp     (variable groups, slices, gather)
  FormattedTableOutput   src/outputs/formatted_table.cpp     (tab/<basename>.<id>.NNNNN.tab)
  HistoryOutput          src/outputs/history.cpp             (<basename>.hydro|mhd.hst)
  MeshBinaryOutput       src/outputs/binary.cpp              (bin/<basename>.<id>.NNNNN.bin)
  RestartOutput          src/outputs/restart.cpp             (rst/<basename>.NNNNN.rst) + read_restart()
The volume sums of the history file are reduced on the device (akmi_history_sums); everything
else here is host-side formatting of arrays copied from the device.  Other file types of the
reference (vtk, pdf, cart, sph, log, trk, cbin) are rejected loudly.
"""
import ctypes as C
import os
import struct

import numpy as np

from . import capi
from .mesh import CellCenterX


def _fatal(msg):
    raise RuntimeError("### FATAL ERROR " + msg)


def _cfmt(fmt, val):
    """printf with a C format string (the decks carry C formats such as %12.5e)"""
    return fmt % val


def CellCenterIndex(x, n, xmin, xmax):
    """src/coordinates/cell_locations.hpp:48-50"""
    return int(((x - xmin)/(xmax - xmin))*float(n))


class OutputParameters:
    """src/outputs/outputs.hpp: OutputParameters"""

    def __init__(self):
        self.block_number = 0
        self.block_name = ""
        self.last_time = -1.0
        self.dt = 0.0
        self.dcycle = 0
        self.file_number = 0
        self.file_basename = ""
        self.file_type = ""
        self.variable = ""
        self.file_id = ""
        self.include_gzs = False
        self.gid = -1
        self.slice1 = self.slice2 = self.slice3 = False
        self.slice_x1 = self.slice_x2 = self.slice_x3 = 0.0
        self.data_format = " %12.5e"
        self.user_hist_only = False


class OutputMeshBlockInfo:
    def __init__(self, gid, ois, oie, ojs, oje, oks, oke, size):
        self.mb_gid = gid
        self.ois, self.oie, self.ojs, self.oje, self.oks, self.oke = ois, oie, ojs, oje, oks, oke
        self.x1min, self.x1max = size.x1min, size.x1max
        self.x2min, self.x2max = size.x2min, size.x2max
        self.x3min, self.x3max = size.x3min, size.x3max


# variable groups of basetype_output.cpp:196-520 for Newtonian hydro/MHD without scalars:
# name -> list of (label, component, array)
def _outvars(variable, is_mhd, is_ideal=True):
    blk = "mhd" if is_mhd else "hydro"
    u = [("dens", 0, "u0"), ("mom1", 1, "u0"), ("mom2", 2, "u0"), ("mom3", 3, "u0"), ("ener", 4, "u0")]
    w = [("dens", 0, "w0"), ("velx", 1, "w0"), ("vely", 2, "w0"), ("velz", 3, "w0"), ("eint", 4, "w0")]
    if not is_ideal:                       # basetype_output.cpp:252-271: energies only if is_ideal
        u, w = u[:4], w[:4]
    b = [("bcc1", 0, "bcc0"), ("bcc2", 1, "bcc0"), ("bcc3", 2, "bcc0")]
    table = {blk + "_u": u, blk + "_w": w}
    for sfx, (lab, n, arr) in zip(("d", "m1", "m2", "m3", "e"), u):
        table["%s_u_%s" % (blk, sfx)] = [(lab, n, arr)]
    for sfx, (lab, n, arr) in zip(("d", "vx", "vy", "vz", "e"), w):
        table["%s_w_%s" % (blk, sfx)] = [(lab, n, arr)]
    if is_mhd:
        table.update({"mhd_bcc": b, "mhd_u_bcc": u + b, "mhd_w_bcc": w + b, "mhd_bcc1": b[0:1],
                      "mhd_bcc2": b[1:2], "mhd_bcc3": b[2:3]})
    if variable not in table:
        _fatal("Output variable '%s' not implemented on this path (choices: %s)"
               % (variable, ", ".join(sorted(table))))
    return table[variable]


class BaseTypeOutput:
    def __init__(self, pin, pm, op):
        self.out_params = op
        self.outvars = []
        self.outmbs = []
        self.outarray = None
        pk = pm.pmb_pack
        if op.file_type not in ("hst", "rst"):
            phys = pk.pmhd if pk.pmhd is not None else pk.phydro
            self.outvars = _outvars(op.variable, pk.pmhd is not None, phys.peos.eos_data.is_ideal)

    def LoadOutputData(self, pm):
        """basetype_output.cpp:729-862: per-block index ranges (ghost zones, slices) and a
        host copy of the selected components"""
        op = self.out_params
        ind = pm.mb_indcs
        pk = pm.pmb_pack
        phys = pk.pmhd if pk.pmhd is not None else pk.phydro
        self.outmbs = []
        for m in range(pk.nmb_thispack):
            if op.gid >= 0 and (m + pk.gids) != op.gid:
                continue
            size = pk.pmb.mb_size[m]
            if op.include_gzs:
                n3, n2, n1 = ind.ncells
                ois, oie, ojs, oje, oks, oke = 0, n1 - 1, 0, n2 - 1, 0, n3 - 1
            else:
                ois, oie, ojs, oje, oks, oke = ind.is_, ind.ie, ind.js, ind.je, ind.ks, ind.ke
            if op.slice1:
                if op.slice_x1 < size.x1min or op.slice_x1 >= size.x1max:
                    continue
                ois = oie = CellCenterIndex(op.slice_x1, ind.nx1, size.x1min, size.x1max) + ind.is_
            if op.slice2:
                if op.slice_x2 < size.x2min or op.slice_x2 >= size.x2max:
                    continue
                ojs = oje = CellCenterIndex(op.slice_x2, ind.nx2, size.x2min, size.x2max) + ind.js
            if op.slice3:
                if op.slice_x3 < size.x3min or op.slice_x3 >= size.x3max:
                    continue
                oks = oke = CellCenterIndex(op.slice_x3, ind.nx3, size.x3min, size.x3max) + ind.ks
            self.outmbs.append(OutputMeshBlockInfo(int(pk.pmb.mb_gid[m]), ois, oie, ojs, oje, oks,
                                                   oke, size))
        if not self.outmbs or not self.outvars:
            self.outarray = None
            return
        o = self.outmbs[0]
        shape = (len(self.outvars), len(self.outmbs), o.oke - o.oks + 1, o.oje - o.ojs + 1,
                 o.oie - o.ois + 1)
        out = np.empty(shape, dtype=np.float64)
        host = {}
        for n, (_, comp, arr) in enumerate(self.outvars):
            for mi, o in enumerate(self.outmbs):
                m = o.mb_gid - pk.gids
                key = (arr, m, comp)
                if key not in host:
                    host[key] = _to_numpy(getattr(phys, arr)[m, comp])
                out[n, mi] = host[key][o.oks:o.oke + 1, o.ojs:o.oje + 1, o.ois:o.oie + 1]
        self.outarray = out

    def _advance(self, pm, pin, numbered=True):
        op = self.out_params
        if numbered:
            op.file_number += 1
            pin.SetInteger(op.block_name, "file_number", op.file_number)
        if op.last_time < 0.0:
            op.last_time = pm.time
        else:
            op.last_time += op.dt
        pin.SetReal(op.block_name, "last_time", op.last_time)


def _to_numpy(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


def _barrier(pm):
    if pm.nranks > 1:
        import torch.distributed as dist
        dist.barrier()


class FormattedTableOutput(BaseTypeOutput):
    def __init__(self, pin, pm, op):
        super().__init__(pin, pm, op)
        if pm.multi_d and not (op.slice1 or op.slice2):
            _fatal("Formatted table outputs can only contain 1D slices\nPlease add additional slice planes")
        if pm.three_d and ((not op.slice2 and not op.slice3) or (not op.slice1 and not op.slice3)):
            _fatal("Formatted table outputs can only contain 1D slices\nPlease add additional slice planes")
        os.makedirs("tab", exist_ok=True)

    def WriteOutputFile(self, pm, pin):
        """formatted_table.cpp:52-195"""
        op = self.out_params
        fname = "tab/%s.%s.%05d.tab" % (op.file_basename, op.file_id, op.file_number)
        fmt = op.data_format
        if pm.my_rank == 0:
            with open(fname, "w") as f:
                f.write("# Athena++ data at time=%e" % pm.time)
                f.write("  cycle=%d \n" % pm.ncycle)
                f.write("# gid  ")
                if not op.slice1:
                    f.write(" i       x1v     ")
                if not op.slice2:
                    f.write(" j       x2v     ")
                if not op.slice3:
                    f.write(" k       x3v     ")
                for (label, _, _) in self.outvars:
                    f.write("    %s     " % label)
                f.write("\n")
        _barrier(pm)
        ind = pm.mb_indcs
        for r in range(pm.nranks):
            if r == pm.my_rank:
                with open(fname, "a") as f:
                    for mi, o in enumerate(self.outmbs):
                        for k in range(o.oks, o.oke + 1):
                            for j in range(o.ojs, o.oje + 1):
                                for i in range(o.ois, o.oie + 1):
                                    line = ["%05d" % o.mb_gid]
                                    if o.oie != o.ois:
                                        line.append(" %04d" % i)
                                        line.append(_cfmt(fmt, CellCenterX(i - ind.is_, ind.nx1, o.x1min, o.x1max)))
                                    if o.oje != o.ojs:
                                        line.append(" %04d" % j)
                                        line.append(_cfmt(fmt, CellCenterX(j - ind.js, ind.nx2, o.x2min, o.x2max)))
                                    if o.oke != o.oks:
                                        line.append(" %04d" % k)
                                        line.append(_cfmt(fmt, CellCenterX(k - ind.ks, ind.nx3, o.x3min, o.x3max)))
                                    for n in range(len(self.outvars)):
                                        line.append(_cfmt(fmt, self.outarray[n, mi, k - o.oks, j - o.ojs, i - o.ois]))
                                    f.write("".join(line) + "\n")
            _barrier(pm)
        self._advance(pm, pin)


class HistoryOutput(BaseTypeOutput):
    """history.cpp: volume sums of the conserved variables, kinetic and magnetic energies"""

    def __init__(self, pin, pm, op):
        super().__init__(pin, pm, op)
        pk = pm.pmb_pack
        self.is_mhd = pk.pmhd is not None
        self.labels = ["mass", "1-mom", "2-mom", "3-mom", "tot-E", "1-KE", "2-KE", "3-KE"]
        phys = pk.pmhd if self.is_mhd else pk.phydro
        if not phys.peos.eos_data.is_ideal:
            self.labels.remove("tot-E")
        if self.is_mhd:
            self.labels += ["1-ME", "2-ME", "3-ME"]
        self.hdata = None
        self.header_written = False

    def LoadOutputData(self, pm):
        """history.cpp:78-160,272-374: the sums run on the device (akmi_history_sums)"""
        import torch
        pk = pm.pmb_pack
        phys = pk.pmhd if self.is_mhd else pk.phydro
        out = torch.zeros(len(self.labels), dtype=torch.float64, device=phys.u0.device)
        L = capi.lib()
        if self.is_mhd:
            b = (capi._p(phys.b0.x1f), capi._p(phys.b0.x2f), capi._p(phys.b0.x3f))
        else:
            b = (None, None, None)
        capi.check(L.akmi_history_sums(C.byref(phys.pack_c), 1 if self.is_mhd else 0, capi._p(phys.u0),
                                       *b, capi._p(out), capi._stream()), "history_sums")
        self.hdata = out

    def WriteOutputFile(self, pm, pin):
        """history.cpp:381-457"""
        op = self.out_params
        h = self.hdata
        if pm.nranks > 1:
            import torch.distributed as dist
            if dist.get_backend() != "nccl":
                h = h.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)      # MPI_Reduce(MPI_SUM) to rank 0
        h = h.cpu().numpy()
        if pm.my_rank == 0:
            fname = "%s.%s.hst" % (op.file_basename, "mhd" if self.is_mhd else "hydro")
            with open(fname, "a") as f:
                if not self.header_written:
                    f.write("# Athena++ history data\n")
                    f.write("#  [%d]=time      " % 1)
                    f.write("[%d]=dt       " % 2)
                    for n, lab in enumerate(self.labels):
                        f.write("[%d]=%.10s    " % (n + 3, lab))
                    f.write("\n")
                    self.header_written = True
                f.write(_cfmt(op.data_format, pm.time))
                f.write(_cfmt(op.data_format, pm.dt))
                for v in h:
                    f.write(_cfmt(op.data_format, float(v)))
                f.write("\n")
        self._advance(pm, pin, numbered=False)


class MeshBinaryOutput(BaseTypeOutput):
    def __init__(self, pin, pm, op):
        super().__init__(pin, pm, op)
        if pin.GetOrAddBoolean(op.block_name, "single_file_per_rank", False):
            _fatal("bin output: single_file_per_rank is not implemented on this path")
        os.makedirs("bin", exist_ok=True)

    def WriteOutputFile(self, pm, pin):
        """binary.cpp:52-319: text pre-header, parameter dump, then per MeshBlock 10 int32
        (ois,oie,ojs,oje,oks,oke,lx1,lx2,lx3,level), 6 Real (block extent) and the variables as
        float32 [nvar][k][j][i]"""
        op = self.out_params
        fname = "bin/%s.%s.%05d.bin" % (op.file_basename, op.file_id, op.file_number)
        msg = ("Athena binary output version=1.1\n  size of preheader=5\n  time=%.16e\n  cycle=%d\n"
               "  size of location=8\n  size of variable=4\n  number of variables=%d\n  variables:  "
               % (pm.time, pm.ncycle, len(self.outvars)))
        msg += "".join("%s  " % lab for (lab, _, _) in self.outvars) + "\n"
        dump = pin.ParameterDump()
        hdr = (msg + "  header offset=%d\n" % len(dump) + dump).encode("ascii")
        if pm.my_rank == 0:
            with open(fname, "wb") as f:
                f.write(hdr)
        _barrier(pm)
        for r in range(pm.nranks):
            if r == pm.my_rank and self.outmbs:
                with open(fname, "ab") as f:
                    for mi, o in enumerate(self.outmbs):
                        l1, l2, l3 = pm.lloc_eachmb[o.mb_gid][:3]
                        # binary.cpp:192-193: loc.level - root_level (0 on uniform meshes, >0 inside refined regions)
                        f.write(struct.pack("<10i", o.ois, o.oie, o.ojs, o.oje, o.oks, o.oke, l1, l2, l3,
                                            pm.level_of(o.mb_gid) - pm.root_level))
                        f.write(struct.pack("<6d", o.x1min, o.x1max, o.x2min, o.x2max, o.x3min, o.x3max))
                        f.write(np.ascontiguousarray(self.outarray[:, mi], dtype="<f4").tobytes())
            _barrier(pm)
        self._advance(pm, pin)


class RestartOutput(BaseTypeOutput):
    """restart.cpp:37-560: one file rst/<basename>.<NNNNN>.rst holding the parameter dump, the mesh
    header, the logical locations and costs of all MeshBlocks and, per MeshBlock (in gid order,
    `data_size` bytes each), the full-precision dependent variables INCLUDING ghost zones:
    [hydro u0][mhd u0][b0.x1f][b0.x2f][b0.x3f].  read_restart() below is the inverse."""

    def __init__(self, pin, pm, op):
        super().__init__(pin, pm, op)
        if pin.GetOrAddBoolean(op.block_name, "single_file_per_rank", False):
            _fatal("single_file_per_rank restart files are not on this path")
        if pm.my_rank == 0:
            os.makedirs("rst", exist_ok=True)

    def LoadOutputData(self, pm):
        """restart.cpp:53-137: everything is taken from the physics arrays at write time"""

    def WriteOutputFile(self, pm, pin):
        op = self.out_params
        fname = os.path.join("rst", "%s.%05d.rst" % (op.file_basename, op.file_number))
        # counters advance first so that the values for the NEXT dump are in the file (restart.cpp:193-200)
        self._advance(pm, pin)
        sbuf = pin.ParameterDump().encode()
        pk = pm.pmb_pack
        arrays = []               # per physics: list of tensors whose [m] slices form one record
        if pk.phydro is not None:
            arrays.append(pk.phydro.u0)
        if pk.pmhd is not None:
            arrays += [pk.pmhd.u0, pk.pmhd.b0.x1f, pk.pmhd.b0.x2f, pk.pmhd.b0.x3f]
        data_size = sum(int(a[0].numel())*8 for a in arrays)
        header = bytearray()
        header += struct.pack("<ii", pm.nmb_total, _root_level(pm))
        header += _pack_region_size(pm.mesh_size, pm.mesh_indcs)
        header += _pack_region_indcs(pm.mesh_indcs, coarse=False)
        header += _pack_region_indcs(pm.mb_indcs, coarse=True)
        header += struct.pack("<ddi", pm.time, pm.dt, pm.ncycle)
        for gid, l in enumerate(pm.lloc_eachmb):              # LogicalLocation {lx1,lx2,lx3,level}: the block's own level
            header += struct.pack("<iiii", l[0], l[1], l[2], pm.level_of(gid))
        header += np.asarray(pm.cost_eachmb, dtype="<f4").tobytes()
        header += struct.pack("<Q", data_size)
        base = len(sbuf) + len(header)
        if pm.my_rank == 0:
            with open(fname, "wb") as f:
                f.write(sbuf)
                f.write(header)
                f.truncate(base + data_size*pm.nmb_total)
        _barrier(pm)
        with open(fname, "r+b") as f:
            for m in range(pk.nmb_thispack):
                f.seek(base + data_size*(pk.gids + m))
                for a in arrays:
                    f.write(np.ascontiguousarray(_to_numpy(a[m]), dtype="<f8").tobytes())
        _barrier(pm)


def _root_level(pm):
    """build_tree.cpp:43-44"""
    nmbmax = max(pm.nmb_rootx1, pm.nmb_rootx2, pm.nmb_rootx3)
    lev = 0
    while (1 << lev) < nmbmax:
        lev += 1
    return lev


def _pack_region_size(ms, ind):
    """struct RegionSize {x1min,x2min,x3min,x1max,x2max,x3max,dx1,dx2,dx3} (mesh.hpp:25-29)"""
    dx = ((ms.x1max - ms.x1min)/float(ind.nx1), (ms.x2max - ms.x2min)/float(ind.nx2),
          (ms.x3max - ms.x3min)/float(ind.nx3))
    return struct.pack("<9d", ms.x1min, ms.x2min, ms.x3min, ms.x1max, ms.x2max, ms.x3max, *dx)


def _pack_region_indcs(ind, coarse):
    """struct RegionIndcs (mesh.hpp:35-41): 19 ints; the coarse members are only set for MeshBlocks
    (mesh.cpp:286-330)"""
    v = [ind.ng, ind.nx1, ind.nx2, ind.nx3, ind.is_, ind.ie, ind.js, ind.je, ind.ks, ind.ke]
    if coarse:
        cjs = ind.ng if ind.nx2 > 1 else 0
        cks = ind.ng if ind.nx3 > 1 else 0
        v += [ind.cnx1, ind.cnx2, ind.cnx3, ind.ng, ind.ng + ind.cnx1 - 1,
              cjs, cjs + ind.cnx2 - 1 if ind.nx2 > 1 else 0, cks, cks + ind.cnx3 - 1 if ind.nx3 > 1 else 0]
    else:
        v += [0]*9
    return struct.pack("<19i", *v)


_RST_HEADER = 2*4 + 9*8 + 2*19*4 + 2*8 + 4          # restart.cpp:296-297 "step1size" without the dump


def read_restart(path):
    """Inverse of RestartOutput: returns (parameter text, header dict, per-gid record reader).
    Mirrors ParameterInput::LoadFromFile (stop at <par_end>), Mesh::BuildTreeFromRestart
    (build_tree.cpp:315-370) and the restart constructor of ProblemGenerator (pgen.cpp:97-330)."""
    with open(path, "rb") as f:
        blob = f.read(1 << 16)
    end = blob.find(b"<par_end>")
    if end < 0:
        _fatal("<par_end> is not found in the first 64KBytes of restart file " + path)
    end = blob.index(b"\n", end) + 1
    text = blob[:end].decode()
    with open(path, "rb") as f:
        f.seek(end)
        h = f.read(_RST_HEADER)
        nmb_total, root_level = struct.unpack_from("<ii", h, 0)
        mesh_size = struct.unpack_from("<9d", h, 8)
        mesh_indcs = struct.unpack_from("<19i", h, 80)
        mb_indcs = struct.unpack_from("<19i", h, 156)
        time, dt, ncycle = struct.unpack_from("<ddi", h, 232)
        lloc = np.frombuffer(f.read(16*nmb_total), dtype="<i4").reshape(nmb_total, 4).copy()
        cost = np.frombuffer(f.read(4*nmb_total), dtype="<f4").copy()
        (data_size,) = struct.unpack("<Q", f.read(8))
        base = f.tell()
    hdr = dict(nmb_total=nmb_total, root_level=root_level, mesh_size=mesh_size, mesh_indcs=mesh_indcs,
               mb_indcs=mb_indcs, time=time, dt=dt, ncycle=ncycle, lloc=lloc, cost=cost,
               data_size=data_size, data_offset=base)

    def record(gid):
        with open(path, "rb") as f:
            f.seek(base + data_size*gid)
            return np.frombuffer(f.read(data_size), dtype="<f8")
    return text, hdr, record


class Outputs:
    """outputs.cpp:47-304"""

    def __init__(self, pin, pm):
        self.pout_list = []
        num_hst = num_rst = 0
        for name in list(pin.blocks):
            if not name.startswith("output"):
                continue
            op = OutputParameters()
            op.block_number = int(name[6:] or 0)
            op.block_name = name
            op.last_time = pin.GetOrAddReal(name, "last_time", -1.0)
            if pin.DoesParameterExist(name, "dcycle"):
                op.dcycle = pin.GetInteger(name, "dcycle")
                op.dt = 0.0
            else:
                op.dt = pin.GetReal(name, "dt")
                op.dcycle = 0
            if op.dcycle == 0 and op.dt <= 0.0:
                continue
            op.file_number = pin.GetOrAddInteger(name, "file_number", 0)
            op.file_basename = pin.GetString("job", "basename")
            op.file_type = pin.GetString(name, "file_type")
            if op.file_type not in ("hst", "rst", "log", "trk"):
                op.variable = pin.GetString(name, "variable")
                op.file_id = pin.GetOrAddString(name, "id", op.variable)
            op.include_gzs = pin.GetOrAddBoolean(name, "ghost_zones", False)
            op.gid = pin.GetOrAddInteger(name, "gid", -1)
            if op.gid >= 0 and pm.nmb_total == 1:
                _fatal("Cannot specify MeshBlock ID in output block '%s' when there is only one" % name)
            if op.gid > pm.nmb_total - 1:
                _fatal("MeshBlock gid=%d in output block '%s' exceeds total number of MeshBlocks"
                       % (op.gid, name))
            ms = pm.mesh_size
            for q, lo, hi in ((1, ms.x1min, ms.x1max), (2, ms.x2min, ms.x2max), (3, ms.x3min, ms.x3max)):
                key = "slice_x%d" % q
                if pin.DoesParameterExist(name, key):
                    x = pin.GetReal(name, key)
                    if not (lo <= x < hi):
                        _fatal("Slice at x%d=%g in output block '%s' is out of range of Mesh" % (q, x, name))
                    setattr(op, key, x)
                    setattr(op, "slice%d" % q, True)
            if op.file_type == "hst":
                op.user_hist_only = pin.GetOrAddBoolean(name, "user_hist_only", False)
                if op.user_hist_only:
                    _fatal("user history functions are not on this path")
            op.data_format = " " + pin.GetOrAddString(name, "data_format", "%12.5e")
            if op.file_type == "tab":
                self.pout_list.insert(0, FormattedTableOutput(pin, pm, op))
            elif op.file_type == "hst":
                self.pout_list.insert(0, HistoryOutput(pin, pm, op))
                num_hst += 1
            elif op.file_type == "bin":
                self.pout_list.insert(0, MeshBinaryOutput(pin, pm, op))
            elif op.file_type == "rst":
                # tail end of the list, so that the file counters of the other output types are
                # up to date in the restart file (outputs.cpp:285-292)
                self.pout_list.append(RestartOutput(pin, pm, op))
                num_rst += 1
            else:
                _fatal("Unrecognized or unsupported file format = '%s' in output block '%s' "
                       "(tab, hst, bin, rst on this path)" % (op.file_type, name))
        if num_hst > 1 or num_rst > 1:
            _fatal("More than one history or restart output block found in input file")

    def MakeOutputs(self, pm, pin):
        for out in self.pout_list:
            out.LoadOutputData(pm)
            out.WriteOutputFile(pm, pin)

    def TestAndMakeOutputs(self, pm, pin, tlim):
        """driver.cpp:432-445 (comparison at 32-bit precision, as the reference)"""
        time_32 = np.float32(pm.time)
        tlim_32 = np.float32(tlim)
        for out in self.pout_list:
            op = out.out_params
            next_32 = np.float32(op.last_time + op.dt)
            if ((op.dt > 0.0 and time_32 >= next_32 and time_32 < tlim_32) or
                    (op.dcycle > 0 and pm.ncycle % op.dcycle == 0)):
                out.LoadOutputData(pm)
                out.WriteOutputFile(pm, pin)
