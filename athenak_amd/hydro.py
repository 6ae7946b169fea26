"""hydro::Hydro -- arrays and task member functions of the hydrodynamics module.

Mirror of src/hydro/hydro.hpp:73-154 / hydro.cpp:29-301 / hydro_tasks.cpp:48-508: same
member names (u0, w0, u1, uflx, peos->eos_data, pbval_u, dtnew), same task names and
dependency chain (AssembleHydroTasks, hydro_tasks.cpp:48-80), each task body reduced to one
call through the C ABI (include/akmi.h).  Device arrays are torch.float64 CUDA tensors in the
reference's (m,n,k,j,i) LayoutRight order; torch is memory/stream plumbing only.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import capi
from .bvals import MeshBoundaryValues
from .mesh import FLT_MAX, FLT_MIN
from .tasklist import TaskID, TaskStatus


class EOS_Data:
    """src/eos/eos.hpp:27-34; defaults src/eos/eos.cpp:22-25, ideal_mhd.cpp:22"""

    def __init__(self, pin, blk):
        eos = pin.GetString(blk, "eos")
        if eos == "ideal":
            self.is_ideal = True
            self.gamma = pin.GetReal(blk, "gamma")
            self.iso_cs = 0.0
        elif eos == "isothermal":                    # isothermal_hyd.cpp:17-23, isothermal_mhd.cpp
            self.is_ideal = False
            self.iso_cs = pin.GetReal(blk, "iso_sound_speed")
            self.gamma = 0.0
        else:
            raise RuntimeError("### FATAL ERROR <%s>/eos = '%s' not implemented" % (blk, eos))
        self.dfloor = pin.GetOrAddReal(blk, "dfloor", FLT_MIN)
        self.pfloor = pin.GetOrAddReal(blk, "pfloor", FLT_MIN)
        self.tfloor = pin.GetOrAddReal(blk, "tfloor", FLT_MIN)
        self.sfloor = pin.GetOrAddReal(blk, "sfloor", FLT_MIN)
        self.sigma_max = pin.GetOrAddReal(blk, "sigma_max", FLT_MAX)


class EquationOfState:
    """holder so that `phydro.peos.eos_data` reads like the reference"""

    def __init__(self, pin, blk):
        self.eos_data = EOS_Data(pin, blk)


class FaceFld:
    """DvceFaceFld4D/5D (src/athena.hpp:178-196)"""

    def __init__(self, nmb, nvar, n3, n2, n1, device, face_shaped=True):
        f = 1 if face_shaped else 0
        sh = (nmb,) + ((nvar,) if nvar else ())
        self.x1f = torch.zeros(sh + (n3, n2, n1 + f), dtype=torch.float64, device=device)
        self.x2f = torch.zeros(sh + (n3, n2 + f, n1), dtype=torch.float64, device=device)
        self.x3f = torch.zeros(sh + (n3 + f, n2, n1), dtype=torch.float64, device=device)


class EdgeFld:
    """DvceEdgeFld4D (src/athena.hpp:223-231)"""

    def __init__(self, nmb, n3, n2, n1, device):
        self.x1e = torch.zeros((nmb, n3 + 1, n2 + 1, n1), dtype=torch.float64, device=device)
        self.x2e = torch.zeros((nmb, n3 + 1, n2, n1 + 1), dtype=torch.float64, device=device)
        self.x3e = torch.zeros((nmb, n3, n2 + 1, n1 + 1), dtype=torch.float64, device=device)


class FluidBase:
    """what Hydro and MHD share: pack descriptor, EOS, boundary values, dt bookkeeping"""

    def _setup(self, ppack, pin, blk, device):
        self.pmy_pack = ppack
        self.device = device
        self.L = capi.lib()           # raises loudly when libakmi.so is missing
        pm = ppack.pmesh
        indcs = pm.mb_indcs
        self.peos = EquationOfState(pin, blk)
        recon = pin.GetOrAddString(blk, "reconstruct", "plm")
        if recon not in capi.RECON:
            raise RuntimeError("### FATAL ERROR <%s> recon = '%s' not implemented" % (blk, recon))
        self.recon_method = capi.RECON[recon]
        if recon in ("ppm4", "ppmx", "wenoz", "teno") and indcs.ng < 3:       # hydro.cpp:173-179
            raise RuntimeError("### FATAL ERROR PPM/WENOZ reconstruction requires at least 3 "
                               "ghost zones, but <mesh>/nghost=%d" % indcs.ng)
        self.nscalars = pin.GetOrAddInteger(blk, "nscalars", 0)
        self.nmb = ppack.nmb_thispack
        self.dx_dev = torch.from_numpy(ppack.pmb.dx.copy()).to(device)
        e = self.peos.eos_data
        self.nfluid = 5 if e.is_ideal else 4         # nhydro / nmhd: no energy when isothermal
        self.nvars = self.nfluid + self.nscalars     # scalars follow the fluid variables
        self.pack_c = capi.Pack(self.nmb, self.nvars, indcs.nx1, indcs.nx2, indcs.nx3, indcs.ng,
                                self.dx_dev.data_ptr(), e.gamma, e.dfloor, e.pfloor, e.tfloor,
                                e.sfloor, e.sigma_max, e.iso_cs, 1 if e.is_ideal else 0)
        # <hydro|mhd>/fused_stage = true | false | auto (default): an explicit true / false is kept as it is
        fs = pin.GetOrAddString(blk, "fused_stage", "auto").lower()
        # (the boolean spellings of ParameterInput::GetBoolean: "true" / "false" in any case, or an integer, non-zero = true)
        if fs not in ("auto", "true", "false") and not fs.isdigit():
            raise RuntimeError("### FATAL ERROR <%s>/fused_stage = %s: true, false, an integer or auto" % (blk, fs))
        fused_given = fs != "auto"
        self.fused = (int(fs) != 0) if fs.isdigit() else fs != "false"
        # small 3-D MHD packs: the marching kernels of the fused stage are chains of dependent steps over a few hundred
        # workgroups; the task-granular chain with one thread per face is faster there (64^3: 1 113 against 1 042
        # Mcell-updates/s, 48^3: 657 / 556; equal at 72^3, 80^3: 1 392 / 1 494).  Hydro packs keep the fused stage at every
        # size: the one-kernel form wins everywhere (64^3: 2 838 against 1 854, 32^3: 485 / 303; PPM4 64^3: 1 839 / 1 583).
        # profiles/r06_small_packs.txt.  Same bits either way.
        # <hydro|mhd>/small_pack_tasks = false keeps the fused kernels (the parity tests do: their meshes are all small);
        # AKMI_SMALL_PACK_TASKS=0: off.  (Not with passive scalars: the task path's sweeps do not carry them.)
        # (read without adding it to the deck: the parameter dump of the bin/rst writers stays what the reference's is)
        # The switch applies only when the deck says <hydro|mhd>/fused_stage = auto (or nothing); the threshold is the one
        # constant AKMI_SMALL_PACK_CELLS of include/akmi.h (capi.SMALL_PACK_CELLS), shared with the C++ host and with the
        # library's choice of thread-per-face sweeps on the task path.
        small_ok = pin.GetBoolean(blk, "small_pack_tasks") if pin.DoesParameterExist(blk, "small_pack_tasks") else True
        if (blk == "mhd" and small_ok and not fused_given and self.fused and indcs.nx3 > 1 and self.nscalars == 0
                and os.environ.get("AKMI_SMALL_PACK_TASKS", "1") != "0"
                and self.nmb*indcs.nx1*indcs.nx2*indcs.nx3 <= capi.SMALL_PACK_CELLS):
            self.fused = False
        # (the fused stage kernels cover both equations of state and carry passive scalars along; FOFC,
        # the diffusion hooks and refined meshes use the task-granular kernels, see below)
        # first-order flux correction, hydro.cpp:153-190 / mhd.cpp:199-235
        self.use_fofc = pin.GetOrAddBoolean(blk, "fofc", False)
        if self.use_fofc:
            need = 3 if recon == "plm" else (4 if recon in ("ppm4", "ppmx", "wenoz", "teno") else 2)
            if indcs.ng < need:
                raise RuntimeError("### FATAL ERROR FOFC and %s reconstruction requires at least %d "
                                   "ghost zones, but <mesh>/nghost=%d" % (recon, need, indcs.ng))
            if self.nscalars > 0 or (blk == "mhd" and not e.is_ideal):
                raise RuntimeError("### FATAL ERROR <%s>/fofc with passive scalars%s is not on "
                                   "this path" % (blk, " or the isothermal EOS" if blk == "mhd" else ""))
            self.fused = False       # FOFC works on the flux arrays of the task-granular path
            n3, n2, n1 = indcs.ncells
            self.fofc = torch.zeros((self.nmb, n3, n2, n1), dtype=torch.uint8, device=device)
            self.nfofc = torch.zeros(1, dtype=torch.int32, device=device)   # EventCounters::nfofc
        # <mesh_refinement>/prolong_primitives converts with SingleC2P_IdealHyd / _IdealMHD whatever the EOS of the run
        # is (prolong_prims.cpp:35-186); this path offers it for the ideal gas only -- said here, when the physics
        # module is built, not by the first Prolongate of the run (akmi_smr_c2p_coarse would refuse there)
        if pm.multilevel and getattr(pm, "prolong_prims", False) and not e.is_ideal:
            raise RuntimeError("### FATAL ERROR <mesh_refinement>/prolong_primitives = true needs the ideal-gas EOS "
                               "(<%s>/eos = %s)" % (blk, pin.GetString(blk, "eos")))
        # viscosity / conduction / resistivity objects (hydro.cpp:77-98, mhd.cpp:104-130)
        from .diffusion import make_diffusion
        if make_diffusion(self, pin, blk):
            self.fused = False       # the flux adders work on the flux arrays of the task path
        self.counters = torch.zeros(3, dtype=torch.int32, device=device)
        self.dt3 = torch.zeros(3, dtype=torch.float64, device=device)
        self.dtnew = FLT_MAX
        self.ws = None
        self.multilevel = pm.multilevel
        if pm.multilevel:
            # the restricted fluxes of finer neighbours replace face fluxes between Fluxes and
            # RKUpdate (SendFlux/RecvFlux): the flux arrays of the task-granular path are needed.
            # (Updating in the sweeps and redoing the cells behind corrected faces was built in round 3, measured
            # slower -- profiles/r03_config5.txt -- and removed in round 4.)
            self.fused = False
            # the coarse buffers seen as a pack of nx/2 cells: HydroBCsCoarse / BFieldBCsCoarse are
            # the BC helpers on coarse indices (src/bvals/physics/hydro_bcs.cpp:51-67)
            self.cpack_c = capi.Pack(self.nmb, self.nvars, indcs.nx1//2,
                                     indcs.nx2//2 if indcs.nx2 > 1 else 1,
                                     indcs.nx3//2 if indcs.nx3 > 1 else 1, indcs.ng,
                                     self.dx_dev.data_ptr(), e.gamma, e.dfloor, e.pfloor, e.tfloor,
                                     e.sfloor, e.sigma_max, e.iso_cs, 1 if e.is_ideal else 0)

    def _coarse_shape(self):
        """(cn3, cn2, cn1) of coarse_u0 (hydro.cpp:300-310)"""
        i = self.pmy_pack.pmesh.mb_indcs
        return (i.nx3//2 + 2*i.ng if i.nx3 > 1 else 1, i.nx2//2 + 2*i.ng if i.nx2 > 1 else 1,
                i.nx1//2 + 2*i.ng)

    def _workspace(self, is_mhd):
        if self.ws is None:
            nbytes = int(self.L.akmi_stage_workspace_bytes(C.byref(self.pack_c), is_mhd))
            self.ws = torch.empty((nbytes + 7)//8, dtype=torch.float64, device=self.device)
        return self.ws

    def _diffusion_newdt(self):
        """hydro_newdt.cpp:128-133, mhd_newdt.cpp:159-167"""
        if self.pcond is not None:
            self.pcond.NewTimeStep(self.w0)
        if self.pvisc is not None:
            self.pvisc.NewTimeStep()
        if self.presist is not None:
            self.presist.NewTimeStep()

    def _finish_newdt(self):
        """host side of NewTimeStep: hydro_newdt.cpp:121-124 (blocking 24-byte D2H read)"""
        pm = self.pmy_pack.pmesh
        d = self.dt3.cpu().numpy()
        dtnew = float(d[0])
        if pm.multi_d:
            dtnew = min(dtnew, float(d[1]))
        if pm.three_d:
            dtnew = min(dtnew, float(d[2]))
        self.dtnew = dtnew


    def _oop_first(self, pdrive, stage):
        """task-granular path, first stage: CopyCons folded into an out-of-place RKUpdate / CT
        (akmi_rk_update_oop, akmi_mhd_ct_oop) and the registers swapped -- no copy traffic.  Not with FOFC (its
        trial update reads u1/b1 before RKUpdate), RK4 (CopyCons updates the second register itself), or the
        update-in-the-sweeps option."""
        return (stage == 1 and not self.fused and not self.use_fofc
                and pdrive.integrator != "rk4" and _TASK_OOP)

    @staticmethod
    def _copy_flag(pdrive, stage, phases):
        """copy_u1 of include/akmi.h: the first stage writes its result into the second register and
        the registers are swapped (no CopyCons traffic); the C2P part alone sees swapped pointers
        already; RK4's second register is updated by CopyCons itself, so it keeps the folded copy"""
        if stage != 1:
            return 0
        if pdrive.integrator == "rk4" or os.environ.get("AKMI_OUT_OF_PLACE", "1") == "0":    # (A/B switch)
            return 1
        return 2 if phases & (capi.PHASE_SWEEPS | capi.PHASE_EMF_CT) else 0


import os as _os
_MERGE_C2P = _os.environ.get("AKMI_MERGE_C2P", "1") != "0"      # A/B switch (profiles/r03_whatif_merge_c2p.txt)
_FUSE_C2P = _os.environ.get("AKMI_FUSE_C2P", "1") != "0"        # hydro: ConsToPrim inside the stage kernel (akmi_hydro_stage_w)
_TASK_OOP = _os.environ.get("AKMI_TASK_OOP", "1") != "0"        # A/B switch: first stage of the task path out of place


class Hydro(FluidBase):
    def __init__(self, ppack, pin, device=None, bvals_kernels=None, smr_kernels=None):
        device = device or capi.DEVICE
        self._setup(ppack, pin, "hydro", device)
        rs = pin.GetString("hydro", "rsolver")
        # dynamic problems: llf/hlle/hllc/roe; kinematic problems: advect (hydro.cpp:244-278)
        self.kinematic = pin.GetOrAddString("time", "evolution", "dynamic") == "kinematic"
        if self.kinematic:
            if rs != "advect":
                raise RuntimeError("### FATAL ERROR <hydro> rsolver = '%s' not implemented for "
                                   "kinematic problems" % rs)
            self.fused = False
        elif rs not in ("llf", "hlle", "hllc", "roe"):          # hydro.cpp: Hydro_RSolver
            raise RuntimeError("### FATAL ERROR <hydro> rsolver = '%s' not implemented for dynamic "
                               "problems (llf, hlle, hllc, roe on this path)" % rs)
        self.rsolver_method = capi.RSOLVER[rs]
        self.nhydro = self.nfluid
        n3, n2, n1 = ppack.pmesh.mb_indcs.ncells
        sh = (self.nmb, self.nvars, n3, n2, n1)
        z = lambda: torch.zeros(sh, dtype=torch.float64, device=device)
        self.u0, self.w0, self.u1 = z(), z(), z()
        # task-granular path keeps the reference's cell-shaped flux arrays (hydro.cpp:290-292)
        self.uflx = None if self.fused else FaceFld(self.nmb, self.nvars, n3, n2, n1, device, face_shaped=False)
        self.pbval_u = MeshBoundaryValues(ppack, bvals_kernels, device)
        self.pbval_u.set_pack(self.pack_c, self.nvars)
        self.psmr = None
        if self.multilevel:
            from .bvals_smr import MeshBoundaryValuesSMR
            c3, c2, c1 = self._coarse_shape()
            self.coarse_u0 = torch.zeros((self.nmb, self.nvars, c3, c2, c1), dtype=torch.float64,
                                         device=device)
            self.psmr = MeshBoundaryValuesSMR(ppack, self.nvars, smr_kernels, device)
            self.psmr.set_pack(self.pack_c)

    # ---- task list assembly: hydro_tasks.cpp:48-80 ---------------------------------
    def AssembleHydroTasks(self, tl):
        none = TaskID(0)
        self.id = {}
        i = self.id
        i["irecv"] = tl["before_stagen"].AddTask(self.InitRecv, none)
        s = tl["stagen"]
        i["copyu"] = s.AddTask(self.CopyCons, none)
        i["flux"] = s.AddTask(self.Fluxes, i["copyu"])
        i["sendf"] = s.AddTask(self.SendFlux, i["flux"])
        i["recvf"] = s.AddTask(self.RecvFlux, i["sendf"])
        i["rkupdt"] = s.AddTask(self.RKUpdate, i["recvf"])
        i["srctrms"] = s.AddTask(self.HydroSrcTerms, i["rkupdt"])
        i["sendu_oa"] = s.AddTask(self.SendU_OA, i["srctrms"])
        i["recvu_oa"] = s.AddTask(self.RecvU_OA, i["sendu_oa"])
        i["restu"] = s.AddTask(self.RestrictU, i["recvu_oa"])
        i["sendu"] = s.AddTask(self.SendU, i["restu"])
        i["recvu"] = s.AddTask(self.RecvU, i["sendu"])
        i["sendu_shr"] = s.AddTask(self.SendU_Shr, i["recvu"])
        i["recvu_shr"] = s.AddTask(self.RecvU_Shr, i["sendu_shr"])
        i["prol"] = s.AddTask(self.Prolongate, i["recvu_shr"])
        i["bcs"] = s.AddTask(self.ApplyPhysicalBCs, i["prol"])
        i["c2p"] = s.AddTask(self.ConToPrim, i["bcs"])
        i["newdt"] = s.AddTask(self.NewTimeStep, i["c2p"])
        i["csend"] = tl["after_stagen"].AddTask(self.ClearSend, none)
        i["crecv"] = tl["after_stagen"].AddTask(self.ClearRecv, i["csend"])

    # ---- tasks ---------------------------------------------------------------------
    def _noop(self, pdrive, stage):
        return TaskStatus.complete

    InitRecv = HydroSrcTerms = SendU_OA = RecvU_OA = _noop
    SendU_Shr = RecvU_Shr = ClearSend = ClearRecv = _noop

    def SendFlux(self, pdrive, stage):
        """hydro_tasks.cpp:206-215: restricted fluxes at fine/coarse boundaries (SMR only)"""
        if self.multilevel:
            return self.psmr.PackAndSendFluxCC(self.uflx, False)
        return TaskStatus.complete

    def RecvFlux(self, pdrive, stage):
        """hydro_tasks.cpp:222-232"""
        if self.multilevel:
            return self.psmr.RecvAndUnpackFluxCC(self.uflx, False)
        return TaskStatus.complete

    def RestrictU(self, pdrive, stage):
        """hydro_tasks.cpp:291-300"""
        if self.multilevel:
            return self.psmr.RestrictCC(self.u0, self.coarse_u0)
        return TaskStatus.complete

    def Prolongate(self, pdrive, stage):
        """hydro_tasks.cpp:381-400"""
        if self.multilevel:
            self.psmr.FillCoarseInBndryCC(self.u0, self.coarse_u0)
            if not self.pmy_pack.pmesh.strictly_periodic:
                self.pbval_u.k.hydro_bcs(self.cpack_c, self.nvars, self.pbval_u.bcs, self.coarse_u0,
                                         self.pbval_u.u_in)
            if self.pmy_pack.pmesh.prolong_prims:          # hydro_tasks.cpp:388-392
                if getattr(self, "coarse_w0", None) is None:
                    import torch
                    self.coarse_w0 = torch.zeros_like(self.coarse_u0)
                self.psmr.ConsToPrimCoarseBndry(self.coarse_u0, None, self.coarse_w0)
                self.psmr.ProlongateCC(self.w0, self.coarse_w0)
                self.psmr.PrimToConsFineBndry(self.w0, None, self.u0)
            else:
                self.psmr.ProlongateCC(self.u0, self.coarse_u0)
        return TaskStatus.complete

    def CopyCons(self, pdrive, stage):
        """hydro_tasks.cpp:130-152 (folded into the fused stage kernel when fused)"""
        if self._oop_first(pdrive, stage):
            pass                                  # RKUpdate writes the new state into u1 and swaps the registers
        elif stage == 1 and not self.fused:
            capi.check(self.L.akmi_copy_cons(C.byref(self.pack_c), capi._p(self.u0),
                                             capi._p(self.u1), capi._stream()), "copy_cons")
        elif stage > 1 and pdrive.integrator == "rk4":
            capi.check(self.L.akmi_rk4_copy_cons(
                C.byref(self.pack_c), C.c_double(pdrive.delta[stage - 1]), capi._p(self.u0),
                capi._p(self.u1), capi._stream()), "rk4_copy_cons")
        return TaskStatus.complete

    def Fluxes(self, pdrive, stage):
        """hydro_tasks.cpp:159-201"""
        if self.fused:
            return TaskStatus.complete
        # hydro_fluxes.cpp:92-101: ranges extended by one cell when FOFC is on
        fn = self.L.akmi_hydro_fluxes_fofc if self.use_fofc else self.L.akmi_hydro_fluxes
        capi.check(fn(C.byref(self.pack_c), self.recon_method, self.rsolver_method, capi._p(self.w0),
                      capi._p(self.uflx.x1f), capi._p(self.uflx.x2f), capi._p(self.uflx.x3f), 0,
                      capi._stream()), "hydro_fluxes")
        if self.pcond is not None:                       # hydro_tasks.cpp:184-189
            self.pcond.AddHeatFluxes(self.w0, self.uflx, 0)
        if self.pvisc is not None:
            self.pvisc.AddViscousFluxes(self.w0, self.uflx, 0)
        if self.use_fofc:                                # hydro_tasks.cpp:192-194 -> Hydro::FOFC
            capi.check(self.L.akmi_hydro_fofc(
                C.byref(self.pack_c), C.c_double(pdrive.gam0[stage - 1]),
                C.c_double(pdrive.gam1[stage - 1]),
                C.c_double(pdrive.beta[stage - 1]*self.pmy_pack.pmesh.dt), capi._p(self.w0),
                capi._p(self.u0), capi._p(self.u1), capi._p(self.uflx.x1f), capi._p(self.uflx.x2f),
                capi._p(self.uflx.x3f), 0, capi._p(self.fofc), capi._p(self.nfofc),
                capi._stream()), "hydro_fofc")
        return TaskStatus.complete

    def RKUpdate(self, pdrive, stage):
        """hydro_update.cpp:23-83"""
        gam0, gam1 = pdrive.gam0[stage - 1], pdrive.gam1[stage - 1]
        beta_dt = pdrive.beta[stage - 1]*self.pmy_pack.pmesh.dt
        if self.fused and self.pbval_u.peers:
            # off-rank neighbours: only the sweeps + update here, so that SendU can post the
            # halo messages before the c2p of the active cells is enqueued (see SendU)
            self._stage_phase(pdrive, stage, capi.PHASE_SWEEPS)
        elif self.fused and _FUSE_C2P and self._w_eligible():
            # the stage kernel converts the cells it finishes (their new state is in its registers) into the second
            # primitive array; ConToPrim then only has the ghost shell left (after the ghost fill)
            self._stage_w(pdrive, stage)
        elif self.fused and _MERGE_C2P:
            # no off-rank neighbour: ONE ConsToPrim over all cells after the ghost fill (ConToPrim below) instead of
            # c2p(active cells) here + c2p(ghost shell) there
            self._stage_phase(pdrive, stage, capi.PHASE_SWEEPS)
        elif self.fused:
            # pass A + ConsToPrim of the active cells (+ CFL scan on the last stage) in one
            # call; the ghost shell is converted in ConToPrim after the halo
            self._stage_phase(pdrive, stage, capi.PHASE_ALL)
        elif self._oop_first(pdrive, stage):
            capi.check(self.L.akmi_rk_update_oop(
                C.byref(self.pack_c), capi.d(gam0), capi.d(gam1), capi.d(beta_dt), capi._p(self.u0),
                capi._p(self.u1), capi._p(self.uflx.x1f), capi._p(self.uflx.x2f),
                capi._p(self.uflx.x3f), 0, capi._stream()), "rk_update_oop")
            self.u0, self.u1 = self.u1, self.u0
        else:
            capi.check(self.L.akmi_rk_update(
                C.byref(self.pack_c), capi.d(gam0), capi.d(gam1), capi.d(beta_dt), capi._p(self.u0),
                capi._p(self.u1), capi._p(self.uflx.x1f), capi._p(self.uflx.x2f),
                capi._p(self.uflx.x3f), 0, capi._stream()), "rk_update")
        return TaskStatus.complete

    def _w_eligible(self):
        """akmi_hydro_stage_w + akmi_hydro_ghost_uw: no off-rank neighbour, every boundary's value rule commutes with
        ConsToPrim (neighbour / periodic copies, outflow, reflect), no user boundary function"""
        fn = getattr(self.L, "akmi_hydro_stage_w_eligible", None) if not hasattr(self.L, "R") else None   # (the CPU stand-in
        if not fn or self.pbval_u.peers:                                                                    # of the tests has none)
            return False
        pgen = self.pmy_pack.pmesh.pgen
        if pgen is not None and pgen.user_bcs:
            return False
        ok = tuple(capi.BC[k] for k in ("block", "periodic", "outflow", "reflect"))
        if any(int(f) not in ok for f in np.asarray(self.pmy_pack.pmb.mb_bcs).ravel()):
            return False
        return bool(fn(C.byref(self.pack_c), self.recon_method, self.rsolver_method))

    def _stage_w(self, pdrive, stage):
        """akmi_hydro_stage_w: the whole stage, ConsToPrim of the active cells inside the update kernel, the new
        primitives in the second primitive array (the two trade places afterwards)"""
        import torch
        gam0, gam1 = pdrive.gam0[stage - 1], pdrive.gam1[stage - 1]
        beta_dt = pdrive.beta[stage - 1]*self.pmy_pack.pmesh.dt
        do_dt = 1 if stage == pdrive.nexp_stages else 0
        copy = self._copy_flag(pdrive, stage, capi.PHASE_ALL)
        if getattr(self, "w1", None) is None:
            self.w1 = torch.zeros_like(self.w0)
        ev = getattr(self, "stage_events", None)     # bench.py: HIP event pair around the launch group
        if ev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev.append((e0, e1))
            e0.record()
        wrote = C.c_int(0)
        capi.check(self.L.akmi_hydro_stage_w(
            C.byref(self.pack_c), self.recon_method, self.rsolver_method, capi.d(gam0), capi.d(gam1), capi.d(beta_dt),
            None, copy, capi._p(self.w0), capi._p(self.w1), capi._p(self.u0), capi._p(self.u1), do_dt,
            capi._p(self.counters), capi._p(self.dt3), capi._p(self._workspace(0)), capi._stream(), C.byref(wrote)),
            "hydro_stage_w")
        if ev is not None:
            e1.record()
        if copy == 2:                      # out-of-place first stage: the registers trade places
            self.u0, self.u1 = self.u1, self.u0
        if wrote.value:
            self.w0, self.w1 = self.w1, self.w0
        self._interior_done = True
        self._dt_ready = bool(do_dt)
        self._want_ghost_c2p = True

    def _stage_phase(self, pdrive, stage, phases):
        """akmi_hydro_stage_phase: the parts of the fused stage named by the mask `phases`"""
        # stage 0 = Driver::InitBoundaryValuesAndPrimitives: only the c2p part may run then (the RK
        # weights of "stage 0" do not exist; [stage - 1] would silently pick the last stage's)
        assert stage >= 1 or phases == capi.PHASE_C2P, (stage, phases)
        if stage >= 1:
            gam0, gam1 = pdrive.gam0[stage - 1], pdrive.gam1[stage - 1]
            beta_dt = pdrive.beta[stage - 1]*self.pmy_pack.pmesh.dt
        else:
            gam0, gam1, beta_dt = 1.0, 0.0, 0.0
        do_dt = 1 if stage == pdrive.nexp_stages else 0
        copy = self._copy_flag(pdrive, stage, phases)
        ev = getattr(self, "stage_events", None)     # bench.py: HIP event pair around the launch group
        if ev is not None:
            import torch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev.append((e0, e1))
            e0.record()
        capi.check(self.L.akmi_hydro_stage_phase(
            C.byref(self.pack_c), self.recon_method, self.rsolver_method, capi.d(gam0),
            capi.d(gam1), capi.d(beta_dt), copy, capi._p(self.w0),
            capi._p(self.u0), capi._p(self.u1), do_dt, capi._p(self.counters),
            capi._p(self.dt3), phases, capi._p(self._workspace(0)), capi._stream()),
            "hydro_stage_phase")
        if ev is not None:
            e1.record()
        if copy == 2:                      # out-of-place first stage: the registers trade places
            self.u0, self.u1 = self.u1, self.u0
        if phases & capi.PHASE_C2P:
            self._interior_done = True
            self._dt_ready = bool(do_dt)

    def SendU(self, pdrive, stage):
        if self.multilevel:
            return self.psmr.PackAndSendCC(self.u0, self.coarse_u0)
        if getattr(self, "_want_ghost_c2p", False):
            # the stage kernel has converted the active cells (akmi_hydro_stage_w): ghost zones of u0 AND of the new primitive
            # array by the same gather + boundary functions, one launch, nothing converted twice (akmi_hydro_ghost_uw)
            self._want_ghost_c2p = False
            bv, pm = self.pbval_u, self.pmy_pack.pmesh
            hb = np.ascontiguousarray(self.pmy_pack.pmb.mb_bcs, dtype=np.int32)
            capi.check(self.L.akmi_hydro_ghost_uw(
                C.byref(self.pack_c), capi._p(bv.nghbr), capi._p(bv.bcs), hb.ctypes.data_as(C.c_void_p), capi._p(self.u0),
                capi._p(self.w0), capi._p(self._workspace(0)), capi._p(self.counters), capi._stream()), "hydro_ghost_uw")
            bv._u_bcs_done = True
            self._shell_done = True
            self._dt3_reset = False
            return TaskStatus.complete
        reset = None
        if (self.fused and self.pbval_u.fold_bcs and stage >= 1 and stage == pdrive.nexp_stages
                and not getattr(self, "_interior_done", False)):
            reset = self.dt3                 # the gather of the last stage also resets the CFL minima
        st = self.pbval_u.PackAndSendCC(self.u0, reset)
        self._dt3_reset = reset is not None
        if self.fused and self.pbval_u.peers:
            # the messages are in flight on the transport's stream: convert the active cells
            # (they do not depend on the halo) underneath them
            self._stage_phase(pdrive, stage, capi.PHASE_C2P)
        return st

    def RecvU(self, pdrive, stage):
        if self.multilevel:
            return self.psmr.RecvAndUnpackCC(self.u0, self.coarse_u0)
        return self.pbval_u.RecvAndUnpackCC(self.u0)

    def ApplyPhysicalBCs(self, pdrive, stage):
        """hydro_tasks.cpp:357-375"""
        if self.pmy_pack.pmesh.strictly_periodic:
            return TaskStatus.complete
        self.pbval_u.HydroBCs(self.u0)
        pgen = self.pmy_pack.pmesh.pgen
        if pgen is not None and pgen.user_bcs:                   # hydro_tasks.cpp:368-371
            pgen.user_bcs_func()
        return TaskStatus.complete

    def ConToPrim(self, pdrive, stage):
        """hydro_tasks.cpp:404-412: all cells including ghosts (the fused path also performs
        the CFL scan of NewTimeStep in the same kernel on the last stage)"""
        n3, n2, n1 = self.pmy_pack.pmesh.mb_indcs.ncells
        if self.fused and getattr(self, "_interior_done", False):
            self._interior_done = False
            if not getattr(self, "_shell_done", False):
                capi.check(self.L.akmi_hydro_c2p_shell(
                    C.byref(self.pack_c), capi._p(self.u0), capi._p(self.w0), capi._p(self.counters),
                    capi._stream()), "hydro_c2p_shell")
            self._shell_done = False
            return TaskStatus.complete
        if self.fused:
            do_dt = 1 if stage == pdrive.nexp_stages else 0
            ev = getattr(self, "stage_events", None)     # bench.py: the conversion belongs to the stage's launch group
            if ev is not None:
                import torch
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev.append((e0, e1))
                e0.record()
            capi.check(self.L.akmi_hydro_c2p_newdt(
                C.byref(self.pack_c), capi._p(self.u0), capi._p(self.w0),
                2 if (do_dt and getattr(self, "_dt3_reset", False)) else do_dt,
                capi._p(self.counters), capi._p(self.dt3), capi._stream()), "hydro_c2p_newdt")
            self._dt3_reset = False
            if ev is not None:
                e1.record()
            self._dt_ready = bool(do_dt)
            return TaskStatus.complete
        if self.multilevel and not self.kinematic:
            # refined meshes (task-granular chain): the last conversion carries the CFL scan along, the others use the same
            # entry without it (see mhd.py)
            do_dt = 1 if stage == pdrive.nexp_stages else 0
            capi.check(self.L.akmi_hydro_c2p_newdt(
                C.byref(self.pack_c), capi._p(self.u0), capi._p(self.w0), do_dt,
                capi._p(self.counters), capi._p(self.dt3), capi._stream()), "hydro_c2p_newdt")
            self._dt_ready = bool(do_dt)
            return TaskStatus.complete
        capi.check(self.L.akmi_hydro_c2p(C.byref(self.pack_c), capi._p(self.u0), capi._p(self.w0),
                                         0, n1 - 1, 0, n2 - 1, 0, n3 - 1, capi._p(self.counters),
                                         capi._stream()), "hydro_c2p")
        return TaskStatus.complete

    def NewTimeStep(self, pdrive, stage):
        """hydro_newdt.cpp:30-139: last stage only"""
        if stage != pdrive.nexp_stages:
            return TaskStatus.complete
        if self.kinematic:                                       # hydro_newdt.cpp:55-72
            capi.check(self.L.akmi_kinematic_newdt(C.byref(self.pack_c), capi._p(self.w0),
                                                   capi._p(self.dt3), capi._stream()), "kinematic_newdt")
        elif not getattr(self, "_dt_ready", False):
            capi.check(self.L.akmi_hydro_newdt(C.byref(self.pack_c), capi._p(self.w0),
                                               capi._p(self.dt3), capi._stream()), "hydro_newdt")
        self._dt_ready = False
        self._finish_newdt()
        self._diffusion_newdt()
        return TaskStatus.complete
