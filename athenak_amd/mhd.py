"""mhd::MHD -- arrays and task member functions of the MHD module.

Mirror of src/mhd/mhd.hpp:93-199 / mhd.cpp:30-382 / mhd_tasks.cpp:38-699: members u0, w0,
u1, b0, b1, bcc0, uflx, efld, e3x1..e1x3, peos, pbval_u, pbval_b, dtnew; the `stagen` chain
of AssembleMHDTasks (mhd_tasks.cpp:48-75) with each body one call through include/akmi.h.
"""
import ctypes as C

import torch

from . import capi
from .bvals import MeshBoundaryValues
from .hydro import EdgeFld, FaceFld, FluidBase
from .tasklist import TaskID, TaskStatus


import os as _os
_MERGE_C2P = _os.environ.get("AKMI_MERGE_C2P", "1") != "0"      # A/B switch (profiles/r03_whatif_merge_c2p.txt)


class MHD(FluidBase):
    def __init__(self, ppack, pin, device=None, bvals_kernels=None, smr_kernels=None):
        device = device or capi.DEVICE
        self._setup(ppack, pin, "mhd", device)
        rs = pin.GetString("mhd", "rsolver")
        # dynamic problems: llf/hlle/hlld; kinematic problems: advect (mhd.cpp:292-326)
        self.kinematic = pin.GetOrAddString("time", "evolution", "dynamic") == "kinematic"
        if self.kinematic:
            if rs != "advect":
                raise RuntimeError("### FATAL ERROR <mhd> rsolver = '%s' not implemented for "
                                   "kinematic problems" % rs)
            self.fused = False
        elif rs not in ("llf", "hlle", "hlld"):                 # mhd.cpp: MHD_RSolver
            raise RuntimeError("### FATAL ERROR <mhd> rsolver = '%s' not implemented for dynamic "
                               "problems (llf, hlle, hlld on this path)" % rs)
        self.rsolver_method = capi.RSOLVER[rs]
        self.nmhd = self.nfluid
        n3, n2, n1 = ppack.pmesh.mb_indcs.ncells
        nmb = self.nmb
        z5 = lambda: torch.zeros((nmb, self.nvars, n3, n2, n1), dtype=torch.float64, device=device)
        self.u0, self.w0, self.u1 = z5(), z5(), z5()
        self.bcc0 = torch.zeros((nmb, 3, n3, n2, n1), dtype=torch.float64, device=device)
        self.b0 = FaceFld(nmb, 0, n3, n2, n1, device)
        self.b1 = FaceFld(nmb, 0, n3, n2, n1, device)
        if not self.fused:
            zc = lambda: torch.zeros((nmb, n3, n2, n1), dtype=torch.float64, device=device)
            self.uflx = FaceFld(nmb, self.nvars, n3, n2, n1, device)        # mhd.cpp:341-343
            self.efld = EdgeFld(nmb, n3, n2, n1, device)           # mhd.cpp:344-346
            self.e3x1, self.e2x1, self.e1x2 = zc(), zc(), zc()     # mhd.cpp:349-354
            self.e3x2, self.e2x3, self.e1x3 = zc(), zc(), zc()
        self.pbval_u = MeshBoundaryValues(ppack, bvals_kernels, device)
        self.pbval_u.set_pack(self.pack_c, self.nvars)
        self.pbval_b = self.pbval_u       # same neighbour tables; separate FC channel inside
        self.psmr = None
        if self.multilevel:
            # coarse buffers (mhd.cpp:368-380) and the level-aware boundary values
            from .bvals_smr import MeshBoundaryValuesSMR
            c3, c2, c1 = self._coarse_shape()
            self.coarse_u0 = torch.zeros((nmb, self.nvars, c3, c2, c1), dtype=torch.float64, device=device)
            self.coarse_b0 = FaceFld(nmb, 0, c3, c2, c1, device)
            self.psmr = MeshBoundaryValuesSMR(ppack, self.nvars, smr_kernels, device)
            self.psmr.set_pack(self.pack_c)

    # ---- task list assembly: mhd_tasks.cpp:38-84 -----------------------------------
    def AssembleMHDTasks(self, tl):
        none = TaskID(0)
        self.id = {}
        i = self.id
        i["savest"] = tl["before_timeintegrator"].AddTask(self.SaveMHDState, none)
        i["irecv"] = tl["before_stagen"].AddTask(self.InitRecv, none)
        s = tl["stagen"]
        chain = [("copyu", self.CopyCons), ("flux", self.Fluxes), ("sendf", self.SendFlux),
                 ("recvf", self.RecvFlux), ("rkupdt", self.RKUpdate), ("srctrms", self.MHDSrcTerms),
                 ("sendu_oa", self.SendU_OA), ("recvu_oa", self.RecvU_OA), ("restu", self.RestrictU),
                 ("sendu", self.SendU), ("recvu", self.RecvU), ("sendu_shr", self.SendU_Shr),
                 ("recvu_shr", self.RecvU_Shr), ("efld", self.EField), ("sende", self.SendE),
                 ("recve", self.RecvE), ("ct", self.CT), ("sendb_oa", self.SendB_OA),
                 ("recvb_oa", self.RecvB_OA), ("restb", self.RestrictB), ("sendb", self.SendB),
                 ("recvb", self.RecvB), ("sendb_shr", self.SendB_Shr), ("recvb_shr", self.RecvB_Shr),
                 ("prol", self.Prolongate), ("bcs", self.ApplyPhysicalBCs), ("c2p", self.ConToPrim),
                 ("newdt", self.NewTimeStep)]
        dep = none
        for name, fn in chain:
            i[name] = s.AddTask(fn, dep)
            dep = i[name]
        i["csend"] = tl["after_stagen"].AddTask(self.ClearSend, none)
        i["crecv"] = tl["after_stagen"].AddTask(self.ClearRecv, i["csend"])

    # ---- tasks ---------------------------------------------------------------------
    def _noop(self, pdrive, stage):
        return TaskStatus.complete

    SaveMHDState = InitRecv = MHDSrcTerms = SendU_OA = RecvU_OA = _noop
    SendU_Shr = RecvU_Shr = SendB_OA = RecvB_OA = _noop
    SendB_Shr = RecvB_Shr = ClearSend = ClearRecv = _noop

    def SendE(self, pdrive, stage):
        """mhd_tasks.cpp:402-417 (PackAndSendFluxFC + RecvAndUnpackFluxFC).  On a uniform mesh every
        copy of a shared edge EMF is computed by the same deterministic kernel from identical inputs:
        the reference's sum and average, (a+a)*0.5 on faces and (((a+a)+a)+a)/4 on edges, return a
        bit for bit (2a and 4a are exact, fl(3a)+a rounds to 4a), so nothing is exchanged there.
        With levels this is the flux correction of the field: akmi_smr_emf_exchange."""
        if self.multilevel:
            return self.psmr.PackAndSendFluxFC(self.efld)
        return TaskStatus.complete

    def SendFlux(self, pdrive, stage):
        """mhd_tasks.cpp:225-233"""
        if self.multilevel:
            return self.psmr.PackAndSendFluxCC(self.uflx, True)
        return TaskStatus.complete

    def RecvFlux(self, pdrive, stage):
        """mhd_tasks.cpp:240-250"""
        if self.multilevel:
            return self.psmr.RecvAndUnpackFluxCC(self.uflx, True)
        return TaskStatus.complete

    def RecvE(self, pdrive, stage):
        """mhd_tasks.cpp:410-417: sum over same-level owners, zero at finer neighbours, sum their
        restricted values, average"""
        if self.multilevel:
            return self.psmr.RecvAndUnpackFluxFC(self.efld)
        return TaskStatus.complete

    def RestrictU(self, pdrive, stage):
        """mhd_tasks.cpp:315-322"""
        if self.multilevel:
            return self.psmr.RestrictCC(self.u0, self.coarse_u0)
        return TaskStatus.complete

    def RestrictB(self, pdrive, stage):
        """mhd_tasks.cpp:691-697"""
        if self.multilevel:
            return self.psmr.RestrictFC(self.b0, self.coarse_b0)
        return TaskStatus.complete

    def Prolongate(self, pdrive, stage):
        """mhd_tasks.cpp:527-552"""
        if self.multilevel:
            ps, pb = self.psmr, self.pbval_u
            ps.FillCoarseInBndryCC(self.u0, self.coarse_u0)
            ps.FillCoarseInBndryFC(self.b0, self.coarse_b0)
            if not self.pmy_pack.pmesh.strictly_periodic:
                pb.k.hydro_bcs(self.cpack_c, self.nvars, pb.bcs, self.coarse_u0, pb.u_in)
                pb.k.bfield_bcs(self.cpack_c, pb.bcs, self.coarse_b0.x1f, self.coarse_b0.x2f,
                                self.coarse_b0.x3f, pb.b_in)
            if self.pmy_pack.pmesh.prolong_prims:          # mhd_tasks.cpp:539-544
                if getattr(self, "coarse_w0", None) is None:
                    import torch
                    self.coarse_w0 = torch.zeros_like(self.coarse_u0)
                ps.ConsToPrimCoarseBndry(self.coarse_u0, self.coarse_b0, self.coarse_w0)
                ps.ProlongateCC(self.w0, self.coarse_w0)
                ps.ProlongateFC(self.b0, self.coarse_b0)
                ps.PrimToConsFineBndry(self.w0, self.b0, self.u0)
            else:
                ps.ProlongateCC(self.u0, self.coarse_u0)
                ps.ProlongateFC(self.b0, self.coarse_b0)
        return TaskStatus.complete

    def _b(self, f):
        return capi._p(f.x1f), capi._p(f.x2f), capi._p(f.x3f)

    def CopyCons(self, pdrive, stage):
        """mhd_tasks.cpp:162-170"""
        if self._oop_first(pdrive, stage):
            return TaskStatus.complete            # RKUpdate / CT write u1 / b1 and swap the registers
        if stage == 1 and not self.fused:
            capi.check(self.L.akmi_copy_cons(C.byref(self.pack_c), capi._p(self.u0),
                                             capi._p(self.u1), capi._stream()), "copy_cons")
            self.b1.x1f.copy_(self.b0.x1f)
            self.b1.x2f.copy_(self.b0.x2f)
            self.b1.x3f.copy_(self.b0.x3f)
        return TaskStatus.complete

    def Fluxes(self, pdrive, stage):
        """mhd_tasks.cpp:177-216"""
        if not self.fused:
            efc = [capi._p(x) for x in (self.e3x1, self.e2x1, self.e1x2, self.e3x2, self.e2x3, self.e1x3)]
            fn = self.L.akmi_mhd_fluxes_fofc if self.use_fofc else self.L.akmi_mhd_fluxes
            capi.check(fn(C.byref(self.pack_c), self.recon_method, self.rsolver_method,
                          capi._p(self.w0), capi._p(self.bcc0), *self._b(self.b0), *self._b(self.uflx),
                          *efc, capi._stream()), "mhd_fluxes")
            if self.pcond is not None:                   # mhd_tasks.cpp:198-206
                self.pcond.AddHeatFluxes(self.w0, self.uflx, 1)
            if self.pvisc is not None:
                self.pvisc.AddViscousFluxes(self.w0, self.uflx, 1)
            if self.presist is not None and self.peos.eos_data.is_ideal:
                self.presist.AddResistiveFluxes(self.b0, self.uflx)
            if self.use_fofc:                    # mhd_tasks.cpp:209-211 -> MHD::FOFC
                capi.check(self.L.akmi_mhd_fofc(
                    C.byref(self.pack_c), C.c_double(pdrive.gam0[stage - 1]),
                    C.c_double(pdrive.gam1[stage - 1]),
                    C.c_double(pdrive.beta[stage - 1]*self.pmy_pack.pmesh.dt), capi._p(self.w0),
                    capi._p(self.bcc0), *self._b(self.b0), *self._b(self.b1), capi._p(self.u0),
                    capi._p(self.u1), *self._b(self.uflx), *efc, capi._p(self.fofc),
                    capi._p(self.nfofc), capi._stream()), "mhd_fofc")
        return TaskStatus.complete

    def RKUpdate(self, pdrive, stage):
        """mhd_update.cpp:24-84; the fused path also performs EField and CT here"""
        gam0, gam1 = pdrive.gam0[stage - 1], pdrive.gam1[stage - 1]
        beta_dt = pdrive.beta[stage - 1]*self.pmy_pack.pmesh.dt
        if self.fused and self.pbval_u.peers:
            # off-rank neighbours: the stage is issued in parts so that the halo messages are
            # posted as early as the reference's task list allows and travel underneath the
            # remaining kernels:  sweeps+update | SendU | CornerE+CT | SendB | c2p of the
            # active cells | RecvU, RecvB | BCs | c2p of the ghost shell
            self._stage_phase(pdrive, stage, capi.PHASE_SWEEPS)
        elif self.fused and _MERGE_C2P:
            # no off-rank neighbour: nothing travels underneath an early conversion of the active cells, so the
            # stage ends with ONE ConsToPrim over all cells incl. the ghost zones, after the ghost fill
            # (ConToPrim below), instead of c2p(active) here + c2p(ghost shell, thin slabs) there
            self._stage_phase(pdrive, stage, capi.PHASE_SWEEPS | capi.PHASE_EMF_CT)
        elif self.fused:
            # pass A (fluxes, update, CornerE, CT) + ConsToPrim of the active cells (+ CFL scan
            # on the last stage) in one call
            self._stage_phase(pdrive, stage, capi.PHASE_ALL)
        elif self._oop_first(pdrive, stage):
            capi.check(self.L.akmi_rk_update_oop(
                C.byref(self.pack_c), capi.d(gam0), capi.d(gam1), capi.d(beta_dt), capi._p(self.u0),
                capi._p(self.u1), *self._b(self.uflx), 1, capi._stream()), "rk_update_oop")
            self.u0, self.u1 = self.u1, self.u0
        else:
            capi.check(self.L.akmi_rk_update(
                C.byref(self.pack_c), capi.d(gam0), capi.d(gam1), capi.d(beta_dt), capi._p(self.u0),
                capi._p(self.u1), *self._b(self.uflx), 1, capi._stream()), "rk_update")
        return TaskStatus.complete

    def _stage_phase(self, pdrive, stage, phases):
        """akmi_mhd_stage_phase: the parts of the fused stage named by the mask `phases`"""
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
        capi.check(self.L.akmi_mhd_stage_phase(
            C.byref(self.pack_c), self.recon_method, self.rsolver_method, capi.d(gam0),
            capi.d(gam1), capi.d(beta_dt), copy, capi._p(self.w0),
            capi._p(self.bcc0), capi._p(self.u0), capi._p(self.u1), *self._b(self.b0),
            *self._b(self.b1), do_dt, capi._p(self.counters), capi._p(self.dt3), phases,
            capi._p(self._workspace(1)), capi._stream()), "mhd_stage_phase")
        if ev is not None:
            e1.record()
        if copy == 2:                      # out-of-place first stage: the registers trade places
            if phases & capi.PHASE_SWEEPS:
                self.u0, self.u1 = self.u1, self.u0
            if phases & capi.PHASE_EMF_CT:
                self.b0, self.b1 = self.b1, self.b0
        if phases & capi.PHASE_C2P:
            self._interior_done = True
            self._dt_ready = bool(do_dt)

    def SendU(self, pdrive, stage):
        if self.multilevel:
            return self.psmr.PackAndSendCC(self.u0, self.coarse_u0)
        return self.pbval_u.PackAndSendCC(self.u0)

    def RecvU(self, pdrive, stage):
        if self.multilevel:
            return self.psmr.RecvAndUnpackCC(self.u0, self.coarse_u0)
        if self.fused and self.pbval_u.peers:
            return TaskStatus.complete      # completed in RecvB, after the kernels it can hide under
        return self.pbval_u.RecvAndUnpackCC(self.u0)

    def EField(self, pdrive, stage):
        """MHD::CornerE, mhd_corner_e.cpp:26-417"""
        if not self.fused:
            capi.check(self.L.akmi_mhd_corner_e(
                C.byref(self.pack_c), capi._p(self.w0), capi._p(self.bcc0), capi._p(self.e3x1),
                capi._p(self.e2x1), capi._p(self.e1x2), capi._p(self.e3x2), capi._p(self.e2x3),
                capi._p(self.e1x3), *self._b(self.uflx), capi._p(self.efld.x1e),
                capi._p(self.efld.x2e), capi._p(self.efld.x3e), capi._stream()), "mhd_corner_e")
            if self.presist is not None:                 # mhd_tasks.cpp:381-383
                self.presist.AddResistiveEMFs(self.b0, self.efld)
        return TaskStatus.complete

    def CT(self, pdrive, stage):
        """mhd_ct.cpp:23-80"""
        if self.fused and self.pbval_u.peers:
            self._stage_phase(pdrive, stage, capi.PHASE_EMF_CT)
        elif not self.fused:
            gam0, gam1 = pdrive.gam0[stage - 1], pdrive.gam1[stage - 1]
            beta_dt = pdrive.beta[stage - 1]*self.pmy_pack.pmesh.dt
            if self._oop_first(pdrive, stage):
                capi.check(self.L.akmi_mhd_ct_oop(
                    C.byref(self.pack_c), capi.d(gam0), capi.d(gam1), capi.d(beta_dt),
                    capi._p(self.efld.x1e), capi._p(self.efld.x2e), capi._p(self.efld.x3e),
                    *self._b(self.b0), *self._b(self.b1), capi._stream()), "mhd_ct_oop")
                self.b0, self.b1 = self.b1, self.b0
                return TaskStatus.complete
            capi.check(self.L.akmi_mhd_ct(
                C.byref(self.pack_c), capi.d(gam0), capi.d(gam1), capi.d(beta_dt),
                capi._p(self.efld.x1e), capi._p(self.efld.x2e), capi._p(self.efld.x3e),
                *self._b(self.b0), *self._b(self.b1), capi._stream()), "mhd_ct")
        return TaskStatus.complete

    def SendB(self, pdrive, stage):
        if self.multilevel:
            return self.psmr.PackAndSendFC(self.b0, self.coarse_b0)
        st = self.pbval_b.PackAndSendFC(self.b0)
        if self.fused and self.pbval_u.peers:
            self._stage_phase(pdrive, stage, capi.PHASE_C2P)
        return st

    def RecvB(self, pdrive, stage):
        if self.multilevel:
            return self.psmr.RecvAndUnpackFC(self.b0, self.coarse_b0)
        if self.fused and self.pbval_u.peers:
            self.pbval_u.RecvAndUnpackCC(self.u0)
        return self.pbval_b.RecvAndUnpackFC(self.b0)

    def ApplyPhysicalBCs(self, pdrive, stage):
        """mhd_tasks.cpp:501-520"""
        if self.pmy_pack.pmesh.strictly_periodic:
            return TaskStatus.complete
        self.pbval_u.HydroBCs(self.u0)
        self.pbval_b.BFieldBCs(self.b0)
        pgen = self.pmy_pack.pmesh.pgen
        if pgen is not None and pgen.user_bcs:                   # mhd_tasks.cpp:514-517
            pgen.user_bcs_func()
        return TaskStatus.complete

    def ConToPrim(self, pdrive, stage):
        """mhd_tasks.cpp: all cells incl. ghosts; fused path: + CFL scan on the last stage"""
        n3, n2, n1 = self.pmy_pack.pmesh.mb_indcs.ncells
        if self.fused and getattr(self, "_interior_done", False):
            self._interior_done = False
            capi.check(self.L.akmi_mhd_c2p_shell(
                C.byref(self.pack_c), capi._p(self.u0), *self._b(self.b0), capi._p(self.w0),
                capi._p(self.bcc0), capi._p(self.counters), capi._stream()), "mhd_c2p_shell")
            return TaskStatus.complete
        if self.fused:
            do_dt = 1 if stage == pdrive.nexp_stages else 0
            ev = getattr(self, "stage_events", None)     # bench.py: the conversion belongs to the stage's launch group
            if ev is not None:
                import torch
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev.append((e0, e1))
                e0.record()
            capi.check(self.L.akmi_mhd_c2p_newdt(
                C.byref(self.pack_c), capi._p(self.u0), *self._b(self.b0), capi._p(self.w0),
                capi._p(self.bcc0), do_dt, capi._p(self.counters), capi._p(self.dt3),
                capi._stream()), "mhd_c2p_newdt")
            if ev is not None:
                e1.record()
            self._dt_ready = bool(do_dt)
            return TaskStatus.complete
        if self.multilevel and not self.kinematic:
            # refined meshes (task-granular chain): the conversion of the last stage carries the CFL scan of
            # NewTimeStep along (one pass over w0 less; same bits: the fused path does the same); the other stages use the
            # same entry without the scan: on large packs its kernel converts two cells per thread with 16-byte accesses
            # (960 blocks of 32^3: 1 380 against 1 480 us for akmi_mhd_c2p)
            do_dt = 1 if stage == pdrive.nexp_stages else 0
            capi.check(self.L.akmi_mhd_c2p_newdt(
                C.byref(self.pack_c), capi._p(self.u0), *self._b(self.b0), capi._p(self.w0),
                capi._p(self.bcc0), do_dt, capi._p(self.counters), capi._p(self.dt3),
                capi._stream()), "mhd_c2p_newdt")
            self._dt_ready = bool(do_dt)
            return TaskStatus.complete
        capi.check(self.L.akmi_mhd_c2p(
            C.byref(self.pack_c), capi._p(self.u0), *self._b(self.b0), capi._p(self.w0),
            capi._p(self.bcc0), 0, n1 - 1, 0, n2 - 1, 0, n3 - 1, capi._p(self.counters),
            capi._stream()), "mhd_c2p")
        return TaskStatus.complete

    def NewTimeStep(self, pdrive, stage):
        """mhd_newdt.cpp:31-174: last stage only"""
        if stage != pdrive.nexp_stages:
            return TaskStatus.complete
        if self.kinematic:                                       # mhd_newdt.cpp:56-73
            capi.check(self.L.akmi_kinematic_newdt(C.byref(self.pack_c), capi._p(self.w0),
                                                   capi._p(self.dt3), capi._stream()), "kinematic_newdt")
        elif not getattr(self, "_dt_ready", False):
            capi.check(self.L.akmi_mhd_newdt(C.byref(self.pack_c), capi._p(self.w0),
                                             capi._p(self.bcc0), capi._p(self.dt3), capi._stream()),
                       "mhd_newdt")
        self._dt_ready = False
        self._finish_newdt()
        self._diffusion_newdt()
        return TaskStatus.complete
