"""Driver: RK stage tables and the main loop over task lists.

Mirror of src/driver/driver.cpp:93-162 (RK weights), :290-307 (ExecuteTaskList), :314-371
(Initialize), :380-459 (Execute), :569-653 (InitBoundaryValuesAndPrimitives).
"""
import time as _time

from .tasklist import TaskListStatus


class Driver:
    def __init__(self, pin, pmesh):
        self.time_evolution = pin.GetOrAddString("time", "evolution", "dynamic")
        if self.time_evolution not in ("dynamic", "kinematic"):
            raise RuntimeError("### FATAL ERROR <time> evolution = '%s' is not on this path "
                               "(dynamic, kinematic)" % self.time_evolution)
        self.integrator = pin.GetOrAddString("time", "integrator", "rk2")
        self.tlim = pin.GetReal("time", "tlim")
        self.nlim = pin.GetOrAddInteger("time", "nlim", -1)
        self.ndiag = pin.GetOrAddInteger("time", "ndiag", 1)
        self.gam0, self.gam1, self.beta, self.delta = [0.0]*4, [0.0]*4, [0.0]*4, [0.0]*4
        if self.integrator == "rk1":
            self.nexp_stages, self.cfl_limit = 1, 1.0
            self.gam0[0], self.gam1[0], self.beta[0] = 0.0, 1.0, 1.0
        elif self.integrator == "rk2":
            self.nexp_stages, self.cfl_limit = 2, 1.0
            self.gam0[0], self.gam1[0], self.beta[0] = 0.0, 1.0, 1.0
            self.gam0[1], self.gam1[1], self.beta[1] = 0.5, 0.5, 0.5
        elif self.integrator == "rk3":
            self.nexp_stages, self.cfl_limit = 3, 1.0
            self.gam0[0], self.gam1[0], self.beta[0] = 0.0, 1.0, 1.0
            self.gam0[1], self.gam1[1], self.beta[1] = 0.25, 0.75, 0.25
            self.gam0[2], self.gam1[2], self.beta[2] = 2.0/3.0, 1.0/3.0, 2.0/3.0
        elif self.integrator == "rk4":
            # RK4()4[2S] of Ketcheson (2010), driver.cpp:131-160
            self.nexp_stages, self.cfl_limit = 4, 1.3925
            self.gam0[0], self.gam1[0], self.beta[0] = 0.0, 1.0, 1.193743905974738
            self.gam0[1], self.gam1[1], self.beta[1] = (0.121098479554482, 0.721781678111411,
                                                        0.099279895495783)
            self.gam0[2], self.gam1[2], self.beta[2] = (-3.843833699660025, 2.121209265338722,
                                                        1.131678018054042)
            self.gam0[3], self.gam1[3], self.beta[3] = (0.546370891121863, 0.198653035682705,
                                                        0.310665766509336)
            self.delta = [1.0, 0.217683334308543, 1.065841341361089, 0.0]
        else:
            raise RuntimeError("### FATAL ERROR integrator=%s not implemented. Valid choices on "
                               "this path are [rk1,rk2,rk3,rk4]." % self.integrator)
        self.nimp_stages = 0
        self.nmb_updated_ = 0
        self.run_time_ = 0.0

    def ExecuteTaskList(self, pm, tl, stage):
        """driver.cpp:290-307 (one pack per rank)"""
        pmbp = pm.pmb_pack
        tlist = pmbp.tl_map[tl]
        if tlist.Empty():
            return
        tlist.Reset()
        while not tlist.IsComplete():
            if tlist.DoAvailable(self, stage) == TaskListStatus.complete:
                break

    def InitBoundaryValuesAndPrimitives(self, pm):
        """driver.cpp:569-653: one halo exchange + BCs + c2p everywhere"""
        self._begin_stage(pm)
        ph = pm.pmb_pack.phydro
        if ph is not None:
            ph.RestrictU(self, 0)
            ph.InitRecv(self, -1)
            ph.SendU(self, 0)
            ph.ClearSend(self, -1)
            ph.ClearRecv(self, -1)
            ph.RecvU(self, 0)
            ph.Prolongate(self, 0)
            ph.ApplyPhysicalBCs(self, 0)
            ph.ConToPrim(self, 0)
        pmhd = pm.pmb_pack.pmhd
        if pmhd is not None:
            pmhd.RestrictU(self, 0)
            pmhd.RestrictB(self, 0)
            pmhd.InitRecv(self, -1)
            pmhd.SendU(self, 0)
            pmhd.RecvU(self, 0)
            pmhd.SendB(self, 0)
            pmhd.RecvB(self, 0)
            pmhd.ClearSend(self, -1)
            pmhd.ClearRecv(self, -1)
            pmhd.Prolongate(self, 0)
            pmhd.ApplyPhysicalBCs(self, 0)
            pmhd.ConToPrim(self, 0)

    def Initialize(self, pm, pin=None, pout=None, res_flag=False):
        """driver.cpp:314-371; with pout: the initial outputs (driver.cpp:340-346), which a restarted
        run does not repeat"""
        self.pout, self.pin_ = pout, pin
        self.InitBoundaryValuesAndPrimitives(pm)
        ph, pmhd = pm.pmb_pack.phydro, pm.pmb_pack.pmhd
        if ph is not None:
            ph.NewTimeStep(self, self.nexp_stages)
        if pmhd is not None:
            pmhd.NewTimeStep(self, self.nexp_stages)
        pm.NewTimeStep(self.tlim)
        self.nmb_updated_ = 0
        if pout is not None and not res_flag:
            pout.MakeOutputs(pm, pin)

    @staticmethod
    def _begin_stage(pm):
        """the hand-shake flags between SendU / SendB and ApplyPhysicalBCs / ConToPrim live inside one stage: none may be
        left standing for the next one (a task list that drops the consumer)"""
        pk = pm.pmb_pack
        for ph in (getattr(pk, "phydro", None), getattr(pk, "pmhd", None)):
            if ph is None:
                continue
            ph._dt3_reset = False
            ph._want_ghost_c2p = False
            ph._shell_done = False
            for bv in (getattr(ph, "pbval_u", None), getattr(ph, "pbval_b", None)):
                if bv is not None and hasattr(bv, "_u_bcs_done"):
                    bv._u_bcs_done = bv._b_bcs_done = False

    def _cycle(self, pm):
        self.ExecuteTaskList(pm, "before_timeintegrator", 0)
        for stage in range(1, self.nexp_stages + 1):
            self._begin_stage(pm)
            self.ExecuteTaskList(pm, "before_stagen", stage)
            self.ExecuteTaskList(pm, "stagen", stage)
            self.ExecuteTaskList(pm, "after_stagen", stage)
        self.ExecuteTaskList(pm, "after_timeintegrator", 1)
        pm.time = pm.time + pm.dt
        pm.ncycle += 1
        self.nmb_updated_ += pm.nmb_total
        if getattr(self, "pout", None) is not None:
            self.pout.TestAndMakeOutputs(pm, self.pin_, self.tlim)     # driver.cpp:432-445
        pm.NewTimeStep(self.tlim)

    def Finalize(self, pm, pin, pout=None):
        """driver.cpp:467-500: final outputs, then the problem generator's final work (the
        linear-wave error file)"""
        if pout is not None:
            pout.MakeOutputs(pm, pin)
        if pm.pgen is not None and pm.pgen.pgen_final_func is not None:
            pm.pgen.write_errors_file = True
            errs = pm.pgen.pgen_final_func()
            pm.pgen.write_errors_file = False
            return errs
        return None

    def Execute(self, pm, pin=None, max_cycles=None):
        """driver.cpp:380-459; returns cycles executed by this call"""
        n = 0
        t0 = _time.time()
        while (pm.time < self.tlim) and (pm.ncycle < self.nlim or self.nlim < 0):
            if max_cycles is not None and n >= max_cycles:
                break
            self._cycle(pm)
            n += 1
        self.run_time_ += _time.time() - t0
        return n

    def zone_cycles_per_second(self, pm):
        """driver.cpp:513-522"""
        return self.nmb_updated_*pm.NumberOfMeshBlockCells()/max(self.run_time_, 1e-30)
