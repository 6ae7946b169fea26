"""gpu: run-time options of the fused stage that change HOW a result is computed, never the result.  Each is read
once per process, so every case runs in a process of its own and must be bit-identical to the oracle:
  AKMI_MERGE_C2P=0  c2p of the active cells inside the stage call + c2p of the ghost shell afterwards (the order a
                    rank with off-rank neighbours uses) instead of one conversion after the ghost fill
  AKMI_OUT_OF_PLACE=0  C++ host: CopyCons + in-place first stage instead of the out-of-place stage with swapped registers
(the sign-word, two-kernel-x12 and k_march3ct options of rounds 3-4 lost their measurements and left the source in round 5)
  AKMI_MHD_ONE_KERNEL=1  the three MHD sweeps + update in one tile kernel (k_mhd_stage3d; opt-in, slower)
  AKMI_FUSE_C2P=0 / AKMI_HS2=0 / AKMI_STREAM_NONBLOCKING=1  hydro one-kernel stage: see test_hydro_stage_option_does_not_change_a_bit
C++ host, single rank, uniform mesh:
  AKMI_RUN_AHEAD=0  the new time step read back at the end of every cycle (one host synchronisation per cycle) instead of
                    Mesh::NewTimeStep on the device with the host one cycle ahead (default when eligible)
  AKMI_FOLD_BCS=0   same-rank gather + one kernel per bounded direction instead of akmi_bvals_*_local_bcs (both hosts)
  AKMI_C2P_PAIRS=0  ConsToPrim with one cell per thread at every size (two cells per thread from 4 M cells per launch otherwise;
                    the full-size tests of test_gpu_fullsize.py run the paired kernel against the oracle)
refined meshes:
  AKMI_SMR_DIRECT=0        same-level cell-centred ghost zones through the pack/unpack buffers instead of directly
  AKMI_SMR_LISTS=0         the level-boundary kernels launched over all nmb*56 (block, slot) pairs instead of the work
                           lists of akmi_smr_build_lists
  AKMI_SMR_FC_MAP=0        face fields of refined meshes through akmi_smr_pack_fc + the slot-by-slot akmi_smr_unpack_fc
                           instead of the copy list of akmi_smr_fc_map (default)
  AKMI_SMR_CC_MAP=0        cell-centred variables of refined meshes through pack / unpack / same-level gather instead of
                           the copy list of akmi_smr_cc_map (default on one rank)
  AKMI_SMR_EMF_SPLIT=1     the four kernels of RecvAndUnpackFluxFC (sum, zero, sum, average) instead of one launch per
                           (MeshBlock, component) that runs them back to back
  AKMI_FACE_SWEEPS=0       task-granular path, small packs and MeshBlocks of up to 96 cells per side: x2/x3 fluxes by the marching sweeps instead of one thread per face
  AKMI_TASK_OOP=0          task-granular path: CopyCons + in-place RKUpdate / CT on the first stage instead of the
                           out-of-place update with swapped registers (akmi_rk_update_oop, akmi_mhd_ct_oop)
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SCRIPT = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import parity_util as pu
# fused=True: the fixtures are small packs, which the hosts would otherwise hand to the task-granular chain
for native in (False, True):
    for n, mb, kw in ((32, 32, {}), (32, 16, {}), (40, 20, {}), ((66, 34, 18), (66, 34, 18), {}),
                      (32, 16, dict(integrator="rk3")), (24, 24, dict(integrator="rk1"))):
        r = pu.compare_run("orszag_tang", n=n, dims=3, mb=mb, cycles=3, native=native, fused=True, **kw)
        assert r["bitwise_equal"] and r["cycles"] == 3 and r["dt"][0] == r["dt"][1], (native, n, mb, kw, r)
r = pu.compare_run("blast", 32, 3, 16, cycles=3, fused=True, recon="plm")
assert r["bitwise_equal"], r
r = pu.compare_run("sod", n=32, dims=3, mb=16, cycles=3, fused=True)
assert r["bitwise_equal"], r
print("ok")
""" % (ROOT, os.path.join(ROOT, "tests"))


@pytest.mark.parametrize("env", [{"AKMI_MERGE_C2P": "0"}, {"AKMI_OUT_OF_PLACE": "0"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in sorted(e.items())))
def test_option_does_not_change_a_bit(env):
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=dict(os.environ, **env), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("env", [{"AKMI_FUSE_C2P": "0"}, {"AKMI_HS2": "0"}, {"AKMI_STREAM_NONBLOCKING": "1"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in sorted(e.items())))
def test_hydro_stage_option_does_not_change_a_bit(env):
    """the hydro one-kernel stage: AKMI_FUSE_C2P=0 ConsToPrim as a pass of its own after the ghost fill instead of inside the
    stage kernel (akmi_hydro_stage_w + second primitive array); AKMI_HS2=0 the one-plane kernel of round 5 (which also
    excludes the conversion inside it); AKMI_STREAM_NONBLOCKING=1 the C++ host's stream created non-blocking.  The sweep of
    tools/h3_check.py (shapes, decompositions, DC / PLM, every Riemann solver, isothermal, passive scalars, RK1-3, both hosts)
    stays bitwise equal to the oracle -- as it is with the defaults (test_gpu_parity.py, test_gpu_schemes.py)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "h3_check.py")], env=dict(os.environ, **env),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "bad: 0" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_mhd_one_kernel_stage_is_bit_identical():
    """AKMI_MHD_ONE_KERNEL=1: k_mhd_stage3d (x1/x2/x3 sweeps + RK update of a 3-D MHD PLM + HLLD stage in one tile kernel,
    csrc/akmi_mhd_stage3d.hpp) instead of k_sweep12s + the x3 march.  Measured slower (profiles/r06_mhd_stage3d.txt), kept
    as an opt-in path; the sweep of tools/m3_check.py (Orszag-Tang in several shapes and decompositions, RK1-3, blast at
    ng = 4, a linear wave on two blocks with cell sizes that are no powers of two, both hosts) has to stay bitwise equal."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "m3_check.py")],
                       env=dict(os.environ, AKMI_MHD_ONE_KERNEL="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "bad: 0" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


SMR_SCRIPT = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import parity_util as pu
CASES = (("linear_wave_mhd_smr", (32, 16, 16), (8, 4, 4), dict(rsolver="hlld")),
         ("linear_wave_mhd_smr", (32, 16, 16), (8, 8, 8), dict(recon="ppm4", ng=4, rsolver="hlld")),
         ("blast_smr", (32, 32, 32), (8, 8, 8), {}),
         ("linear_wave_mhd_smr", (32, 16, 16), (8, 4, 4),
          dict(rsolver="hlld", integrator="rk3", extra=("refined_region1/level=2",))))
# the C++ host on every mesh, the Python host on the first and on config 5's shape
for native, cases in ((True, CASES), (False, (CASES[0], CASES[2]))):
    for problem, n, mb, kw in cases:
        r = pu.compare_run(problem, n, 3, mb, cycles=2, native=native, **kw)
        assert r["bitwise_equal"] and r["cycles"] == 2 and r["dt"][0] == r["dt"][1], (native, problem, kw, r)
print("ok")
""" % (ROOT, os.path.join(ROOT, "tests"))


@pytest.mark.parametrize("env", [{"AKMI_SMR_DIRECT": "0"}, {"AKMI_TASK_OOP": "0"},
                                 {"AKMI_SMR_LISTS": "0"}, {"AKMI_FACE_SWEEPS": "0"}, {"AKMI_SMR_FC_MAP": "0"}, {"AKMI_SMR_CC_MAP": "0"},
                                 {"AKMI_SMR_EMF_SPLIT": "1"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in sorted(e.items())))
def test_smr_option_does_not_change_a_bit(env):
    r = subprocess.run([sys.executable, "-c", SMR_SCRIPT], env=dict(os.environ, **env), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


MARCH_SCRIPT = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import parity_util as pu
import test_gpu_schemes as sc
for case in sc.MULTI_D:
    problem, n, dims, mb, cycles, kw = case
    if dims != 3:
        continue
    for native in (False, True):
        r = pu.compare_run(problem, n, dims, mb, cycles, fused=False, native=native, **kw)
        assert r["bitwise_equal"] and r["cycles"] == cycles, (case, native, r)
for n, mb in ((32, 32), (32, 16), (40, 20)):
    for problem in ("orszag_tang", "sod"):
        r = pu.compare_run(problem, n=n, dims=3, mb=mb, cycles=3, fused=False)
        assert r["bitwise_equal"], (problem, n, mb, r)
print("ok")
""" % (ROOT, os.path.join(ROOT, "tests"))


def test_storing_marches_on_small_packs():
    """akmi_*_fluxes run their x2/x3 sweeps as one thread per face for packs up to 0.7 M cells -- every fixture of
    the suite -- and as marches that store their fluxes above.  AKMI_FACE_SWEEPS=0 puts the marches back: the 3-D
    task-granular cases of the scheme matrix through both hosts, bit-identical to the oracle."""
    r = subprocess.run([sys.executable, "-c", MARCH_SCRIPT], env=dict(os.environ, AKMI_FACE_SWEEPS="0"),
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


RUN_SCRIPT = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import parity_util as pu
# whole runs in ONE Execute call (the host enqueues cycle n+1 while cycle n runs), the last cycle clipped at tlim:
# end state, clock, next dt and cycle count against the oracle's run
for problem, n, dims, mb, kw in (("sod", 32, 3, 16, dict(cfl=0.3, extra=["time/tlim=0.06"])),
                                 ("sod", 64, 1, 32, dict(cfl=0.3, extra=["time/tlim=0.05"])),
                                 ("orszag_tang", 32, 3, 16, dict(cfl=0.3, extra=["time/tlim=0.03"])),
                                 ("blast", 24, 3, 12, dict(extra=["time/tlim=0.05"])),
                                 ("orszag_tang", 32, 2, 16, dict(cfl=0.3, integrator="rk3", extra=["time/tlim=0.03"])),
                                 ("sod", 32, 3, 16, dict(cfl=0.3, integrator="rk4", extra=["time/tlim=0.03"])),
                                 # MHD with bounded faces in two directions, 2 x 2 x 2 MeshBlocks: cell- and face-centred gathers
                                 # that compose a boundary function with a neighbour (edges and corners of the blocks)
                                 ("orszag_tang", 32, 3, 16, dict(cfl=0.3, extra=["time/tlim=0.02", "mesh/ix1_bc=outflow",
                                                                               "mesh/ox1_bc=reflect", "mesh/ix2_bc=reflect",
                                                                               "mesh/ox2_bc=outflow"])),
                                 ("blast", 24, 3, 12, dict(recon="plm", extra=["time/tlim=0.03", "mesh/ix3_bc=diode", "mesh/ox3_bc=diode",
                                                                              "mesh/ix1_bc=outflow", "mesh/ox1_bc=outflow"])),
                                 ("orszag_tang", 24, 3, 24, dict(cfl=0.3, integrator="rk1", extra=["time/tlim=0.02"]))):
    for native in (True, False):
        sim, osim, is_mhd = pu.make_pair(problem, n, dims, mb, fused=True, native=native, **kw)
        sim.Execute()
        osim.run()
        d = pu.compare_fields(pu.product_arrays(sim), pu.oracle_arrays(osim, is_mhd), is_mhd)
        assert d["bitwise_equal"], (problem, native, d)
        assert sim.pmesh.time == osim.time == osim.tlim and sim.pmesh.dt == osim.dt, (problem, native, sim.pmesh.time, osim.time, sim.pmesh.dt, osim.dt)
        assert sim.pmesh.ncycle == osim.ncycle and osim.ncycle > 3, (problem, sim.pmesh.ncycle, osim.ncycle)
print("ok")
""" % (ROOT, os.path.join(ROOT, "tests"))


@pytest.mark.parametrize("env", [{}, {"AKMI_FOLD_BCS": "0", "AKMI_RUN_AHEAD": "0", "AKMI_C2P_PAIRS": "0"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in sorted(e.items())) or "defaults")
def test_whole_runs_to_tlim_in_one_execute_call(env):
    r = subprocess.run([sys.executable, "-c", RUN_SCRIPT], env=dict(os.environ, **env), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
