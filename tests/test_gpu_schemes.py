"""gpu: every reconstruction x Riemann-solver pair of the reference's lwave1d matrix
(test_nr_lwave1d_cpu.py:98-105: plm/ppm4/ppmx/wenoz x hydro llf/hlle/hllc/roe, mhd llf/hlle/hlld;
plus dc and teno) on the HIP path -- fused stage kernels, task-granular kernels and the C++
driver -- against the CPU oracle.  Bar: bit-identical (the oracle itself is pinned on the
reference's thresholds for the whole matrix in test_oracle_pins.py)."""
import ctypes as C

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import parity_util as pu  # noqa: E402
from oracle import akref  # noqa: E402

RECONS = ["dc", "plm", "ppm4", "ppmx", "wenoz", "teno"]
RS = {"hydro": ["llf", "hlle", "hllc", "roe"], "mhd": ["llf", "hlle", "hlld"]}


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "split"])
@pytest.mark.parametrize("recon", RECONS)
@pytest.mark.parametrize("soe", ["hydro", "mhd"])
def test_lwave1d_matrix_is_bit_identical(soe, recon, fused):
    """the reference's lwave1d run arguments (N=64, 4 MeshBlocks, ng=3, cfl 0.4), RK2 and RK3,
    every solver: 12 cycles each"""
    for rs in RS[soe]:
        for integ in ("rk2", "rk3"):
            res = pu.compare_run("linear_wave_%s" % soe, 64, 1, 16, 12, fused=fused, ng=3,
                                 recon=recon, rsolver=rs, integrator=integ, cfl=0.4,
                                 extra=["problem/along_x1=true", "problem/amp=1.0e-6"])
            assert res["cycles"] == 12
            assert res["time"][0] == res["time"][1]
            assert res["bitwise_equal"], (soe, recon, rs, integ, res["diffs"])


MULTI_D = [
    # problem, n, dims, mb, cycles, kwargs
    ("orszag_tang", 24, 3, 12, 3, dict(cfl=0.3, ng=3, recon="wenoz", rsolver="hlle")),
    ("orszag_tang", 24, 3, 12, 3, dict(cfl=0.3, ng=3, recon="ppmx", rsolver="llf")),
    ("orszag_tang", 24, 3, 24, 3, dict(cfl=0.3, ng=3, recon="teno", rsolver="hlld", integrator="rk3")),
    ("orszag_tang", 32, 2, 16, 4, dict(cfl=0.3, ng=3, recon="wenoz", rsolver="llf")),
    ("orszag_tang", 32, 2, 16, 4, dict(cfl=0.3, recon="plm", rsolver="hlle")),
    ("blast", 24, 3, 12, 3, dict(rsolver="hlle")),                       # ppm4, ng=4, strong shock
    ("blast", 32, 2, 16, 4, dict(recon="ppmx", rsolver="llf")),
    ("sod", 32, 3, 16, 4, dict(cfl=0.3, ng=3, recon="ppmx", rsolver="roe")),
    ("sod", 32, 3, 16, 4, dict(cfl=0.3, ng=3, recon="wenoz", rsolver="llf")),
    ("sod", 32, 3, 32, 4, dict(cfl=0.3, recon="plm", rsolver="hlle", integrator="rk3")),
    ("sod", 64, 2, 32, 5, dict(cfl=0.3, ng=3, recon="teno", rsolver="roe")),
    ("linear_wave_hydro", 24, 3, 12, 3, dict(ng=3, recon="ppm4", rsolver="roe")),
    # Ryu-Jones 2a (test_nr_rj2a_cpu.py): two blocks, outflow, every MHD solver
    ("rj2a", 256, 1, 128, 12, dict(cfl=0.3, rsolver="llf")),
    ("rj2a", 256, 1, 128, 12, dict(cfl=0.3, ng=3, recon="ppmx", rsolver="hlle", integrator="rk3")),
    ("rj2a", 256, 1, 128, 12, dict(cfl=0.3, ng=3, recon="wenoz", rsolver="hlld", integrator="rk3")),
]


def _id(c):
    return "%s-%d^%d-mb%d-%s-%s" % (c[0], c[1], c[2], c[3], c[5].get("recon", "deck"), c[5]["rsolver"])


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "split"])
@pytest.mark.parametrize("case", MULTI_D, ids=_id)
def test_multi_d_schemes_are_bit_identical(case, fused):
    """marching x2/x3 kernels, CT-extended sweeps, multi-block halos with the wider stencils"""
    problem, n, dims, mb, cycles, kw = case
    res = pu.compare_run(problem, n, dims, mb, cycles, fused=fused, **kw)
    assert res["cycles"] == cycles
    assert res["time"][0] == res["time"][1], res["time"]
    assert res["bitwise_equal"], res["diffs"]


@pytest.mark.parametrize("case", [MULTI_D[0], MULTI_D[7], MULTI_D[6]], ids=_id)
def test_native_cpp_driver_schemes(case):
    problem, n, dims, mb, cycles, kw = case
    res = pu.compare_run(problem, n, dims, mb, cycles, native=True, **kw)
    assert res["cycles"] == cycles and res["bitwise_equal"], res["diffs"]


# hydro DC/PLM in 3-D: the whole stage update is one kernel (k_hydro_stage3d) whose tile shape depends on
# the block size -- every hydro solver, both reconstructions, blocks that do not fill the tiles, many small
# blocks, outflow faces, the phased issue of multi-stage integrators
HYDRO_ONE_KERNEL = [
    ("sod", 24, 3, 12, 4, dict(cfl=0.3, recon="dc", rsolver="llf")),
    ("sod", 24, 3, 24, 4, dict(cfl=0.3, recon="dc", rsolver="hllc", integrator="rk3")),
    ("sod", (36, 28, 20), 3, (18, 14, 10), 4, dict(cfl=0.3, recon="plm", rsolver="roe", integrator="rk3")),
    ("sod", (70, 12, 10), 3, (70, 12, 10), 3, dict(cfl=0.3, recon="plm", rsolver="hllc")),
    ("sod", 32, 3, 8, 3, dict(cfl=0.3, recon="plm", rsolver="hlle", integrator="rk1")),
    ("sod", 40, 3, 40, 3, dict(cfl=0.3, ng=3, recon="plm", rsolver="hllc",
                               extra=("mesh/ix2_bc=outflow", "mesh/ox2_bc=outflow", "mesh/ix3_bc=reflect",
                                      "mesh/ox3_bc=reflect"))),
    ("linear_wave_hydro", 32, 3, 16, 3, dict(recon="plm", rsolver="llf", integrator="rk4")),
    # oblique waves: every flux component of every direction is non-trivial
    ("linear_wave_hydro", (32, 16, 24), 3, (16, 16, 8), 3, dict(recon="plm", rsolver="hllc")),
    ("linear_wave_hydro", (20, 24, 28), 3, (20, 12, 14), 3, dict(recon="dc", rsolver="hlle", integrator="rk3")),
    ("linear_wave_hydro", 24, 3, 24, 3, dict(recon="plm", rsolver="roe", ng=4)),
    # blocks smaller than any tile / too narrow for one (falls back to the three-kernel sequence)
    ("linear_wave_hydro", (8, 4, 4), 3, (4, 2, 2), 3, dict(recon="plm", rsolver="hllc")),
    ("linear_wave_hydro", (4, 8, 8), 3, (2, 4, 4), 3, dict(recon="plm", rsolver="hllc")),
]


def _id1(c):
    n = c[1] if isinstance(c[1], int) else "x".join(map(str, c[1]))
    mb = c[3] if isinstance(c[3], int) else "x".join(map(str, c[3]))
    return "%s-%s-mb%s-%s-%s-%s" % (c[0], n, mb, c[5]["recon"], c[5]["rsolver"], c[5].get("integrator", "rk2"))


@pytest.mark.parametrize("case", HYDRO_ONE_KERNEL, ids=_id1)
def test_hydro_one_kernel_stage_is_bit_identical(case):
    problem, n, dims, mb, cycles, kw = case
    res = pu.compare_run(problem, n, dims, mb, cycles, fused=True, **kw)
    assert res["cycles"] == cycles
    assert res["time"][0] == res["time"][1], res["time"]
    assert res["bitwise_equal"], res["diffs"]


RK4 = [
    ("linear_wave_hydro", 64, 1, 32, 8, dict(integrator="rk4", recon="wenoz", ng=3, rsolver="hllc")),
    ("sod", 24, 3, 12, 4, dict(integrator="rk4", cfl=0.3, rsolver="hlle")),
    ("sod", 32, 2, 16, 5, dict(integrator="rk4", cfl=0.3, recon="ppm4", ng=3, rsolver="roe")),
]


@pytest.mark.parametrize("mode", ["fused", "split", "native"])
@pytest.mark.parametrize("case", RK4, ids=_id)
def test_rk4_is_bit_identical(case, mode):
    """integrator rk4: u1 += delta*u0 on the active cells before stages 2-4 (akmi_rk4_copy_cons)"""
    problem, n, dims, mb, cycles, kw = case
    res = pu.compare_run(problem, n, dims, mb, cycles, fused=(mode != "split"),
                         native=(mode == "native"), **kw)
    assert res["cycles"] == cycles and res["time"][0] == res["time"][1]
    assert res["bitwise_equal"], res["diffs"]


# <hydro>/fofc: receding streams (see test_fofc_rescues_double_rarefaction) along x1/x2/x3 so that cells
# are flagged next to MeshBlock faces in every direction; the last case floods the flags (density
# floor above the ambient density) and the isothermal one takes the density-only test
def _rare(d, v=4.0, *more):
    # ul/ur are the velocities along shock_dir (shock_tube.cpp rotates the state)
    return ("problem/shock_dir=%d" % d, "problem/ul=%r" % -v, "problem/ur=%r" % v, "problem/dr=1.0",
            "problem/pl=0.4", "problem/pr=0.4", "hydro/fofc=true") + more


FOFC = [
    ("sod", 128, 1, 64, 40, dict(cfl=0.4, recon="ppm4", ng=4, rsolver="hllc", extra=_rare(1))),
    ("sod", 128, 1, 32, 23, dict(cfl=0.4, recon="wenoz", ng=4, rsolver="hllc", integrator="rk3", extra=_rare(1))),
    ("sod", 32, 2, 16, 25, dict(cfl=0.3, recon="ppm4", ng=4, rsolver="hllc", extra=_rare(2))),
    ("sod", 32, 2, (32, 8), 22, dict(cfl=0.3, recon="wenoz", ng=4, rsolver="hllc", extra=_rare(1))),
    ("sod", 24, 3, 12, 20, dict(cfl=0.3, recon="ppm4", ng=4, rsolver="hllc", extra=_rare(3))),
    ("sod", 64, 1, 32, 12, dict(cfl=0.3, ng=3, rsolver="roe",
                                extra=_rare(1, 8.0, "hydro/eos=isothermal", "hydro/iso_sound_speed=0.5",
                                            "hydro/dfloor=0.05"))),
]


def _rare_mhd(d, v=8.0):
    tube = {"ul": -v, "ur": v, "vl": 0.0, "wl": 0.0, "dl": 1.0, "dr": 1.0, "pl": 0.04, "pr": 0.04,
            "bxl": 0.1, "bxr": 0.1, "byl": 0.2, "byr": 0.2, "bzl": 0.1, "bzr": 0.1}
    return ("problem/shock_dir=%d" % d, "mhd/fofc=true", "mhd/gamma=1.4") + tuple(
        "problem/%s=%r" % kv for kv in tube.items())


_BLAST = ("mhd/fofc=true", "problem/prat=1.0e4", "problem/b_amb=10.0")
FOFC += [
    ("rj2a", 128, 1, 64, 60, dict(cfl=0.3, recon="ppm4", ng=4, rsolver="hlld", extra=_rare_mhd(1))),
    ("rj2a", 128, 1, 32, 60, dict(cfl=0.3, recon="ppm4", ng=4, rsolver="hlle", integrator="rk3",
                                  extra=_rare_mhd(1))),
    ("rj2a", 32, 2, 16, 50, dict(cfl=0.3, recon="ppm4", ng=4, rsolver="hlld", extra=_rare_mhd(2))),
    ("blast", 32, 2, 16, 30, dict(rsolver="hlld", extra=_BLAST)),
    ("blast", 32, 2, (32, 8), 25, dict(rsolver="hlle", recon="wenoz", extra=_BLAST)),
    ("blast", 24, 3, 12, 16, dict(rsolver="hlld", extra=_BLAST)),
]


@pytest.mark.parametrize("native", [False, True], ids=["py", "cpp"])
@pytest.mark.parametrize("case", FOFC, ids=lambda c: "%s-%d^%d-%s-%s" % (c[0], c[1], c[2], c[5].get("recon", "deck"), c[5]["rsolver"]))
def test_fofc_is_bit_identical(case, native):
    """akmi_{hydro,mhd}_fluxes_fofc + akmi_{hydro,mhd}_fofc against the oracle, with cells actually
    flagged (MHD: the face EMFs of flagged cells are replaced too and feed CornerE/CT)"""
    problem, n, dims, mb, cycles, kw = case
    sim, osim, is_mhd = pu.make_pair(problem, n, dims, mb, **kw)
    for _ in range(cycles):
        assert sim.Execute(max_cycles=1) and osim.step()
    assert osim.nfofc > 0
    if not native:
        assert int(sim.phys.nfofc.item()) == osim.nfofc
    d = pu.compare_fields(pu.product_arrays(sim), pu.oracle_arrays(osim, is_mhd), is_mhd)
    assert sim.pmesh.time == osim.time and d["bitwise_equal"], d


DIFFUSION = [
    # viscosity + conduction, hydro: 1-D (cell-shaped flux arrays), 2-D, 3-D multi-block
    ("linear_wave_hydro", 64, 1, 32, 12, dict(rsolver="hllc"), dict(nu_iso=0.01, alpha_iso=0.02)),
    ("sod", 32, 2, 16, 6, dict(cfl=0.3, rsolver="hlle"), dict(nu_iso=0.005, alpha_iso=0.01)),
    ("sod", 24, 3, 12, 4, dict(cfl=0.3, rsolver="hllc", recon="ppm4", ng=3), dict(nu_iso=0.004, alpha_iso=0.003)),
    ("sod", 24, 3, 12, 4, dict(cfl=0.3, rsolver="roe", ng=3, extra=("hydro/fofc=true",)), dict(nu_iso=0.004)),
    # viscosity + conduction + Ohmic resistivity, MHD (face-shaped fluxes, edge EMFs) in 1-D/2-D/3-D
    ("linear_wave_mhd", 64, 1, 32, 10, dict(rsolver="hlld"), dict(nu_iso=0.01, alpha_iso=0.01, eta_ohm=0.02)),
    ("orszag_tang", 32, 2, 16, 6, dict(rsolver="hlld"), dict(eta_ohm=0.002, nu_iso=0.002)),
    ("orszag_tang", 24, 3, 12, 4, dict(rsolver="hlle", cfl=0.3), dict(eta_ohm=0.002, alpha_iso=0.004)),
    ("blast", 24, 3, 12, 4, dict(rsolver="hlld"), dict(eta_ohm=0.003, nu_iso=0.003, alpha_iso=0.003)),
    ("linear_wave_mhd", 32, 2, (16, 32), 5, dict(rsolver="hlld", extra=("mhd/eos=isothermal",)),
     dict(eta_ohm=0.003, nu_iso=0.003)),
    # ambipolar diffusion (isothermal MHD): EMFs from edge-averaged J and B, cell-reduced time step
    ("linear_wave_mhd", 64, 1, 32, 8, dict(rsolver="hlld", extra=("mhd/eos=isothermal",)), dict(eta_ad=0.01)),
    ("linear_wave_mhd", 32, 2, 16, 6, dict(rsolver="hlle", extra=("mhd/eos=isothermal", "problem/amp=1.0e-2")),
     dict(eta_ad=0.02, eta_ohm=0.003)),
    ("linear_wave_mhd", 24, 3, 12, 4, dict(rsolver="hlld", extra=("mhd/eos=isothermal", "problem/amp=1.0e-2")),
     dict(eta_ad=0.01)),
    ("linear_wave_mhd", 24, 3, (12, 24, 8), 4, dict(rsolver="llf", extra=("mhd/eos=isothermal", "problem/amp=0.1")),
     dict(eta_ad=0.02)),
    # ideal gas: + the ambipolar Poynting flux in the energy equation
    ("linear_wave_mhd", 64, 1, 32, 8, dict(rsolver="hlld", extra=("problem/amp=1.0e-2",)), dict(eta_ad=0.01)),
    ("orszag_tang", 32, 2, 16, 6, dict(rsolver="hlld"), dict(eta_ad=0.01, eta_ohm=0.002)),
    ("orszag_tang", 24, 3, 12, 4, dict(rsolver="hlle", cfl=0.3), dict(eta_ad=0.01)),
    ("blast", 24, 3, (12, 24, 8), 4, dict(rsolver="hlld"), dict(eta_ad=0.002, nu_iso=0.001)),
]


@pytest.mark.parametrize("native", [False, True], ids=["py", "cpp"])
@pytest.mark.parametrize("case", DIFFUSION, ids=lambda c: "%s-%d^%d-%s" % (c[0], c[1], c[2], "+".join(sorted(c[6]))))
def test_diffusion_hooks_are_bit_identical(case, native):
    """akmi_viscous_fluxes / akmi_heat_fluxes / akmi_resistive_fluxes / akmi_resistive_emfs /
    akmi_conduction_newdt inside the task chain, and the diffusive time-step limits"""
    problem, n, dims, mb, cycles, kw, params = case
    sim, osim, is_mhd = pu.make_pair(problem, n, dims, mb, params=params, native=native, **kw)
    assert (sim.dt if native else sim.pmesh.dt) == osim.dt
    for _ in range(cycles):
        assert sim.Execute(max_cycles=1) and osim.step()
        assert sim.pmesh.dt == osim.dt
    d = pu.compare_fields(pu.product_arrays(sim), pu.oracle_arrays(osim, is_mhd), is_mhd)
    assert sim.pmesh.time == osim.time and d["bitwise_equal"], d


# ---- kinematic runs: <time>/evolution = kinematic, rsolver = advect ---------------------------------
_KIN = ("time/evolution=kinematic",)
KINEMATIC = [
    ("linear_wave_hydro", 64, 1, 32, 10, dict(rsolver="advect", extra=_KIN + ("problem/vx0=0.5",)), {}),
    ("sod", 32, 2, 16, 6, dict(rsolver="advect", cfl=0.3, recon="wenoz", ng=3,
                               extra=_KIN + ("problem/ul=0.7", "problem/ur=-0.4", "problem/vl=0.3")), dict(nu_iso=0.01)),
    ("sod", 24, 3, 12, 4, dict(rsolver="advect", cfl=0.3, recon="ppm4", ng=3,
                               extra=_KIN + ("problem/shock_dir=3", "problem/ul=-0.5", "problem/ur=0.6")),
     dict(alpha_iso=0.02, nu_iso=0.01)),
    ("linear_wave_hydro", 32, 2, 16, 6, dict(rsolver="advect", extra=_KIN + ("problem/vx0=0.5", "hydro/eos=isothermal")),
     dict(nu_iso=0.02)),
    # kinematic MHD: advect_mhd leaves the energy flux alone, so the resistive Poynting flux accumulates in it
    ("linear_wave_mhd", 64, 1, 32, 8, dict(rsolver="advect", extra=_KIN + ("problem/vx0=0.5",)), dict(eta_ohm=0.02)),
    ("orszag_tang", 32, 2, 16, 6, dict(rsolver="advect", extra=_KIN), dict(eta_ohm=0.005, nu_iso=0.002)),
    ("orszag_tang", 24, 3, 12, 4, dict(rsolver="advect", cfl=0.3, recon="wenoz", ng=3, extra=_KIN), dict(eta_ohm=0.004)),
    ("linear_wave_mhd", 32, 2, 16, 5, dict(rsolver="advect", extra=_KIN + ("problem/vx0=0.5", "mhd/eos=isothermal")),
     dict(eta_ohm=0.01)),
]


@pytest.mark.parametrize("native", [False, True], ids=["py", "cpp"])
@pytest.mark.parametrize("case", KINEMATIC, ids=lambda c: "%s-%d^%d" % (c[0], c[1], c[2]))
def test_kinematic_advect_runs_are_bit_identical(case, native):
    """advect_hyd + akmi_kinematic_newdt (+ the diffusion adders, which is what kinematic runs are for)"""
    problem, n, dims, mb, cycles, kw, params = case
    sim, osim, is_mhd = pu.make_pair(problem, n, dims, mb, params=params, native=native, **kw)
    assert (sim.dt if native else sim.pmesh.dt) == osim.dt
    for _ in range(cycles):
        assert sim.Execute(max_cycles=1) and osim.step()
        assert sim.pmesh.dt == osim.dt
    d = pu.compare_fields(pu.product_arrays(sim), pu.oracle_arrays(osim, is_mhd), is_mhd)
    assert sim.pmesh.time == osim.time and d["bitwise_equal"], d


@pytest.mark.parametrize("which", ["visc", "cond2d", "resist"])
def test_diffusion_pgen_matches_the_cpu_backend(which):
    """the reference's diffusion regressions (test_diffusion_{visc,conduct}_cpu.py) run against this
    implementation through the oracle-backed host (tools/run_reference_suite.sh); here the same
    set-up runs on the HIP kernels and must give the same error file, digit for digit"""
    import tempfile
    from athenak_amd.__main__ import main
    import cpu_backend
    args = {"visc": ["mesh/nx1=64", "meshblock/nx1=32", "problem/viscosity_test=true", "problem/vel_comp=3",
                     "hydro/nu_iso=0.25", "time/tlim=0.3"],
            "cond2d": ["mesh/nx1=32", "mesh/nx2=32", "meshblock/nx1=16", "meshblock/nx2=16",
                       "problem/conduction_test=true", "problem/spread_x2=true", "hydro/alpha_iso=0.5",
                       "time/tlim=0.2"],
            "resist": ["mesh/nx1=64", "meshblock/nx1=32", "problem/vel_comp=3", "time/tlim=0.3"]}[which]
    deck, base = ("diffusion_mhd.athinput", "diffusion_resist") if which == "resist" else \
                 ("diffusion.athinput", "diffusion")
    here = os.getcwd()
    out = []
    try:
        for backend in ("hip", "oracle"):
            with tempfile.TemporaryDirectory() as d:
                if backend == "oracle":
                    cpu_backend.install()
                try:
                    assert main(["-i", deck, "-d", d] + args) == 0
                finally:
                    os.chdir(here)
                    if backend == "oracle":
                        cpu_backend.uninstall()
                out.append(open(os.path.join(d, base + "-errs.dat")).read())
    finally:
        os.chdir(here)
    assert out[0] == out[1]
    err = float(out[0].splitlines()[-1].split()[4])
    assert 0.0 < err < 2e-9


@pytest.mark.parametrize("fused", [True, False, "cpp"], ids=["fused", "split", "cpp"])
@pytest.mark.parametrize("case", [
    ("sod", 64, 1, 32, 30, dict(cfl=0.4, rsolver="hllc", extra=("mesh/ix1_bc=diode", "mesh/ox1_bc=vacuum"))),
    ("sod", 32, 2, 16, 10, dict(cfl=0.3, rsolver="hlle", extra=("mesh/ix1_bc=reflect", "mesh/ox1_bc=diode",
                                                               "mesh/ix2_bc=diode", "mesh/ox2_bc=outflow"))),
    # (a vacuum face next to a magnetised gas produces NaN in the reference's arithmetic too: the field
    # is copied, the density is zero; MHD cases therefore use diode)
    ("rj2a", 64, 1, 32, 20, dict(cfl=0.3, rsolver="hlld", extra=("mesh/ix1_bc=diode", "mesh/ox1_bc=diode"))),
    ("blast", 24, 3, 12, 5, dict(rsolver="hlld", extra=("mesh/ix1_bc=diode", "mesh/ox1_bc=outflow", "mesh/ix2_bc=reflect",
                                                         "mesh/ox2_bc=diode", "mesh/ix3_bc=diode", "mesh/ox3_bc=outflow"))),
], ids=lambda c: "%s-%d^%d" % (c[0], c[1], c[2]))
def test_diode_and_vacuum_boundaries(case, fused):
    """hydro_bcs.cpp:105-118, bfield_bcs.cpp:88-97 in whole runs"""
    problem, n, dims, mb, cycles, kw = case
    res = pu.compare_run(problem, n, dims, mb, cycles, fused=(fused is not False), native=(fused == "cpp"), **kw)
    assert res["cycles"] == cycles and res["time"][0] == res["time"][1]
    assert res["bitwise_equal"], res["diffs"]


def _wild_states(shape5, rng, mhd):
    """primitive states with jumps of many decades between neighbouring cells: exercises the
    supersonic branches, the HLLE/HLLC pressure estimates, Roe's negative-density fallback and
    the L/R floors of ppmx/wenoz/teno"""
    w = np.empty(shape5)
    w[:, 0] = 10.0**rng.uniform(-4, 2, size=w[:, 0].shape)
    w[:, 1:4] = rng.normal(0, 3.0, size=w[:, 1:4].shape)
    w[:, 4] = 10.0**rng.uniform(-5, 2, size=w[:, 4].shape)
    return w


@pytest.mark.parametrize("recon", RECONS)
@pytest.mark.parametrize("rs", RS["hydro"])
def test_task_hydro_fluxes_wild_states(recon, rs):
    from athenak_amd import capi
    o = akref.Sim(nx1=20, nx2=12, nx3=8, mb_nx1=10, mb_nx2=12, mb_nx3=8, ng=3, nstages=2, cfl=0.3,
                  tlim=1.0, nlim=-1, is_mhd=0, recon="plm", rsolver="hllc", gamma=1.4,
                  pgen="shock_tube", shock_dir=1, xshock=0.0, wl=[1, 0, 0, 0, 1, 0, 0, 0],
                  wr=[0.125, 0, 0, 0, 0.1, 0, 0, 0], bcs=["outflow"]*6, dfloor=1e-3, pfloor=1e-4)
    o.initialize()
    L, R = capi.lib(), akref.lib()
    pk = o.pack()
    dxd = _t(o.array("dx"))
    pkd = capi.Pack.from_buffer_copy(bytes(pk))
    pkd.dx = dxd.data_ptr()
    rng = np.random.default_rng(1234)
    n3, n2, n1 = o.dims()
    w0 = _wild_states((o.nmb, 5, n3, n2, n1), rng, False)
    f = [np.zeros((o.nmb, 5, n3, n2, n1 + 1)), np.zeros((o.nmb, 5, n3, n2 + 1, n1)),
         np.zeros((o.nmb, 5, n3 + 1, n2, n1))]
    rc, sc = akref.RECON[recon], akref.RSOLVER[rs]
    assert R.akref_hydro_fluxes(C.byref(pk), rc, sc, akref.ptr(w0), *[akref.ptr(x) for x in f], 1) == 0
    fd = [_t(np.zeros_like(x)) for x in f]
    w0d = _t(w0)
    capi.check(L.akmi_hydro_fluxes(C.byref(pkd), rc, sc, capi._p(w0d), *[capi._p(x) for x in fd], 1,
                                   None), "fluxes")
    for a, b in zip(f, fd):
        b = b.cpu().numpy()
        assert np.isfinite(a).all()
        assert np.array_equal(a, b)


@pytest.mark.parametrize("recon", RECONS)
@pytest.mark.parametrize("rs", RS["mhd"])
def test_task_mhd_fluxes_wild_states(recon, rs):
    from athenak_amd import capi
    o = akref.Sim(nx1=16, nx2=12, nx3=8, mb_nx1=8, mb_nx2=12, mb_nx3=8, ng=3, nstages=2, cfl=0.3,
                  tlim=1.0, nlim=-1, is_mhd=1, recon="plm", rsolver="hlld", gamma=1.666666667,
                  pgen="orszag_tang", bcs=["periodic"]*6, dfloor=1e-3, pfloor=1e-4)
    o.initialize()
    L, R = capi.lib(), akref.lib()
    pk = o.pack()
    dxd = _t(o.array("dx"))
    pkd = capi.Pack.from_buffer_copy(bytes(pk))
    pkd.dx = dxd.data_ptr()
    rng = np.random.default_rng(4321)
    n3, n2, n1 = o.dims()
    h = {"w0": _wild_states((o.nmb, 5, n3, n2, n1), rng, True),
         "bcc0": rng.normal(0, 2.0, size=(o.nmb, 3, n3, n2, n1)),
         "b0x1f": rng.normal(0, 2.0, size=(o.nmb, n3, n2, n1 + 1)),
         "b0x2f": rng.normal(0, 2.0, size=(o.nmb, n3, n2 + 1, n1)),
         "b0x3f": rng.normal(0, 2.0, size=(o.nmb, n3 + 1, n2, n1))}
    h["b0x1f"][:, :, :, ::3] = 0.0          # Bx = 0 faces (degenerate HLLD branches)
    names_in = ["w0", "bcc0", "b0x1f", "b0x2f", "b0x3f"]
    names_out = ["flx1", "flx2", "flx3", "e3x1", "e2x1", "e1x2", "e3x2", "e2x3", "e1x3"]
    for k in names_out:
        h[k] = np.zeros_like(o.array(k))
    dv = {k: _t(v) for k, v in h.items()}
    rc, sc = akref.RECON[recon], akref.RSOLVER[rs]
    assert R.akref_mhd_fluxes(C.byref(pk), rc, sc, *[akref.ptr(h[k]) for k in names_in + names_out]) == 0
    capi.check(L.akmi_mhd_fluxes(C.byref(pkd), rc, sc, *[capi._p(dv[k]) for k in names_in + names_out],
                                 None), "mhd_fluxes")
    for k in names_out:
        assert np.isfinite(h[k]).all(), k
        assert np.array_equal(h[k], dv[k].cpu().numpy()), k


def test_unknown_scheme_is_rejected():
    """hydro has no hlld, mhd has no hllc/roe: the entry points fail loudly"""
    from athenak_amd import capi
    o = akref.Sim(nx1=8, nx2=1, nx3=1, mb_nx1=8, mb_nx2=1, mb_nx3=1, ng=2, nstages=2, cfl=0.3,
                  tlim=1.0, nlim=-1, is_mhd=0, recon="plm", rsolver="hllc", gamma=1.4,
                  pgen="shock_tube", shock_dir=1, xshock=0.0, wl=[1, 0, 0, 0, 1, 0, 0, 0],
                  wr=[0.125, 0, 0, 0, 0.1, 0, 0, 0], bcs=["outflow"]*6)
    o.initialize()
    L = capi.lib()
    pk = o.pack()
    dxd = _t(o.array("dx"))
    pkd = capi.Pack.from_buffer_copy(bytes(pk))
    pkd.dx = dxd.data_ptr()
    w0 = _t(o.array("w0"))
    n3, n2, n1 = o.dims()
    f = [_t(np.zeros((1, 5, n3, n2, n1 + 1))), _t(np.zeros((1, 5, n3, n2 + 1, n1))),
         _t(np.zeros((1, 5, n3 + 1, n2, n1)))]
    assert L.akmi_hydro_fluxes(C.byref(pkd), 1, 3, capi._p(w0), *[capi._p(x) for x in f], 1, None) < 0
    assert b"rsolver" in L.akmi_last_error()
    assert L.akmi_hydro_fluxes(C.byref(pkd), 4, 2, capi._p(w0), *[capi._p(x) for x in f], 1, None) < 0
    assert b"nghost" in L.akmi_last_error()


# ---- isothermal EOS (fused stage kernels: the isothermal solvers are their RS + 10; and the task-granular ones) ----
ISO_RS = {"hydro": ["llf", "hlle", "roe"], "mhd": ["llf", "hlle", "hlld"]}


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "split"])
@pytest.mark.parametrize("recon", ["plm", "ppm4", "ppmx", "wenoz"])
@pytest.mark.parametrize("soe", ["hydro", "mhd"])
def test_isothermal_lwave1d_matrix_is_bit_identical(soe, recon, fused):
    """run arguments of test_nr_isolwave1d_cpu.py (eos=isothermal, N=64, 4 blocks, ng=3) for every
    solver; initial data from the product's own problem generator (inject=False) so that the
    isothermal eigenvectors of pgen.py are compared with the oracle's as well"""
    for rs in ISO_RS[soe]:
        for wave in ((0, 3) if soe == "hydro" else (0, 2, 5)):
            res = pu.compare_run("linear_wave_%s" % soe, 64, 1, 16, 10, ng=3, recon=recon, rsolver=rs,
                                 integrator="rk2" if recon == "plm" else "rk3", cfl=0.4, inject=False, fused=fused,
                                 extra=["problem/along_x1=true", "problem/amp=1.0e-6",
                                        "problem/wave_flag=%d" % wave, "%s/eos=isothermal" % soe])
            assert res["cycles"] == 10 and res["time"][0] == res["time"][1]
            assert res["max_rel_l1"] <= pu.TOL, (soe, recon, rs, wave, res["diffs"])


@pytest.mark.parametrize("case", [
    ("linear_wave_hydro", 24, 3, 12, 3, dict(ng=3, recon="wenoz", rsolver="roe")),
    ("linear_wave_hydro", 24, 3, 12, 3, dict(recon="plm", rsolver="roe")),          # fused: the one-kernel hydro stage
    ("linear_wave_mhd", 24, 3, 12, 3, dict(recon="plm", rsolver="hlld", extra=("problem/amp=0.1",))),   # fused: x1 sweep + marches + CornerE/CT
    ("linear_wave_hydro", 32, 2, 16, 4, dict(recon="plm", rsolver="hlle")),
    ("linear_wave_mhd", 24, 3, 12, 3, dict(ng=3, recon="ppmx", rsolver="hlld", integrator="rk3")),
    ("linear_wave_mhd", 32, 2, 16, 4, dict(recon="plm", rsolver="llf")),
    ("sod", 64, 1, 32, 8, dict(cfl=0.3, recon="plm", rsolver="roe")),
    ("rj2a", 128, 1, 64, 8, dict(cfl=0.3, rsolver="hlld")),
], ids=lambda c: "%s-%d^%d-%s" % (c[0], c[1], c[2], c[5]["rsolver"]))
@pytest.mark.parametrize("native", [False, True], ids=["py", "cpp"])
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "split"])
def test_isothermal_multi_d_runs_are_bit_identical(case, native, fused):
    problem, n, dims, mb, cycles, kw = case
    blk = "hydro" if problem in ("linear_wave_hydro", "sod") else "mhd"
    kw = dict(kw)
    kw["extra"] = list(kw.get("extra", [])) + ["%s/eos=isothermal" % blk]
    res = pu.compare_run(problem, n, dims, mb, cycles, native=native, fused=fused, **kw)
    assert res["cycles"] == cycles and res["time"][0] == res["time"][1]
    assert res["bitwise_equal"], res["diffs"]


@pytest.mark.parametrize("rs", ["llf", "hlle", "roe"])
@pytest.mark.parametrize("recon", ["plm", "wenoz"])
def test_task_hydro_fluxes_wild_states_isothermal(recon, rs):
    from athenak_amd import capi
    o = akref.Sim(nx1=20, nx2=12, nx3=8, mb_nx1=10, mb_nx2=12, mb_nx3=8, ng=3, nstages=2, cfl=0.3,
                  tlim=1.0, nlim=-1, is_mhd=0, recon="plm", rsolver="llf", gamma=1.4, is_ideal=0,
                  iso_cs=0.7, pgen="shock_tube", shock_dir=1, xshock=0.0, wl=[1, 0, 0, 0, 1, 0, 0, 0],
                  wr=[0.125, 0, 0, 0, 0.1, 0, 0, 0], bcs=["outflow"]*6, dfloor=1e-3)
    o.initialize()
    L, R = capi.lib(), akref.lib()
    pk = o.pack()
    dxd = _t(o.array("dx"))
    pkd = capi.Pack.from_buffer_copy(bytes(pk))
    pkd.dx = dxd.data_ptr()
    rng = np.random.default_rng(99)
    n3, n2, n1 = o.dims()
    w0 = _wild_states((o.nmb, 5, n3, n2, n1), rng, False)[:, :4].copy()
    f = [np.zeros((o.nmb, 4, n3, n2, n1 + 1)), np.zeros((o.nmb, 4, n3, n2 + 1, n1)),
         np.zeros((o.nmb, 4, n3 + 1, n2, n1))]
    rc, sc = akref.RECON[recon], akref.RSOLVER[rs]
    assert R.akref_hydro_fluxes(C.byref(pk), rc, sc, akref.ptr(w0), *[akref.ptr(x) for x in f], 1) == 0
    fd = [_t(np.zeros_like(x)) for x in f]
    w0d = _t(w0)
    capi.check(L.akmi_hydro_fluxes(C.byref(pkd), rc, sc, capi._p(w0d), *[capi._p(x) for x in fd], 1,
                                   None), "fluxes")
    for a, b in zip(f, fd):
        assert np.isfinite(a).all() and np.array_equal(a, b.cpu().numpy())
    # hllc does not exist for the isothermal EOS
    assert L.akmi_hydro_fluxes(C.byref(pkd), rc, 2, capi._p(w0d), *[capi._p(x) for x in fd], 1, None) < 0


@pytest.mark.parametrize("rs", ["llf", "hlle", "hlld"])
@pytest.mark.parametrize("recon", ["plm", "wenoz"])
def test_task_mhd_fluxes_wild_states_isothermal(recon, rs):
    from athenak_amd import capi
    o = akref.Sim(nx1=16, nx2=12, nx3=8, mb_nx1=8, mb_nx2=12, mb_nx3=8, ng=3, nstages=2, cfl=0.3,
                  tlim=1.0, nlim=-1, is_mhd=1, recon="plm", rsolver="llf", gamma=1.4, is_ideal=0,
                  iso_cs=0.7, pgen="shock_tube", shock_dir=1, xshock=0.0,
                  wl=[1, 0, 0, 0, 1, 0.5, 1, 0], wr=[0.125, 0, 0, 0, 0.1, 0.5, -1, 0],
                  bcs=["outflow"]*6, dfloor=1e-3)
    o.initialize()
    L, R = capi.lib(), akref.lib()
    pk = o.pack()
    dxd = _t(o.array("dx"))
    pkd = capi.Pack.from_buffer_copy(bytes(pk))
    pkd.dx = dxd.data_ptr()
    rng = np.random.default_rng(77)
    n3, n2, n1 = o.dims()
    h = {"w0": _wild_states((o.nmb, 5, n3, n2, n1), rng, True)[:, :4].copy(),
         "bcc0": rng.normal(0, 2.0, size=(o.nmb, 3, n3, n2, n1)),
         "b0x1f": rng.normal(0, 2.0, size=(o.nmb, n3, n2, n1 + 1)),
         "b0x2f": rng.normal(0, 2.0, size=(o.nmb, n3, n2 + 1, n1)),
         "b0x3f": rng.normal(0, 2.0, size=(o.nmb, n3 + 1, n2, n1))}
    h["b0x1f"][:, :, :, ::3] = 0.0
    names_in = ["w0", "bcc0", "b0x1f", "b0x2f", "b0x3f"]
    names_out = ["flx1", "flx2", "flx3", "e3x1", "e2x1", "e1x2", "e3x2", "e2x3", "e1x3"]
    for k in names_out:
        h[k] = np.zeros_like(o.array(k))
    dv = {k: _t(v) for k, v in h.items()}
    rc, sc = akref.RECON[recon], akref.RSOLVER[rs]
    assert R.akref_mhd_fluxes(C.byref(pk), rc, sc, *[akref.ptr(h[k]) for k in names_in + names_out]) == 0
    capi.check(L.akmi_mhd_fluxes(C.byref(pkd), rc, sc, *[capi._p(dv[k]) for k in names_in + names_out],
                                 None), "mhd_fluxes")
    for k in names_out:
        assert np.isfinite(h[k]).all(), k
        assert np.array_equal(h[k], dv[k].cpu().numpy()), k


# ---- passive scalars (fused stage + task-granular kernels) -----------------------------------------------
@pytest.mark.parametrize("case", [
    ("sod", 32, 3, 16, 4, dict(cfl=0.3, recon="plm", rsolver="hllc")),
    ("sod", 64, 1, 32, 8, dict(cfl=0.3, ng=3, recon="wenoz", rsolver="roe")),
    ("orszag_tang", 24, 3, 12, 3, dict(cfl=0.3, recon="plm", rsolver="hlld")),
    ("orszag_tang", 32, 2, 16, 4, dict(cfl=0.3, ng=3, recon="ppm4", rsolver="hlle")),
    ("linear_wave_mhd", 32, 1, 16, 6, dict(ng=3, recon="plm", rsolver="llf",
                                            extra=["problem/along_x1=true", "mhd/eos=isothermal"])),
], ids=lambda c: "%s-%d^%d-%s" % (c[0], c[1], c[2], c[5]["rsolver"]))
@pytest.mark.parametrize("native", [False, True], ids=["py", "cpp"])
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "split"])
def test_passive_scalars(case, native, fused):
    """two passive scalars (hydro_fluxes.cpp:135-147, ideal_hyd.cpp:94-101): s0 == 1, s1 = a
    profile.  Bit-identical to the oracle, and the mass density of scalar 0 stays bit-identical to
    the density itself (its flux is the mass flux times exactly 1)."""
    import torch
    problem, n, dims, mb, cycles, kw = case
    blk = "hydro" if problem in ("linear_wave_hydro", "sod") else "mhd"
    kw = dict(kw)
    kw["extra"] = list(kw.get("extra", [])) + ["%s/nscalars=2" % blk]
    # fused: the scalars ride along with the fused stage kernels (k_scalar_update; ideal gas only, the
    # isothermal case stays on the task-granular kernels either way)
    sim, osim, is_mhd = pu.make_pair(problem, n, dims, mb, native=native, fused=fused, **kw)
    u = osim.array("u0")
    nf = u.shape[1] - 2
    prof = 0.5 + 0.25*np.sin(np.arange(u[:, 0].size, dtype=np.float64)*0.37).reshape(u[:, 0].shape)
    u[:, nf] = u[:, 0]*1.0
    u[:, nf + 1] = u[:, 0]*prof
    osim.reinitialize()
    sim.phys.u0.copy_(torch.from_numpy(u.copy()))
    if native:
        sim.Initialize()
    else:
        sim.pdriver.Initialize(sim.pmesh, sim.pin)
    for _ in range(cycles):
        assert sim.Execute(max_cycles=1) == 1 and osim.step() == 1
    got, ref = sim.phys.u0.cpu().numpy(), osim.array("u0")
    assert sim.pmesh.time == osim.time
    assert np.array_equal(got, ref)
    assert np.array_equal(sim.phys.w0.cpu().numpy(), osim.array("w0"))
    ind = sim.pmesh.mb_indcs
    a = (slice(None), slice(ind.ks, ind.ke + 1), slice(ind.js, ind.je + 1), slice(ind.is_, ind.ie + 1))
    if kw.get("recon") in ("plm", "ppm4"):                       # scalar 0: rho*1 == rho, exactly
        assert np.array_equal(got[:, nf][a], got[:, 0][a])        # (limited slopes of a constant are 0)
    else:                                                        # WENO weights do not sum to 1 exactly
        assert np.allclose(got[:, nf][a], got[:, 0][a], rtol=1e-13, atol=0.0)
    s1 = (got[:, nf + 1]/got[:, 0])[a]
    assert s1.min() >= 0.25 - 1e-12 and s1.max() <= 0.75 + 1e-12 and s1.std() > 0.01


@pytest.mark.parametrize("native", [False, True], ids=["python-host", "cpp-host"])
@pytest.mark.parametrize("bcs", ["outflow", "reflect"])
def test_hydro_conversion_inside_the_stage_kernel_with_binding_floors(bcs, native):
    """round 6: akmi_hydro_stage_w converts the cells it finishes inside k_hydro_stage3d2 (floors, floor counters, CFL scan) and
    akmi_hydro_ghost_uw fills the ghost zones of u0 and w0 with the same gather (converting a floored ghost copy again would
    not reproduce the reference: (efloor + e_kin) - e_kin need not be efloor).  Receding streams (u = -+4 either side of the middle of a
    uniform gas) empty the middle of the tube: density and pressure fall below the floors there (dfloor 0.5, pfloor 0.39) cycle
    after cycle, in active cells and -- through outflow / reflecting boundary functions of the transverse faces -- in ghost
    cells.  u0 and w0 incl. ghosts, dt, and the three floor counters of the whole run against the oracle, both hosts."""
    extra = ["mesh/%s=%s" % (f, bcs) for f in ("ix1_bc", "ox1_bc", "ix2_bc", "ox2_bc", "ix3_bc", "ox3_bc")]
    extra += ["problem/ul=-4.0", "problem/ur=4.0", "problem/dr=1.0", "problem/pl=0.4", "problem/pr=0.4"]
    sim, osim, is_mhd = pu.make_pair("sod", 32, 3, 16, fused=True, native=native, cfl=0.3, extra=extra,
                                     params={"dfloor": 0.5, "pfloor": 0.39})
    for _ in range(8):
        assert sim.Execute(max_cycles=1) == 1 and osim.step()
    d = pu.compare_fields(pu.product_arrays(sim), pu.oracle_arrays(osim, is_mhd), is_mhd)
    assert d["bitwise_equal"], d
    assert sim.pmesh.time == osim.time and sim.pmesh.dt == osim.dt
    ph = sim.phys
    assert np.array_equal(ph.w0.cpu().numpy(), osim.array("w0"))
    assert ph.u0.cpu().numpy()[:, 0].min() == 0.5                 # the density floor binds
    ocnt = osim.array("counters")
    assert ocnt[0] > 100 and ocnt[1] > 100, ocnt
    if not native:                                                # (the C++ host keeps its counters to itself)
        assert np.array_equal(ph.counters.cpu().numpy(), ocnt), (ph.counters.cpu().numpy(), ocnt)
