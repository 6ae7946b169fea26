"""gpu: LONG whole runs against the oracle, bit for bit -- hundreds of cycles instead of a handful, so that the
branches a developing flow reaches (every HLLD / HLLC fan region, degenerate rotational waves, limiter clips of PPM4,
floors, the CFL scan following a shock) are compared where they occur, not only at the smooth start.  Sizes are
chosen so that the oracle needs a few seconds per case on the granted cores.  Through the fused stage and the
task-granular chain, the Python and the C++ host, one block and several, uniform and refined meshes."""
import pytest

pytestmark = pytest.mark.gpu

import parity_util as pu  # noqa: E402

LONG = [
    # problem, n, dims, mb, cycles, kwargs
    ("orszag_tang", 64, 3, 64, 140, dict(cfl=0.3, fused=True)),                 # C3's deck well past shock formation
    ("orszag_tang", 48, 3, 24, 100, dict(cfl=0.3, fused=True, native=True)),    # eight blocks, C++ host
    ("orszag_tang", 48, 3, 24, 70, dict(cfl=0.3, fused=False)),                 # task-granular chain
    ("orszag_tang", 128, 2, 64, 350, dict(cfl=0.3)),                            # 2-D: current sheets by cycle ~300
    ("sod", 64, 3, 32, 90, dict(cfl=0.3, fused=True)),                          # C2's deck, shock crosses blocks
    ("blast", 40, 3, 20, 70, dict(fused=True)),                                # PPM4 + HLLD, ng = 4, strong blast
    ("blast", 40, 3, 20, 45, dict(fused=False, native=True, integrator="rk3")),
    ("blast", 64, 2, 32, 300, dict()),
]


@pytest.mark.parametrize("case", LONG, ids=lambda c: "%s-%d^%d-mb%d-%dcyc%s%s" % (
    c[0], c[1], c[2], c[3], c[4], "-task" if c[5].get("fused") is False else "", "-cpp" if c[5].get("native") else ""))
def test_long_run_bitwise(case):
    problem, n, dims, mb, cycles, kw = case
    r = pu.compare_run(problem, n, dims, mb, cycles, **kw)
    assert r["cycles"] >= min(cycles, 45), r["cycles"]          # (a deck's own tlim may end the run before `cycles`)
    assert r["time"][0] == r["time"][1] and r["dt"][0] == r["dt"][1], (r["time"], r["dt"])
    assert r["bitwise_equal"], r["diffs"]


def test_long_run_refined_mesh_bitwise():
    """BASELINE config 5's mesh at fixture size (3-D blast, one refined region, PPM4 + HLLD + CT, ng = 4): 30 cycles,
    the blast wave reaches the fine/coarse boundary; both hosts"""
    for native in (False, True):
        r = pu.compare_run("blast_smr", (32, 32, 32), 3, (8, 8, 8), cycles=30, native=native)
        assert r["cycles"] == 30 and r["dt"][0] == r["dt"][1], (native, r["cycles"], r["dt"])
        assert r["bitwise_equal"], (native, r["diffs"])


def test_refined_mesh_production_block_size_bitwise():
    """config 5's mesh with the MeshBlock size of its production run (32^3, ng = 4) on a 128^3 root grid: 120 MeshBlocks on
    two levels, 3 cycles through the C++ host.  (The same check on 960 MeshBlocks of 16^3 -- the copy lists of the
    exchange at the block count of the production mesh -- passes through both hosts but needs 85 s per host for the
    oracle's set-up: tools/c5_mid_check.py.)"""
    r = pu.compare_run("blast_smr", (128, 128, 128), 3, (32, 32, 32), cycles=3, native=True)
    assert r["cycles"] == 3 and r["dt"][0] == r["dt"][1], (r["cycles"], r["dt"])
    assert r["bitwise_equal"], r["diffs"]
