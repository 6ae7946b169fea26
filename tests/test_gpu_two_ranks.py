"""gpu: the multi-rank path with the HIP kernels.  Two processes (one MeshBlockPack each, as
on a 2-GPU node) share cuda:0 because the test box has one GPU; RCCL refuses two ranks on one
device, so the messages travel over gloo through pinned host buffers (bvals.py `_staged`),
everything else -- block->rank assignment, HIP pack/unpack of off-rank segments, same-rank
gather, fused stage kernels, dt all-reduce -- is the production path.  Each rank's result must
be BIT-IDENTICAL to the single-process oracle run (decomposition invariance)."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from test_distributed_gloo import _free_port  # noqa: E402

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, case, fused, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import parity_util as pu
    from oracle import akref
    from athenak_amd.main import Simulation, load_deck
    problem, n, dims, mb, cycles, kw = case
    deck, ov = pu.deck_overrides(problem, n, dims, mb, **kw)
    pin = load_deck(deck, ov)
    blk = "mhd" if pin.DoesBlockExist("mhd") else "hydro"
    pin.blocks[blk]["fused_stage"] = "true" if fused else "false"
    pin.blocks[blk]["small_pack_tasks"] = "false"      # small fixture: the path the test asks for
    sim = Simulation(pin, my_rank=rank, nranks=world, initialize=False)
    okw = pu.oracle_kwargs(pin)
    if sim.pmesh.multilevel:           # the single-process oracle takes the tree of the whole mesh
        from athenak_amd.mesh import Mesh
        okw.update(pu.smr_tables(Mesh(pin)))
    osim = akref.Sim(**okw)
    osim.initialize()
    pk = sim.pmesh.pmb_pack
    g0, g1 = pk.gids, pk.gide + 1
    ph = sim.phys
    ph.u0.copy_(torch.from_numpy(osim.array("u0")[g0:g1].copy()))
    names = (("x1f", "b0x1f"), ("x2f", "b0x2f"), ("x3f", "b0x3f")) if blk == "mhd" else ()
    for a, b in names:
        getattr(ph.b0, a).copy_(torch.from_numpy(osim.array(b)[g0:g1].copy()))
    sim.pdriver.Initialize(sim.pmesh, pin)
    for _ in range(cycles):
        sim.Execute(max_cycles=1)
        osim.step()
    ok = np.array_equal(ph.u0.cpu().numpy(), osim.array("u0")[g0:g1])
    ok = ok and np.array_equal(ph.w0.cpu().numpy(), osim.array("w0")[g0:g1])
    for a, b in names:
        ok = ok and np.array_equal(getattr(ph.b0, a).cpu().numpy(), osim.array(b)[g0:g1])
    ok = ok and (sim.pmesh.time == osim.time) and (sim.pmesh.dt == osim.dt)
    with open(os.path.join(outdir, "rank%d.txt" % rank), "w") as f:
        f.write("%d %d %d %d\n" % (int(ok), sim.pmesh.ncycle, pk.nmb_thispack,
                                   len(ph.psmr.peers) if sim.pmesh.multilevel else len(ph.pbval_u.peers)))
    dist.barrier()
    dist.destroy_process_group()


CASES = [
    ("orszag_tang", 32, 3, 16, 3, dict(cfl=0.3)),         # 8 blocks, 4 per rank, all 26 directions
    ("orszag_tang", 32, 3, (16, 32, 32), 2, dict(cfl=0.3)),   # ONE block per rank (bench layout)
    ("sod", 128, 1, 32, 5, dict(cfl=0.3)),                # outflow BCs + block boundaries
    ("linear_wave_mhd", 24, 3, 12, 2, dict(ng=3, recon="ppm4", integrator="rk3")),
    ("blast", 32, 2, 16, 3, {}),
    ("linear_wave_hydro", 24, 3, 12, 2, {}),
]


def _id(c):
    return "%s-%s^%d-mb%s" % (c[0], c[1], c[2], c[3])


# every layout through the fused stage; the task-granular chain on the two 3-D MHD layouts (all 26 directions, one block
# per rank) -- the other four layouts x split were multi-process launches for kernels the single-rank suite covers
RUNS = [pytest.param(c, f, id="%s-%s^%d-mb%s-%s" % (c[0], c[1], c[2], c[3], "fused" if f else "split"))
        for c, f in [(c, True) for c in CASES] + [(c, False) for c in CASES[:2]]]


@pytest.mark.parametrize("case,fused", RUNS)
def test_two_ranks_hip_kernels_match_single_process_oracle(case, fused):
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), case, fused, d), nprocs=world, join=True)
        for r in range(world):
            ok, ncyc, nmb, npeers = map(int, open(os.path.join(d, "rank%d.txt" % r)).read().split())
            assert ok == 1, "rank %d differs from the single-process oracle" % r
            assert ncyc == case[4] and nmb >= 1 and npeers == 1


BENCH_LAYOUTS = [
    # bench.py --gpus 4 / 8: one MeshBlock per rank, 2x2x1 / 2x2x2 periodic mesh, fused stage in phases
    ("orszag_tang", (32, 32, 16), 3, (16, 16, 16), 2, dict(cfl=0.3), 4, 3),
    ("orszag_tang", (32, 32, 32), 3, (16, 16, 16), 2, dict(cfl=0.3), 8, 7),
]


@pytest.mark.parametrize("case", BENCH_LAYOUTS, ids=lambda c: "%dranks" % c[6])
def test_bench_layouts_hip_kernels(case):
    world, peers = case[6], case[7]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), case[:6], True, d), nprocs=world, join=True)
        for r in range(world):
            ok, ncyc, nmb, npeers = map(int, open(os.path.join(d, "rank%d.txt" % r)).read().split())
            assert ok == 1, "rank %d differs from the single-process oracle" % r
            assert ncyc == case[4] and nmb == 1 and npeers == peers


SMR_CASES = [
    # problem, mesh, dims, block, cycles, kwargs, ranks: refined meshes cut across ranks
    ("linear_wave_mhd_smr", (32, 16, 16), 3, (8, 4, 4), 2, {}, 2),
    ("linear_wave_mhd_smr", (32, 16, 16), 3, (8, 4, 4), 2, {}, 3),
    ("linear_wave_hydro_smr", (32, 16, 16), 3, (8, 4, 4), 2, {}, 2),
    ("linear_wave_mhd_smr", (32, 16, 1), 2, (8, 4, 1), 3, {}, 2),
    ("blast_smr", (32, 32, 32), 3, (8, 8, 8), 2, {}, 2),               # config 5's shape: PPM4 + HLLD, ng = 4
]


@pytest.mark.parametrize("case", SMR_CASES, ids=lambda c: "%s-%s-mb%s-%dranks" % (c[0], c[1], c[3], c[6]))
def test_refined_mesh_on_several_ranks_hip_kernels(case):
    """level-aware segments, restricted fluxes and edge EMFs between ranks (akmi_smr_pack_* /
    akmi_smr_unpack_* with the soff/roff tables); each rank bit-identical to the single-process oracle"""
    world = case[6]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), case[:6], False, d), nprocs=world, join=True)
        for r in range(world):
            ok, ncyc, nmb, npeers = map(int, open(os.path.join(d, "rank%d.txt" % r)).read().split())
            assert ok == 1, "rank %d differs from the single-process oracle" % r
            assert ncyc == case[4] and nmb >= 1 and npeers >= 1
