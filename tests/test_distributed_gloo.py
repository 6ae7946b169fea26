"""not gpu: the N>1 path (block->rank assignment, neighbour tables, rank-packed segment
ordering, torch.distributed P2P exchange, dt all-reduce) with world_size 2 over gloo.

Kernels are stood in by the oracle (tests/cpu_backend.py) so that the product's host logic
runs on CPU tensors; the multi-rank result must be BIT-IDENTICAL to the single-process oracle
run (the reference is decomposition invariant)."""
import os
import socket
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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpu_backend
    import parity_util as pu
    from oracle import akref
    cpu_backend.install()
    from athenak_amd.main import Simulation, load_deck
    problem, n, dims, mb, cycles, kw = case
    deck, ov = pu.deck_overrides(problem, n, dims, mb, **kw)
    pin = load_deck(deck, ov)
    blk = "mhd" if pin.DoesBlockExist("mhd") else "hydro"
    pin.blocks[blk]["fused_stage"] = "false"
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
    if blk == "mhd":
        for a, b in (("x1f", "b0x1f"), ("x2f", "b0x2f"), ("x3f", "b0x3f")):
            getattr(ph.b0, a).copy_(torch.from_numpy(osim.array(b)[g0:g1].copy()))
    sim.pdriver.Initialize(sim.pmesh, pin)
    for _ in range(cycles):
        sim.Execute(max_cycles=1)
        osim.step()
    ok = np.array_equal(ph.u0.numpy(), osim.array("u0")[g0:g1])
    ok = ok and np.array_equal(ph.w0.numpy(), osim.array("w0")[g0:g1])
    if blk == "mhd":
        for a, b in (("x1f", "b0x1f"), ("x2f", "b0x2f"), ("x3f", "b0x3f")):
            ok = ok and np.array_equal(getattr(ph.b0, a).numpy(), osim.array(b)[g0:g1])
    ok = ok and (sim.pmesh.time == osim.time) and (sim.pmesh.dt == osim.dt)
    npeers = len(ph.psmr.peers) if sim.pmesh.multilevel else len(ph.pbval_u.peers)
    with open(os.path.join(outdir, "rank%d.txt" % rank), "w") as f:
        f.write("%d %d %d %d\n" % (int(ok), sim.pmesh.ncycle, pk.nmb_thispack, npeers))
    dist.barrier()
    dist.destroy_process_group()


CASES = [
    ("orszag_tang", 16, 3, 8, 3, dict(cfl=0.3)),          # 8 blocks, 4 per rank, all 26 directions
    ("orszag_tang", 16, 2, 8, 3, dict(cfl=0.3)),          # 4 blocks in 2-D
    ("sod", 64, 1, 16, 5, dict(cfl=0.3)),                 # outflow BCs + block boundaries
    ("linear_wave_mhd", 16, 3, 8, 2, dict(ng=3, recon="ppm4")),
    ("linear_wave_hydro", 24, 3, 12, 2, {}),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-%d^%d-mb%d" % (c[0], c[1], c[2], c[3]))
def test_two_ranks_match_single_process_oracle(case):
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), case, d), nprocs=world, join=True)
        for r in range(world):
            ok, ncyc, nmb, npeers = map(int, open(os.path.join(d, "rank%d.txt" % r)).read().split())
            assert ok == 1, "rank %d differs from the single-process oracle" % r
            assert ncyc == case[4] and nmb >= 1 and npeers == 1


BENCH_LAYOUTS = [
    # bench.py --gpus N: one MeshBlock per rank, the mesh 2x1x1 / 2x2x1 / 2x2x2 blocks, periodic.  In a
    # direction with two blocks the +/- neighbours are the SAME peer, in a direction with one block the rank
    # is its own neighbour; at 8 ranks every one of the 26 neighbours is off-rank
    ("orszag_tang", (16, 8, 8), 3, (8, 8, 8), 2, dict(cfl=0.3), 2, 1),
    ("orszag_tang", (16, 16, 8), 3, (8, 8, 8), 2, dict(cfl=0.3), 4, 3),
    ("orszag_tang", (16, 16, 16), 3, (8, 8, 8), 2, dict(cfl=0.3), 8, 7),
]


@pytest.mark.parametrize("case", BENCH_LAYOUTS, ids=lambda c: "%dranks" % c[6])
def test_bench_layouts_match_single_process_oracle(case):
    world, peers = case[6], case[7]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), case[:6], d), nprocs=world, join=True)
        for r in range(world):
            ok, ncyc, nmb, npeers = map(int, open(os.path.join(d, "rank%d.txt" % r)).read().split())
            assert ok == 1, "rank %d differs from the single-process oracle" % r
            assert ncyc == case[4] and nmb == 1 and npeers == peers


SMR_CASES = [
    # statically refined meshes cut across ranks: level-aware segments, restricted fluxes and edge EMFs
    # travel between ranks (bvals_smr.py _plan_ranks)
    ("linear_wave_mhd_smr", (32, 16, 16), 3, (8, 4, 4), 2, {}, 2),
    ("linear_wave_mhd_smr", (32, 16, 16), 3, (8, 4, 4), 2, {}, 3),
    ("linear_wave_hydro_smr", (32, 16, 16), 3, (8, 4, 4), 2, {}, 2),
    ("linear_wave_mhd_smr", (32, 16, 1), 2, (8, 4, 1), 3, {}, 2),
    ("blast_smr", (16, 16, 16), 3, (4, 4, 4), 1, {}, 2),               # config 5's shape: PPM4 + HLLD, ng = 4
]


@pytest.mark.parametrize("case", SMR_CASES, ids=lambda c: "%s-%s-mb%s-%dranks" % (c[0], c[1], c[3], c[6]))
def test_refined_mesh_on_several_ranks_matches_single_process_oracle(case):
    world = case[6]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), case[:6], d), nprocs=world, join=True)
        for r in range(world):
            ok, ncyc, nmb, npeers = map(int, open(os.path.join(d, "rank%d.txt" % r)).read().split())
            assert ok == 1, "rank %d differs from the single-process oracle" % r
            assert ncyc == case[4] and nmb >= 1 and npeers >= 1


def test_load_balance_and_zorder():
    """Mesh::LoadBalance (load_balance.cpp:38-88) and Z-ordered gids"""
    from athenak_amd.mesh import LoadBalance, _morton
    r, s, n = LoadBalance([1.0]*8, 8)
    assert r == list(range(8)) and n == [1]*8
    r, s, n = LoadBalance([1.0]*8, 3)
    assert sum(n) == 8 and s == [0, n[0], n[0] + n[1]] and n[0] <= n[2]   # rank 0 gets less
    order = sorted([(l1, l2, l3) for l3 in range(2) for l2 in range(2) for l1 in range(2)],
                   key=lambda l: _morton(*l))
    assert order[:4] == [(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0)]


def _cli_worker(rank, world, port, deck_text, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), AKMI_DIST_BACKEND="gloo")
    import cpu_backend
    cpu_backend.install()
    from athenak_amd.__main__ import main
    deck = os.path.join(outdir, "deck.athinput")
    if rank == 0:
        with open(deck, "w") as f:
            f.write(deck_text)
    else:
        import time
        while not os.path.exists(deck):
            time.sleep(0.05)
        time.sleep(0.2)
    assert main(["-i", deck, "-d", outdir]) == 0


def test_two_ranks_write_the_same_files():
    """tab / bin / rst / hst written by two ranks (gather on rank 0, per-rank record writes into the
    shared restart file) equal the files of a single-process run"""
    import filecmp
    import output_cases as oc
    deck = oc.OT_DECK.replace("FUSED", "false") + "<output4>\nfile_type = rst\ndcycle = 3\n"
    with tempfile.TemporaryDirectory() as d:
        one, two = os.path.join(d, "one"), os.path.join(d, "two")
        os.makedirs(one)
        os.makedirs(two)
        mp.spawn(_cli_worker, args=(1, _free_port(), deck, one), nprocs=1, join=True)
        mp.spawn(_cli_worker, args=(2, _free_port(), deck, two), nprocs=2, join=True)
        files = []
        for root, _, fs in os.walk(one):
            files += [os.path.relpath(os.path.join(root, f), one) for f in fs if not f.endswith(".athinput")]
        assert any(f.startswith("rst") for f in files) and any(f.startswith("bin") for f in files)
        for rel in sorted(files):
            a, b = os.path.join(one, rel), os.path.join(two, rel)
            assert os.path.exists(b), rel
            if rel.endswith(".hst"):
                ra = np.loadtxt(a, comments="#")
                rb = np.loadtxt(b, comments="#")
                assert np.allclose(ra, rb, rtol=1e-12, atol=1e-15), rel    # sums reduced in another order
            else:
                assert filecmp.cmp(a, b, shallow=False), rel
