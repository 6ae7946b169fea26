"""CPU: the C++ host's rank layout and exchange plan (akmi_host_exchange_plan: Mesh::LoadBalance, the
27-direction neighbour table with ranks, peers, remote slots, per-peer message slices, pack/unpack
tables -- csrc/akmi_host_comm.cpp) against the Python host's plan (athenak_amd/bvals.py), which the
2-rank gloo and HIP tests pin bit for bit against the single-process oracle.  Host code only: runs
without a GPU."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity_util as pu  # noqa: E402
from athenak_amd import capi  # noqa: E402
from athenak_amd.bvals import HipBvalsKernels, MeshBoundaryValues  # noqa: E402
from athenak_amd.main import load_deck  # noqa: E402
from athenak_amd.mesh import Mesh  # noqa: E402

CASES = [
    # problem, n, dims, mb, nranks
    ("orszag_tang", 32, 3, 16, 2),
    ("orszag_tang", 32, 3, 16, 3),                 # uneven: 3 + 3 + 2 blocks
    ("orszag_tang", 32, 3, 16, 8),                 # one block per rank: every neighbour is remote
    ("orszag_tang", 32, 3, (32, 32, 32), 1),       # one rank: every direction wraps onto the block itself
    ("orszag_tang", 32, 3, (16, 32, 32), 2),       # the bench layouts (bench.py block_grid) at 2, 4, 8 GPUs:
    ("orszag_tang", 32, 3, (16, 16, 32), 4),       # one block per rank; in the directions with one block the
    ("orszag_tang", 32, 3, (16, 16, 16), 8),       # rank is its own +/- neighbour, in the others both are the same peer
    ("orszag_tang", 64, 3, 16, 5),
    ("sod", 128, 1, 32, 2),                        # outflow: faces without a neighbour
    ("sod", 128, 1, 32, 4),
    ("blast", 32, 2, 16, 2),
    ("blast", 64, 2, 16, 7),
    ("linear_wave_hydro", 24, 3, 12, 4),
]


def _native_plan(deck, rank, nranks, nvar, fc):
    L = capi.lib()
    buf = (C.c_longlong*1)()
    n = L.akmi_host_exchange_plan(deck.encode(), rank, nranks, nvar, fc, buf, 0)
    assert n > 0
    buf = (C.c_longlong*n)()
    assert L.akmi_host_exchange_plan(deck.encode(), rank, nranks, nvar, fc, buf, n) == n
    v = list(buf)
    it = iter(v)
    npeer = next(it)
    peers = [next(it) for _ in range(npeer)]
    slices = [[next(it) for _ in range(4)] for _ in range(npeer)]
    nmb = next(it)
    tab = np.array([next(it) for _ in range(27*nmb)]).reshape(nmb, 27)
    nsend = next(it)
    send_tab = np.array([next(it) for _ in range(2*nsend)]).reshape(nsend, 2)
    send_off = [next(it) for _ in range(nsend)]
    nseg = next(it)
    seg_off = [next(it) for _ in range(nseg)]
    assert next(it, None) is None
    return peers, slices, tab, send_tab, send_off, seg_off


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-%s^%d-mb%s-%dranks" % c)
def test_cpp_plan_equals_python_plan(case):
    problem, n, dims, mb, nranks = case
    deck, ov = pu.deck_overrides(problem, n, dims, mb)
    pin = load_deck(deck, ov)
    is_mhd = pin.DoesBlockExist("mhd")
    nvar = 5
    text = pin.Dump()
    owned = []
    for rank in range(nranks):
        pm = Mesh(pin, my_rank=rank, nranks=nranks)
        pk = pm.pmb_pack
        bv = MeshBoundaryValues(pk, kernels=HipBvalsKernels(), device="cpu")
        ind = pm.mb_indcs
        pack_c = capi.Pack(nmb=pk.nmb_thispack, nvar=nvar, nx1=ind.nx1, nx2=ind.nx2, nx3=ind.nx3, ng=ind.ng)
        bv.set_pack(pack_c, nvar)
        owned.append((pk.gids, pk.nmb_thispack))
        for fc in ((0, 1) if is_mhd else (0,)):
            ch = bv.fc if fc else bv.cc
            peers, slices, tab, send_tab, send_off, seg_off = _native_plan(text, rank, nranks, nvar, fc)
            assert peers == bv.peers
            assert np.array_equal(tab, bv.nghbr_host)
            for r, s in zip(peers, slices):
                assert tuple(s[:2]) == tuple(ch.send_slices[r]) and tuple(s[2:]) == tuple(ch.recv_slices[r])
            if peers:
                assert np.array_equal(send_tab, ch.send_tab.numpy())
                assert send_off == ch.send_off.tolist() and seg_off == ch.seg_off.tolist()
            else:
                assert len(send_off) == 0 and len(seg_off) == 0
    # the ranks tile the block list
    assert owned[0][0] == 0 and sum(nb for _, nb in owned) == pm.nmb_total
    assert all(owned[q][0] + owned[q][1] == owned[q + 1][0] for q in range(nranks - 1))


def test_plan_messages_pair_up():
    """what rank a sends to rank b is what rank b expects from rank a, segment by segment"""
    deck, ov = pu.deck_overrides("orszag_tang", 32, 3, 16)
    text = load_deck(deck, ov).Dump()
    nranks = 4
    plans = [_native_plan(text, r, nranks, 5, 0) for r in range(nranks)]
    for a in range(nranks):
        pa, sa = plans[a][0], plans[a][1]
        for r, s in zip(pa, sa):
            pb, sb = plans[r][0], plans[r][1]
            t = sb[pb.index(a)]
            assert s[1] - s[0] == t[3] - t[2] and s[3] - s[2] == t[1] - t[0]


def test_more_ranks_than_blocks_is_refused():
    import subprocess
    deck, ov = pu.deck_overrides("sod", 64, 1, 32)
    text = load_deck(deck, ov).Dump()
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from athenak_amd import capi; "
            "capi.lib().akmi_host_exchange_plan(%r.encode(), 0, 3, 5, 0, None, 0)" % (pu.ROOT, text))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "no MeshBlock" in r.stderr
