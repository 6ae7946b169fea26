"""gpu: the RCCL point-to-point transport of the C++ host, executed on the one GPU of the test box.

ncclSend / ncclRecv to one's own rank inside ncclGroupStart/End is legal on a one-rank communicator.  With
AKMI_SELF_EXCHANGE=1 the C++ host treats the neighbours on its own rank as neighbours on a peer rank whose
number is its own (csrc/akmi_host_comm.cpp BuildPlan): every ghost zone then travels
    pack kernel -> send buffer -> grouped ncclRecv/ncclSend on the communicator's stream -> receive buffer ->
    unpack kernel,
ordered against the compute stream by the ready/done events of Comm::Post/Wait, and dt goes through
ncclAllReduce(ncclMin) -- the code path of `bench.py --gpus N`, which the reference's MPI calls map to
(src/bvals/bvals_cc.cpp:247-258 Isend/Irecv, src/mesh/mesh.cpp:634-637 Allreduce).  The run must be
BIT-IDENTICAL to the same deck with the same-rank gather (and therefore to the oracle, which
tests/test_gpu_parity.py pins the gather path to).
"""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu


def _run(self_exchange, case, fused, out):
    import ctypes as C
    if self_exchange:
        os.environ["AKMI_SELF_EXCHANGE"] = "1"
    torch.cuda.set_device(0)
    import parity_util as pu
    from athenak_amd import capi, native
    from athenak_amd.main import load_deck
    L = capi.lib()
    problem, n, dims, mb, cycles, kw = case
    deck, ov = pu.deck_overrides(problem, n, dims, mb, **kw)
    pin = load_deck(deck, ov)
    blk = "mhd" if pin.DoesBlockExist("mhd") else "hydro"
    pin.blocks[blk]["fused_stage"] = "true" if fused else "false"
    pin.blocks[blk]["small_pack_tasks"] = "false"         # small fixture: keep the fused phase-split path under test
    if self_exchange:
        idb = C.create_string_buffer(128)
        capi.check(L.akmi_comm_unique_id(idb), "comm_unique_id")
        capi.check(L.akmi_comm_init_rccl(0, 1, idb.raw), "comm_init_rccl")
        v = (C.c_double*2)(3.5, -1.25)                      # ncclAllReduce(min) on a stream, directly
        capi.check(L.akmi_comm_allreduce_min(v, 2, capi._stream()), "comm_allreduce_min")
        assert (v[0], v[1]) == (3.5, -1.25)
    sim = native.NativeSimulation(pin)
    done = sim.Execute(max_cycles=cycles)
    torch.cuda.synchronize()
    ph = sim.phys
    arrs = {"u0": ph.u0.cpu().numpy(), "w0": ph.w0.cpu().numpy()}
    if blk == "mhd":
        arrs.update(b1=ph.b0.x1f.cpu().numpy(), b2=ph.b0.x2f.cpu().numpy(), b3=ph.b0.x3f.cpu().numpy(),
                    bcc=ph.bcc0.cpu().numpy())
    np.savez(out, done=done, time=sim.time, dt=sim.dt, **arrs)
    sim.close()
    if self_exchange:
        native.finalize_comm()


CASES = [
    ("orszag_tang", 32, 3, 32, 3, dict(cfl=0.3)),                 # ONE periodic block: its own +-x/+-y/+-z peer, 26 directions
    ("orszag_tang", 32, 3, 16, 2, dict(cfl=0.3)),                 # 8 blocks, every neighbour through the transport
    ("sod", 96, 1, 32, 4, dict(cfl=0.3)),                         # outflow faces next to exchanged ones
    ("linear_wave_mhd", 24, 3, 12, 2, dict(ng=3, recon="ppm4", integrator="rk3")),
]


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "split"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-%s^%d-mb%s" % (c[0], c[1], c[2], c[3]))
def test_rccl_send_recv_to_self_equals_same_rank_gather(case, fused):
    with tempfile.TemporaryDirectory() as d:
        res = []
        for self_exchange in (False, True):
            out = os.path.join(d, "r%d.npz" % int(self_exchange))
            ctx = mp.get_context("spawn")
            p = ctx.Process(target=_run, args=(self_exchange, case, fused, out))
            p.start()
            p.join(600)
            assert p.exitcode == 0, "child (self_exchange=%s) ended with %s" % (self_exchange, p.exitcode)
            res.append(dict(np.load(out)))
        a, b = res
        assert int(a["done"]) == int(b["done"]) == case[4]
        assert float(a["time"]) == float(b["time"]) and float(a["dt"]) == float(b["dt"])
        for k in a:
            assert np.array_equal(a[k], b[k]), "%s differs between the RCCL self-exchange and the gather" % k
