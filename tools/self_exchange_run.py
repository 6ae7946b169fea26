#!/usr/bin/env python
"""GPU box: C++ host, orszag_tang 256^3 one block, N cycles; AKMI_SELF_EXCHANGE=1 sends every ghost zone through
pack -> ncclSend/ncclRecv (one-rank communicator, comm stream) -> unpack, as a rank with 26 off-rank neighbours would.
Prints ms per cycle and the profiled stage-group time.  usage: self_exchange_run.py [cycles] [nx]"""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from athenak_amd import capi, native  # noqa: E402
from athenak_amd.main import load_deck  # noqa: E402

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 256
torch.cuda.set_device(0)
L = capi.lib()
ov = ["time/cfl_number=0.3", "time/nlim=-1", "time/tlim=1.0e9"]
for q in (1, 2, 3):
    ov += ["mesh/nx%d=%d" % (q, nx), "meshblock/nx%d=%d" % (q, nx)]
pin = load_deck("orszag_tang.athinput", ov)
pin.blocks["mhd"]["fused_stage"] = "true"
if os.environ.get("AKMI_SELF_EXCHANGE", "0") == "1":
    idb = C.create_string_buffer(128)
    capi.check(L.akmi_comm_unique_id(idb), "comm_unique_id")
    capi.check(L.akmi_comm_init_rccl(0, 1, idb.raw), "comm_init_rccl")
sim = native.NativeSimulation(pin)
sim.Execute(max_cycles=3)
capi.check(L.akmi_sim_profile(sim.h, 1), "sim_profile")
torch.cuda.synchronize()
t0 = time.perf_counter()
n = sim.Execute(max_cycles=cycles)
torch.cuda.synchronize()
el = time.perf_counter() - t0
ms, calls = C.c_double(0.0), C.c_longlong(0)
capi.check(L.akmi_sim_profile_read(sim.h, C.byref(ms), C.byref(calls)), "sim_profile_read")
print("self_exchange=%s nx=%d cycles=%d  %.4f ms/cycle  %.1f Mcell-updates/s  stage-group %.4f ms/stage (%d calls)  t=%r dt=%r" % (
    os.environ.get("AKMI_SELF_EXCHANGE", "0"), nx, n, el/n*1e3, nx**3*n/el/1e6, ms.value/max(calls.value, 1)*
    (calls.value/(2.0*n) if calls.value else 0), calls.value, sim.time, sim.dt))
