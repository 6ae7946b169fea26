"""The C entry points of the C++ host (csrc/akmi_host*.cpp: akmi_sim_*, akmi_comm_*, akmi_host_exchange_plan) return
AKMI_FAIL / NULL with the message in akmi_last_error() for RUN-TIME failures -- a device allocation, a HIP call, anything the
C++ runtime throws -- instead of letting an exception cross `extern "C"` and abort the caller (include/akmi.h:24-27; the
TaskStatus::fail of src/tasklist/task_list.hpp:30).  Errors of the input deck keep the reference's "### FATAL ERROR" + exit
(src/eos/eos.cpp:37-39).  Every case runs in a process of its own: the point is that the PROCESS survives."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROLOGUE = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
from athenak_amd import capi
from athenak_amd.main import load_deck
L = capi.lib()
def deck(n, extra=()):
    ov = ["time/nlim=2"]
    for q in (1, 2, 3):
        ov += ["mesh/nx%%d=%%d" %% (q, n), "meshblock/nx%%d=%%d" %% (q, n)]
    return load_deck("orszag_tang.athinput", ov + list(extra)).Dump().encode()
def create(text):
    h = L.akmi_sim_create(text, None)
    return h, L.akmi_last_error().decode()
""" % ROOT


def _run(body, env=None):
    r = subprocess.run([sys.executable, "-c", PROLOGUE + body], env=dict(os.environ, **(env or {})), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, "the process died (rc %d)\n%s\n%s" % (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    return r.stdout


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="the machine has a device")
def test_create_without_a_device_is_an_error_return():
    out = _run(r"""
h, msg = create(deck(16))
assert not h and "akmi_sim_create" in msg and "hipMalloc" in msg, (h, msg)
print("alive:", msg)
""")
    assert "alive" in out


@pytest.mark.gpu
def test_injected_allocation_failure_is_an_error_return():
    """the 7th device allocation of the process fails (the Mesh and half of the physics exist by then)"""
    out = _run(r"""
h, msg = create(deck(16))
assert not h and "akmi_sim_create" in msg and "injected" in msg, (h, msg)
h2, msg2 = create(deck(16))                   # and again: nothing is left in a state that faults
assert not h2 and "injected" in msg2
print("alive:", msg)
""", env={"AKMI_FAIL_ALLOC_AFTER": "7"})
    assert "alive" in out


@pytest.mark.gpu
def test_failures_return_and_the_library_keeps_working():
    out = _run(r"""
import torch
torch.cuda.set_device(0)
# a MeshBlock no device holds: 6144^3 cells x 5 variables x 8 bytes = 9.3 TB for u0 alone
h, msg = create(deck(6144))
assert not h and "hipMalloc" in msg, (h, msg)
print("too large:", msg)
# an exception of the C++ runtime: std::stol on a digit string beyond long (std::out_of_range)
h, msg = create(deck(16, ["mhd/fused_stage=99999999999999999999999999"]))
assert not h and "akmi_sim_create" in msg, (h, msg)
print("runtime exception:", msg)
# execute before initialize
h, msg = create(deck(16))
assert h, msg
h = C.c_void_p(h)
assert L.akmi_sim_execute(h, 1) < 0 and "initialize" in L.akmi_last_error().decode()
L.akmi_sim_destroy(h)
# ... and an ordinary run still works in the same process
sys.path.insert(0, %r)
import parity_util as pu
r = pu.compare_run("orszag_tang", 16, 3, 16, cycles=2, native=True, fused=True)
assert r["bitwise_equal"], r
print("alive")
""" % os.path.join(ROOT, "tests"))
    assert "alive" in out and "too large" in out and "runtime exception" in out
