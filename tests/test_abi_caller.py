"""include/akmi.h from a compiled caller: tests/abi_caller.c is built as C99 (gcc) and as C++17 (g++), linked
against athenak_amd/lib/libakmi.so and the HIP runtime's C API -- no Python, no torch in the process -- and runs a
Hydro task chain (Fluxes -> RKUpdate -> BCs -> ConToPrim -> NewTimeStep) plus the fused stage entry on one
MeshBlock.  not gpu: both builds compile and link against every symbol they use.  gpu: both executables run and
their own checks pass (fixed point, conservation, fused == task chain bit for bit)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "abi_caller.c")
LIBDIR = os.path.join(ROOT, "athenak_amd", "lib")
ROCM = "/opt/rocm"

BUILDS = {"c99": ["gcc", "-std=c99", "-Wall", "-Werror"],
          "cxx17": ["g++", "-std=c++17", "-x", "c++", "-Wall", "-Werror"]}


def _build(kind, outdir):
    from athenak_amd import capi
    capi.lib()                                   # fails loudly if the HIP library has not been built
    exe = os.path.join(outdir, "abi_caller_" + kind)
    cmd = BUILDS[kind] + ["-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", ROCM + "/include", SRC,
                          "-L", LIBDIR, "-lakmi", "-L", ROCM + "/lib", "-lamdhip64", "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None or not os.path.isdir(ROCM + "/include/hip"), reason="needs gcc and the ROCm headers")
@pytest.mark.parametrize("kind", sorted(BUILDS))
def test_caller_compiles_and_links(kind, tmp_path):
    exe = _build(kind, str(tmp_path))
    assert os.path.getsize(exe) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("kind", sorted(BUILDS))
def test_caller_runs_the_task_chain(kind, tmp_path):
    exe = _build(kind, str(tmp_path))
    env = dict(os.environ, LD_LIBRARY_PATH=LIBDIR + ":" + ROCM + "/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout
