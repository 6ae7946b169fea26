#!/bin/bash
# Runs the REFERENCE's own regression scripts (unmodified, from /root/reference/tst/test_suite)
# against this implementation through tests/athena_shim.py.  Build container only: needs
# /root/reference; nothing is written there (scratch tree of symlinks under /tmp).
#   tools/run_reference_suite.sh [pytest -k expression]        (AKMI_SHIM_CPU=1 is the default here)
set -e
REF=/root/reference
REPO=$(cd "$(dirname "$0")/.." && pwd)
S=${AKMI_SUITE_DIR:-/tmp/aktest}
rm -rf $S && mkdir -p $S/tst/build/src
ln -s $REF/vis $S/vis
ln -s $REF/tst/test_suite $S/tst/test_suite
ln -s $REF/tst/inputs $S/tst/inputs
ln -s ../../inputs $S/tst/build/src/inputs
cat > $S/tst/build/src/athena <<SH
#!/bin/bash
exec python $REPO/tests/athena_shim.py "\$@"
SH
chmod +x $S/tst/build/src/athena
export AKMI_SHIM_CPU=${AKMI_SHIM_CPU:-1} PYTHONDONTWRITEBYTECODE=1
cd $S/tst
python - "$@" <<'PY'
import os, sys, pytest
sys.path.insert(0, os.getcwd())
import test_suite.testutils  # noqa: F401  (resolves ../vis/python relative to tst/)
tests = [os.path.abspath("test_suite/nr/test_nr_%s_cpu.py" % t) for t in ("lwave1d", "isolwave1d", "sod", "rj2a", "cpaw_amr")]
# kinematic Gaussian-pulse diffusion regressions (viscosity 1-D, conduction 1-D and 2-D)
tests += [os.path.abspath("test_suite/diffusion/test_diffusion_%s_cpu.py" % t) for t in ("visc", "conduct", "resist", "ambipolar_linwave")]
# AKMI_SUITE_PLANES=1: also the scripts the reference runs on its GPU CI machine (xy/yz/zx planes embedded in 3-D,
# 3-D ambipolar wave); they drive the same executable, +30 min with the CPU backend
if os.environ.get("AKMI_SUITE_PLANES"):
    tests += [os.path.abspath("test_suite/diffusion/test_diffusion_%s_gpu.py" % t) for t in ("visc", "conduct", "resist", "ambipolar_linwave")]
os.chdir("build/src")
args = tests + ["-p", "no:cacheprovider", "-q"] + ([] if os.environ.get("AKMI_SUITE_KEEP_GOING") else ["-x"])
if len(sys.argv) > 1:
    args += ["-k", sys.argv[1]]
sys.exit(pytest.main(args))
PY
