"""Regenerates tests/golden/outputs/*: the files the output writers (athenak_amd/outputs.py,
pgen.WriteErrorsFile) produce for the cases of tests/output_cases.py, with the product's host
logic driven by the CPU oracle (tests/cpu_backend.py), so the fixtures do not depend on a GPU.
When /root/reference is present every file is parsed with the REFERENCE's own readers
(vis/python/athena_read.py, bin_convert.py) before it is committed -- that is what pins the
formats; see tests/test_outputs_formats.py for the checks.

    python tests/golden/make_output_fixtures.py
"""
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import cpu_backend
    import output_cases as oc
    cpu_backend.install()
    dst = os.path.join(HERE, "outputs")
    shutil.rmtree(dst, ignore_errors=True)
    os.makedirs(dst)
    for name in ("sod", "ot", "lwave_hydro", "lwave_mhd"):
        with tempfile.TemporaryDirectory() as d:
            files = oc.run_case(name, d, fused=False)
            for rel in files:
                tgt = os.path.join(dst, name, rel)
                os.makedirs(os.path.dirname(tgt), exist_ok=True)
                shutil.copy(os.path.join(d, rel), tgt)
                print(name, rel, os.path.getsize(tgt))
    cpu_backend.uninstall()


if __name__ == "__main__":
    main()
