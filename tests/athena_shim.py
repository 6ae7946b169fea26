"""TEST INFRASTRUCTURE: stands in for the reference's `athena` executable so that the reference's
own regression scripts (tst/test_suite/nr/*.py, which run `./athena -i <deck> block/name=value ...`
and read tab/-errs.dat files) can drive this implementation unchanged.

    python tests/athena_shim.py -i deck [overrides]            # HIP path (needs a GPU)
    AKMI_SHIM_CPU=1 python tests/athena_shim.py -i deck ...     # host logic + CPU oracle kernels

The second form is what tools/run_reference_suite.sh uses in the GPU-less build container; it
exercises the deck parser, Mesh/TaskList/Driver, problem generators, Outputs and the CLI against
the reference's test harness (the kernels are then the oracle's, which the reference's thresholds
already pin in tests/test_oracle_pins.py).  The task-granular chain is forced there because the
oracle has no fused-stage twin."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    argv = sys.argv[1:]
    if os.environ.get("AKMI_SHIM_CPU", "0") == "1":
        import cpu_backend
        cpu_backend.install()
        from athenak_amd import hydro
        orig = hydro.FluidBase._setup

        def setup(self, ppack, pin, blk, device):
            pin.blocks[blk]["fused_stage"] = "false"
            return orig(self, ppack, pin, blk, device)
        hydro.FluidBase._setup = setup
    from athenak_amd.__main__ import main as run
    return run(argv)


if __name__ == "__main__":
    sys.exit(main())
