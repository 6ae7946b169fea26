"""python -m athenak_amd (-i <deck> | -r <restart file>) [-d <run_dir>] [block/name=value ...]

Command-line entry with the argument conventions of the reference's executable
(src/main.cpp:61-420: -i input file, -d run directory, trailing block/name=value overrides), so
that scripts written around `athena -i ...` (e.g. the reference's regression-test driver) can run
this implementation: reads the deck, builds Mesh/physics/ProblemGenerator/Outputs/Driver, runs
Initialize -> Execute -> Finalize and writes tab/hst/bin/-errs.dat files in the run directory.
"""
import os
import sys
import time


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    deck, rundir, overrides, rstfile = None, None, [], None
    i = 0
    while i < len(argv):
        a = argv[i]
        if a == "-i":
            deck = argv[i + 1]; i += 2
        elif a == "-r":
            rstfile = argv[i + 1]; i += 2
        elif a == "-d":
            rundir = argv[i + 1]; i += 2
        elif a in ("-h", "--help"):
            print(__doc__)
            return 0
        elif a.startswith("-"):
            sys.stderr.write("### FATAL ERROR unknown option %s (supported: -i -r -d -h)\n" % a)
            return 1
        else:
            overrides.append(a); i += 1
    if deck is None and rstfile is None:
        sys.stderr.write("### FATAL ERROR Either an input or restart file must be specified: "
                         "-i <deck> or -r <file>\n")
        return 1
    from .main import Simulation, load_deck, load_restart
    from .outputs import Outputs
    if deck is not None:
        deck = os.path.abspath(deck) if os.path.exists(deck) else deck
    if rstfile is not None:
        rstfile = os.path.abspath(rstfile)
    if rundir:
        os.makedirs(rundir, exist_ok=True)
        os.chdir(rundir)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch
        import torch.distributed as dist
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(os.environ.get("AKMI_DIST_BACKEND", "nccl"))
    if rstfile is not None:
        sim = load_restart(rstfile, overrides, my_rank=rank, nranks=world, initialize=False)
        pin = sim.pin
    else:
        pin = load_deck(deck, overrides)
        sim = Simulation(pin, my_rank=rank, nranks=world, initialize=False)
    pm, drv = sim.pmesh, sim.pdriver
    pout = Outputs(pin, pm)
    drv.Initialize(pm, pin, pout, res_flag=rstfile is not None)
    t0 = time.time()
    drv.Execute(pm, pin)
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    el = time.time() - t0
    drv.Finalize(pm, pin, pout)
    if rank == 0:
        zc = drv.nmb_updated_*pm.NumberOfMeshBlockCells()
        print("\ncycle=%d time=%.14e dt=%.14e" % (pm.ncycle, pm.time, pm.dt))
        print("Terminating on %s" % ("time limit" if pm.time >= drv.tlim else "cycle limit"))
        print("time=%e cycle=%d\ntlim=%e nlim=%d" % (pm.time, pm.ncycle, drv.tlim, drv.nlim))
        print("cpu time used  = %e\nzone-cycles/cpu_second = %e" % (el, zc/max(el, 1e-30)))
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
