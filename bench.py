#!/usr/bin/env python
"""bench.py -- Mcell-updates/s of the MeshBlock finite-volume update on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU.  Started by a launcher (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) the process is one of the
ranks; started plain (`python3 bench.py --gpus N`, no WORLD_SIZE) it starts the N ranks itself through
torch.distributed.run on 127.0.0.1 and passes rank 0's JSON line and the exit status through.

Workload (BASELINE.json configs[2], the config the metric is quoted on): 3-D Orszag-Tang,
256^3 cells per GPU, ideal MHD, PLM + HLLD + CT, RK2, cfl 0.3, one MeshBlock (pack) per GPU;
N GPUs weak-scale the mesh to 2x1x1 / 2x2x1 / 2x2x2 blocks with halo exchange over RCCL.
A "step" is one full RK2 cycle (two stages) of every cell = one cell-update per cell
(the reference's zone-cycle, src/driver/driver.cpp:513-522).  Data are synthetic: the
closed-form Orszag-Tang initial condition, resident in HBM before the timed region.

One JSON line is printed by rank 0 (see the contract in the task description), including
  roofline     -- achieved algorithmic HBM GB/s of the dominant kernel group, from HIP events
  cpu_baseline -- the CPU oracle ("port" of the reference's split-kernel sequence) timed on a
                  bounded sample of the same workload on this box's host cores (N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# algorithmic (compulsory) HBM bytes, SURVEY.md section 8(d) / BASELINE.md section 3:
#   pass A (fluxes+EMF+update+CT): read w0 5 + bcc0 3 + b0 3 + u0 5 + u1 5 + b1 3, write u0 5 + b0 3
#   pass B (c2p[+dt]):              read u0 5 + b0 3, write w0 5 + bcc0 3
BYTES_PASS_A = {"mhd": 32*8, "hydro": 20*8}
BYTES_PASS_B = {"mhd": 16*8, "hydro": 10*8}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=55, help="timed cycles (BASELINE config 3: nlim = 55)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nx", type=int, default=256, help="cells per GPU per dimension")
    ap.add_argument("--mb", type=int, default=0,
                    help="MeshBlock size per dimension (default: --nx, one MeshBlock per GPU); "
                         "smaller blocks put (nx/mb)^3 MeshBlocks into each GPU's pack")
    ap.add_argument("--problem", default="orszag_tang", choices=["orszag_tang", "sod", "linear_wave", "linear_wave_mhd"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the side measurements of the default run (configs[1] and configs[4]'s mesh on one GPU)")
    ap.add_argument("--cpu-sample-nx", type=int, default=128)
    ap.add_argument("--recon", default=None, choices=["dc", "plm", "ppm4", "ppmx", "wenoz"],
                    help="reconstruction (default: the deck's plm); ppm4 = the numerics of BASELINE config 5")
    ap.add_argument("--ng", type=int, default=None, help="ghost cells (default 2; ppm4 needs >= 3)")
    ap.add_argument("--split", action="store_true", help="task-granular chain instead of fused stage")
    ap.add_argument("--set", action="append", default=[], metavar="block/name=value",
                    help="extra deck parameter (side measurements, e.g. mhd/nscalars=2); named in config.workload")
    ap.add_argument("--native-child", default=None, metavar="RESULT_FILE", help=argparse.SUPPRESS)
    ap.add_argument("--native", action="store_true",
                    help="drive the run from the C++ host (akmi_sim_*): Driver/TaskList in C++, halos and the dt "
                         "reduction through RCCL called directly (ncclSend/ncclRecv/ncclAllReduce); the roofline "
                         "entry is then the whole stage, not the kernel group")
    return ap.parse_args()


def block_grid(n):
    return {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}[n]


def make_pin(args, nblk):
    from athenak_amd.main import load_deck
    nx = args.nx
    mesh = [nx*b for b in nblk]
    if args.problem == "orszag_tang":
        deck, blk = "orszag_tang.athinput", "mhd"
        ov = ["time/cfl_number=0.3"]
    elif args.problem == "sod":
        deck, blk = "sod.athinput", "hydro"
        ov = ["time/cfl_number=0.3", "mesh/ix1_bc=outflow", "mesh/ox1_bc=outflow"]
    elif args.problem == "linear_wave_mhd":        # side measurements (isothermal MHD: --set mhd/eos=isothermal)
        deck, blk = "linear_wave_mhd.athinput", "mhd"
        ov = ["problem/amp=0.1"]
    else:
        deck, blk = "linear_wave_hydro.athinput", "hydro"
        ov = []
    for q in range(3):
        ov += ["mesh/nx%d=%d" % (q + 1, mesh[q]), "meshblock/nx%d=%d" % (q + 1, args.mb or nx)]
    ov += ["time/nlim=-1", "time/tlim=1.0e9"]
    if args.recon:
        ov.append("%s/reconstruct=%s" % (blk, args.recon))
    if args.ng or (args.recon in ("ppm4", "ppmx", "wenoz")):
        ov.append("mesh/nghost=%d" % (args.ng or 4))
    pin = load_deck(deck, ov)
    for kv in args.set:                    # may add parameters the deck does not have
        b, rest = kv.split("/", 1)
        name, val = rest.split("=", 1)
        pin.blocks.setdefault(b, {})[name] = val
    if args.split:
        pin.blocks[blk]["fused_stage"] = "false"
    return pin, blk


def lib_sha16():
    """first 16 hex digits of the sha256 of the HIP library this process loaded"""
    import hashlib
    from athenak_amd import capi
    return hashlib.sha256(open(capi.LIB_PATH, "rb").read()).hexdigest()[:16]


def src_sha16():
    """first 16 hex digits of the sha256 over the kernel sources and build flags: identical sources give
    identical kernels even where two builds of libakmi.so differ in their bytes (hipcc embeds temporary
    file names), which is what the traffic counters of a profiling run belong to"""
    import hashlib
    import __graft_entry__ as ge
    h = hashlib.sha256(" ".join(ge.HIPCC_FLAGS).encode())
    csrc = os.path.join(ROOT, "athenak_amd", "csrc")
    for f in sorted(os.listdir(csrc)) + [os.path.join(ROOT, "include", "akmi.h")]:
        path = f if os.path.isabs(f) else os.path.join(csrc, f)
        if path.endswith((".hip", ".hpp", ".cpp", ".h")):
            h.update(os.path.basename(path).encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(args, blk):
    """the oracle (port of the reference's split-kernel CPU sequence) on all host cores, on a
    bounded sample of the same workload (same deck at sample_nx^3, a few cycles)"""
    from oracle import akref
    try:
        nsee = len(os.sched_getaffinity(0))
    except AttributeError:
        nsee = os.cpu_count() or 1
    # what the container may actually use: the cgroup CPU quota (the GPU boxes show 256 hardware threads
    # and grant 16 CPUs: cpu.max = "1600000 100000"); threads beyond it only take turns
    navail, quota = nsee, None
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            t = open(f).read().split()
            if f.endswith("cpu.max"):
                if t[0] != "max":
                    quota = int(t[0])/int(t[1])
            elif int(t[0]) > 0:
                quota = int(t[0])/int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    if quota:
        navail = max(1, min(nsee, int(round(quota))))
    akref.lib()
    n = args.cpu_sample_nx
    rec = args.recon or "plm"
    ngc = args.ng or (4 if rec in ("ppm4", "ppmx", "wenoz") else 2)
    if args.problem == "orszag_tang":
        kw = dict(is_mhd=1, recon=rec, rsolver="hlld", gamma=1.666666667, pgen="orszag_tang",
                  bcs=["periodic"]*6)
    elif args.problem == "sod":
        kw = dict(is_mhd=0, recon=rec, rsolver="hllc", gamma=1.4, pgen="shock_tube", shock_dir=1,
                  xshock=0.0, wl=[1.0, 0, 0, 0, 1.0, 0, 0, 0], wr=[0.125, 0, 0, 0, 0.1, 0, 0, 0],
                  bcs=["outflow", "outflow", "periodic", "periodic", "periodic", "periodic"])
    else:
        kw = dict(is_mhd=0, recon=rec, rsolver="hllc", gamma=1.66666666667, pgen="linear_wave",
                  wave_flag=0, amp=1e-3, dens=1.0, pgas=0.6, bcs=["periodic"]*6, x1min=0.0,
                  x1max=3.0, x2min=0.0, x2max=1.5, x3min=0.0, x3max=1.5)

    def timed(threads, budget):
        akref.lib().akref_set_threads(threads)
        s = akref.Sim(nx1=n, nx2=n, nx3=n, mb_nx1=n, mb_nx2=n, mb_nx3=n, ng=ngc, nstages=2, cfl=0.3,
                      tlim=1e9, nlim=-1, **kw)
        s.initialize()
        s.step()                   # warm-up (page faults, caches)
        t0 = time.time()
        cyc = 0
        while (time.time() - t0 < budget and cyc < 400) or cyc < 1:
            s.step()
            cyc += 1
        dt = time.time() - t0
        s.close()
        return n**3*cyc/dt/1e6, cyc

    # the oracle's OpenMP loops run over the flattened (block, k, j) rows (the Kokkos-OpenMP /
    # flat-MPI semantics of SURVEY 8(d)): 1 core, half and all of the CPUs the container is granted, and
    # twice that (to show that more threads than the quota do not help)
    candidates = sorted({1, max(1, navail//2), navail, min(nsee, 2*navail)})
    best = None
    results = []
    for th in candidates:
        v, cyc = timed(th, 4.0)
        results.append("%d thr: %.3f" % (th, v))
        if best is None or v > best[0]:
            best = (v, th, cyc)
    return {"value": round(best[0], 4), "unit": "Mcell-updates/s", "cores": best[1], "kind": "port",
            "sample": "%s %d^3 RK2 %s, ~4 s per thread count, oracle = port of the reference's "
                      "split-kernel CPU sequence, OpenMP over (block,k,j); Mcell-updates/s by threads: %s "
                      "(the host shows %d hardware threads; the cgroup CPU quota of this container is %s)" % (
                          args.problem, n, rec, "; ".join(results), nsee, ("%g CPUs" % quota) if quota else "unlimited")}


class PythonHost:
    """the Python host (athenak_amd.main) in three steps -- build, W warm-up cycles, K timed cycles -- so that at
    N > 1, where the C++ host is measured first, it can stop after the warm-up once it has confirmed that result;
    the stage launch group (akmi_*_stage_phase) is bracketed by HIP event pairs INSIDE the timed loop"""

    def __init__(self, args, pin, rank, world):
        from athenak_amd.main import Simulation
        self.args, self.world = args, world
        self.sim = Simulation(pin, my_rank=rank, nranks=world)
        pm, drv = self.sim.pmesh, self.sim.pdriver
        self.info = {"fused": bool(self.sim.phys.fused),
                     "ncell_rank": pm.pmb_pack.nmb_thispack*pm.NumberOfMeshBlockCells(),
                     "ncell_total": pm.nmb_total*pm.NumberOfMeshBlockCells(), "nstage": drv.nexp_stages,
                     "ng": pm.mb_indcs.ng,
                     "host": "Python (athenak_amd.main)" + ("" if world == 1 else
                                                            ", halos by torch.distributed batch_isend_irecv")}

    def barrier(self):
        import torch
        torch.cuda.synchronize()
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    def warm(self):
        """W untimed cycles; returns (time, dt) there"""
        pm, drv = self.sim.pmesh, self.sim.pdriver
        for _ in range(self.args.warmup):
            drv._cycle(pm)
        self.barrier()
        return float(pm.time), float(pm.dt)

    def timed(self):
        import torch
        sim, args = self.sim, self.args
        pm, drv = sim.pmesh, sim.pdriver
        sim.phys.stage_events = []
        from athenak_amd import bvals as _bv
        _bv.HALO_PROF = _bv.HaloProfile() if self.world > 1 else None
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            drv._cycle(pm)
        self.barrier()
        el = time.perf_counter() - t0
        if _bv.HALO_PROF is not None:
            self.info["halo"] = _bv.HALO_PROF.summary(args.steps*drv.nexp_stages, args.steps)
            _bv.HALO_PROF = None
        if self.world > 1:
            import torch.distributed as dist
            t = torch.tensor([el], dtype=torch.float64,
                             device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        evs, sim.phys.stage_events = sim.phys.stage_events, None
        group_ms = sum(a.elapsed_time(b) for a, b in evs)
        info = self.info
        info.update(el=el, steps=args.steps, value=info["ncell_total"]*args.steps/el/1e6,
                    ms_per_step=el/args.steps*1e3, group_ms=group_ms, group_calls=len(evs), time=float(pm.time),
                    dt=float(pm.dt), ncycle=int(pm.ncycle))
        return info

    def close(self):
        import torch
        self.sim = None
        torch.cuda.empty_cache()


HALO_KEYS = ("pack_ms", "exposed_wait_ms", "unpack_ms", "dt_reduce_ms")


def comm_profile_read(L, nstages, ncycles):
    """akmi_comm_profile_read -> the per-stage / per-cycle figures of roofline.halo (C++ host, RCCL transport): where a
    multi-rank stage spends its exchange.  Pack, exposed wait and unpack are sums over the channels of a stage (cell-centred +
    face-centred), averaged over the timed stages; the dt all-reduce + read-back is per cycle."""
    import ctypes as C
    from athenak_amd import capi
    out = (C.c_double*12)()
    capi.check(L.akmi_comm_profile_read(out, 12), "comm_profile_read")
    v = list(out)
    ns, nc = max(nstages, 1), max(ncycles, 1)
    return {"pack_ms": round(v[0]/ns, 5), "exposed_wait_ms": round(v[2]/ns, 5), "unpack_ms": round(v[4]/ns, 5),
            "dt_reduce_ms": round(v[6]/nc, 5), "bytes_sent_per_stage": int(v[8]/ns), "peers": int(v[10]),
            "rccl_ranks": int(v[11]), "posts_per_stage": round(v[9]/ns, 3),
            "event_pairs": {"pack": int(v[1]), "wait": int(v[3]), "unpack": int(v[5]), "dt_reduce": int(v[7])},
            "what": "HIP event pairs on the compute stream of rank 0 inside the timed loop: pack kernels of the off-rank "
                    "segments; the stall at hipStreamWaitEvent(receives done) -- zero when the transfer finished under the "
                    "kernels enqueued before it; unpack kernels; per cycle the ncclAllReduce(min) of dt + its read-back"}


def run_cpp_host(args, pin):
    """the same W + K cycles through the C++ host (akmi_sim_*: Mesh/TaskList/Driver in C++, one C-ABI call per
    task) in this process (one rank); launch-group timing by akmi_sim_profile (HIP events in the timed loop)"""
    import ctypes as C
    import torch
    from athenak_amd import capi, native
    L = capi.lib()
    self_ex = os.environ.get("AKMI_SELF_EXCHANGE", "0") == "1"
    if self_ex:
        # functional run of the transport on ONE GPU: a one-rank RCCL communicator; every ghost zone of the block travels
        # pack -> ncclSend/ncclRecv to self on the communicator's stream -> unpack, as on a rank with 26 off-rank
        # neighbours (csrc/akmi_host_comm.cpp SelfExchange); results do not change, roofline.halo shows the parts
        idb = C.create_string_buffer(128)
        capi.check(L.akmi_comm_unique_id(idb), "comm_unique_id")
        capi.check(L.akmi_comm_init_rccl(0, 1, idb.raw), "comm_init_rccl")
    sim = native.NativeSimulation(pin)
    pm = sim.pmesh
    nstage = {"rk1": 1, "rk2": 2, "rk3": 3, "rk4": 4}[pin.GetString("time", "integrator")]
    info = {"ncell_rank": pm.pmb_pack.nmb_thispack*pm.NumberOfMeshBlockCells(),
            "ncell_total": pm.nmb_total*pm.NumberOfMeshBlockCells(), "nstage": nstage, "ng": pm.mb_indcs.ng}
    sim.Execute(max_cycles=args.warmup)
    capi.check(L.akmi_sim_profile(sim.h, 1), "sim_profile")
    capi.check(L.akmi_comm_profile(1), "comm_profile")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = sim.Execute(max_cycles=args.steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms, calls = C.c_double(0.0), C.c_longlong(0)
    capi.check(L.akmi_sim_profile_read(sim.h, C.byref(ms), C.byref(calls)), "sim_profile_read")
    info["halo"] = comm_profile_read(L, done*nstage, done)
    capi.check(L.akmi_comm_profile(0), "comm_profile")
    ra = os.environ.get("AKMI_RUN_AHEAD", "1") != "0" and pm.nranks == 1 and not pm.multilevel
    info.update(host="C++ (akmi_sim_*: Mesh, TaskList, Driver in C++; one C-ABI call per task%s)" % (
                    "; new time step on the device, host one cycle ahead, drained inside the timed region" if ra else ""),
                el=el, steps=done,
                value=info["ncell_total"]*done/el/1e6, ms_per_step=el/max(done, 1)*1e3, group_ms=ms.value,
                group_calls=calls.value, time=sim.time, dt=sim.dt, ncycle=sim.ncycle)
    sim.close()
    if self_ex:
        native.finalize_comm()
    torch.cuda.empty_cache()
    return info


def valu_floor(blk, nx, plain):
    """second roof of the stage: the fp64 issue time of the VALU instructions the stage kernels execute
    (SQ_INSTS_VALU of a separate counter run, tools/pmc_valu.sh -> profiles/valu_counters_latest.json, accepted
    only for the library / sources of this run): instructions x 4 cycles / (1024 SIMDs x 2.4 GHz)"""
    f = os.path.join(ROOT, "profiles", "valu_counters_latest.json" if blk == "mhd" else "valu_counters_hydro_latest.json")
    if not (plain and nx == 256 and os.path.exists(f)):
        return None
    t = json.load(open(f))
    if t.get("lib_sha16") != lib_sha16() and t.get("src_sha16") != src_sha16():
        return {"ms": None, "source": "profiles/%s is of another build (sources %s, this run %s)" % (
            os.path.basename(f), t.get("src_sha16"), src_sha16())}
    insts = sum(v["insts_valu_per_launch"]*v.get("launches_per_stage", 1) for k, v in t["kernels"].items())
    return {"ms": round(insts*4.0/(1024*2.4e9)*1e3, 4), "valu_wave_insts_per_stage": round(insts),
            # the pipe takes an fp64 instruction every 4 cycles, but a wave issues a DEPENDENT one only every ~8: at the two
            # to three waves per SIMD of the stage kernels a division / square-root chain reaches ~2.2 ns per instruction and
            # SIMD (tools/micro/fp64_issue.hip, profiles/r05_fp64_issue.txt) -- the roof such a stream really sits under
            "ms_dependent_issue": round(insts*2.2e-9/1024*1e3, 4),
            "source": "profiles/%s (%s): SQ_INSTS_VALU x 4 cycles / (256 CUs x 4 SIMDs x 2.4 GHz); ms_dependent_issue: "
                      "x 2.2 ns per instruction and SIMD (measured issue rate of dependent fp64 chains at 2-3 waves per SIMD)" % (
                os.path.basename(f), t.get("tag", ""))}


def self_launch(args):
    """`python3 bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks (one per GPU) through
    torch.distributed.run on the loopback address; rank 0's JSON line goes to our stdout, the launcher's exit
    status becomes ours"""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.native_child:
        self_launch(args)
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (start it plain, or with torch.distributed.run "
                 "--nproc-per-node %d)" % (args.gpus, world, args.gpus))
    # developer knobs for a functional check of the N>1 path on a 1-GPU box (RCCL refuses two ranks on one
    # device): AKMI_SHARE_GPU=1 puts every rank on cuda:0, AKMI_DIST_BACKEND=gloo moves the halos through pinned
    # host buffers.  Numbers from such a run mean nothing.
    backend = os.environ.get("AKMI_DIST_BACKEND", "nccl")
    if os.environ.get("AKMI_SHARE_GPU", "0") == "1":
        local = 0
    torch.cuda.set_device(local)

    nblk = block_grid(world)
    pin, blk = make_pin(args, nblk)
    if args.native_child:
        return native_child(args, pin, rank, world)
    if args.native:
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(backend)
        return main_native(args, pin, blk, nblk, rank, world)

    # ---- the two hosts ------------------------------------------------------------------------------------
    # The C++ host is the path north_star names ("host code stays C++ ... RCCL"): it is the headline whenever it
    # completed the K cycles and its (time, dt) agree with the Python host's at the same cycle -- a wrong or skipped
    # halo exchange cannot win the line by being faster.
    #   N = 1: both hosts run the same W + K cycles in this process, the C++ host first; the Python host is reported as
    #          other_host.
    #   N > 1: the C++ host + RCCL is measured FIRST (one child process per rank, so that neither an abort nor a
    #          stall inside a transport that could only be exercised to self where it was built costs the bench);
    #          the Python host then runs its W warm-up cycles, and if the C++ host stood at the same (time, dt)
    #          after ITS warm-up the line is the C++ host's and the Python host stops there.  Only when the C++
    #          host gave nothing, or something else, does the Python host run the K timed cycles (the fallback).
    chk = os.environ.get("AKMI_BENCH_NATIVE_CHECK", "1")
    cpp, why, cpp_ok = None, None, False
    if world == 1:
        # the C++ host first, as at N > 1 (it is the headline; consecutive runs of one process drift by a few per cent
        # with the state of the chip -- profiles/r04_numerics_ab.txt -- and the order should not depend on N)
        if chk != "0":
            try:
                cpp = run_cpp_host(args, pin)
            except Exception as e:           # must never take the measurement down
                why = "C++ host failed: %r" % (e,)
        else:
            why = "C++ host switched off (AKMI_BENCH_NATIVE_CHECK=0)"
        host = PythonHost(args, pin, rank, world)
        host.warm()
        py = host.timed()
        host.close()
        if cpp is not None:
            if cpp.get("steps") != args.steps:
                why = "C++ host completed %s of %d cycles" % (cpp.get("steps"), args.steps)
            elif cpp["time"] != py["time"] or cpp["dt"] != py["dt"]:
                why = ("C++ host ended at (t, dt) = (%r, %r), the Python host at (%r, %r): not accepted"
                       % (cpp["time"], cpp["dt"], py["time"], py["dt"]))
            else:
                cpp_ok = True
        check = "both hosts ended at t = %r, dt = %r after %d + %d cycles" % (
            py["time"], py["dt"], args.warmup, args.steps)
    else:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if (backend == "nccl" and chk != "0") or chk == "force":
            cpp = native_check(args, rank, world)          # rank 0 gets the result
            if rank == 0 and (cpp is None or "failed" in cpp):
                why = "C++ host gave no result: %s" % ((cpp or {}).get("failed", "another rank's child failed (see stderr)"))
                cpp = None
        elif rank == 0:
            why = "C++ host not run (%s)" % ("switched off" if chk == "0" else "backend is not RCCL")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        host = PythonHost(args, pin, rank, world)
        t_w = host.warm()
        verdict = [None]
        if rank == 0:
            if cpp is not None and cpp.get("steps") != args.steps:
                why, cpp = "C++ host completed %s of %d cycles" % (cpp.get("steps"), args.steps), None
            if cpp is not None and args.warmup > 0:
                if (cpp.get("time_w"), cpp.get("dt_w")) == t_w:
                    verdict[0] = "accept"
                else:
                    why = ("C++ host stood at (t, dt) = (%r, %r) after the %d warm-up cycles, the Python host at "
                           "(%r, %r): not accepted" % (cpp.get("time_w"), cpp.get("dt_w"), args.warmup, *t_w))
                    cpp = None
        dist.broadcast_object_list(verdict, src=0)
        # AKMI_BENCH_FULL_CHECK=0: accept the C++ host on the warm-up state alone and skip the Python host's timed cycles
        # (default: the Python host runs all K cycles too and the END states are compared -- an exchange error that only
        #  shows later than the warm-up cannot take the line, and the line carries the other host's number)
        if verdict[0] == "accept" and os.environ.get("AKMI_BENCH_FULL_CHECK", "1") == "0":
            py, cpp_ok = dict(host.info), True
            check = "both hosts stood at t = %r, dt = %r after the %d warm-up cycles (end states not compared)" % (
                *t_w, args.warmup)
        else:
            py = host.timed()
            if rank == 0 and cpp is not None:           # no warm-up to compare at: compare at the end
                if cpp["time"] == py["time"] and cpp["dt"] == py["dt"]:
                    cpp_ok = True
                    check = "both hosts ended at t = %r, dt = %r after %d + %d cycles" % (
                        py["time"], py["dt"], args.warmup, args.steps)
                else:
                    why = ("C++ host ended at (t, dt) = (%r, %r), the Python host at (%r, %r): not accepted"
                           % (cpp["time"], cpp["dt"], py["time"], py["dt"]))
        host.close()
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    head, other = (cpp, py) if cpp_ok else (py, cpp)
    if "value" not in (other or {}):
        other = None                                       # the Python host stopped after confirming the C++ host
    for k in ("ncell_rank", "ncell_total", "nstage", "ng"):
        head.setdefault(k, py[k])

    # ---- roofline of the dominant launch group: HIP events of the headline host's timed loop ----------------
    # akmi_*_stage_fused / _stage_phase = the whole stage except the halo exchange, the BCs and the ghost-shell
    # c2p: algorithmic bytes = SURVEY 8(d)'s per-cell-stage figure (MHD 384 B, hydro 240 B).
    nst = args.steps*py["nstage"]
    stage_bytes = BYTES_PASS_A[blk] + BYTES_PASS_B[blk]
    ncell_rank = py["ncell_rank"]
    plain = not (args.recon or args.ng or args.set or args.split or args.mb) and args.problem in ("orszag_tang", "sod")
    if head.get("group_calls"):
        tS = head["group_ms"]*1e-3/nst                        # seconds per stage in the launch group
        rest_ms = head["ms_per_step"]/py["nstage"] - tS*1e3
        kname = ("the stage's kernel calls akmi_%s_stage_phase (sweeps + update%s) + akmi_%s_c2p_newdt (ConsToPrim of all "
                 "cells + CFL scan), i.e. the stage without ghost fill and boundary conditions" % (
                     blk, " + CornerE + CT" if blk == "mhd" else "", blk)) if world == 1 else \
                "akmi_%s_stage_phase x3 of a rank (the halo messages travel between the phases)" % blk
    else:
        # no launch-group events: the hosts took the task-granular chain (small 3-D pack, --split) -- the roofline is
        # then the whole stage of a rank (every task, ghost fill and BCs included), and the fused kernels' VALU floor
        # does not apply
        tS, rest_ms = head["ms_per_step"]*1e-3/py["nstage"], None
        kname = "whole stage of a rank (task-granular chain: one kernel per reference task, incl. ghost exchange and BCs)"
    ach = stage_bytes*ncell_rank/tS/1e9
    traffic, tsrc = None, None
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json" if blk == "mhd" else
                         "pmc_traffic_hydro_latest.json")
    if os.path.exists(tfile) and args.nx == 256 and plain:
        # HBM-side bytes per stage from rocprofv3 PMC counters (FETCH_SIZE / WRITE_SIZE in separate passes,
        # calibrated on a copy of known size; tools/pmc.sh), recorded for this workload in a separate run
        t = json.load(open(tfile))
        stage_kernels = [k for k in t["kernels"] if k.startswith("akmi::k_sweep") or
                         k.startswith("akmi::k_corner") or k.startswith("akmi::k_ct_copy") or
                         k.startswith("akmi::k_hydro_stage3d") or k.startswith("akmi::k_c2p_newdt")]
        same_lib = t.get("lib_sha16") == lib_sha16()
        same_src = t.get("src_sha16") is not None and t.get("src_sha16") == src_sha16()
        if same_lib or same_src:
            # (the counter run does 2 RK2 cycles = 4 stages: a kernel launched once per stage has 4 launches, 5 with the
            #  conversion of Driver::Initialize; a kernel that only runs there -- the hydro path converts inside its stage
            #  kernel -- does not belong to a stage)
            traffic = round(sum(t["kernels"][k]["hbm_bytes_per_launch"]*round(t["kernels"][k].get("launches", 4)/4.0)
                                for k in stage_kernels))
            tsrc = "profiles/%s (%s, %s)" % (os.path.basename(tfile), t.get("tag", ""),
                                              ("lib %s" % t["lib_sha16"]) if same_lib else
                                              ("same sources %s, rebuilt library" % t["src_sha16"]))
        else:
            tsrc = "profiles/%s is of another build (lib %s / sources %s, this run %s / %s): traffic not reported" % (
                os.path.basename(tfile), t.get("lib_sha16"), t.get("src_sha16"), lib_sha16(), src_sha16())
    whole = stage_bytes*ncell_rank*nst/head["el"]/1e9
    roofline = {"bound": "hbm", "kernel": kname,
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach/HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": tsrc,
                "algorithmic_bytes_per_cell_stage": stage_bytes,
                "algorithmic_bytes_per_launch": stage_bytes*ncell_rank,
                "ms_per_launch": round(tS*1e3, 4),
                "timing": "HIP event pairs on the launch stream around every call of the group inside the timed "
                          "loop of the headline host (%d calls)" % (head.get("group_calls") or 0),
                "halo_bcs_shell_c2p_ms": None if rest_ms is None else round(rest_ms, 4),
                # N > 1 (or AKMI_SELF_EXCHANGE=1): the exchange of the C++ host taken apart -- pack, exposed wait, unpack per
                # stage, dt all-reduce per cycle, bytes and peers, the ranks RCCL reports (akmi_comm_profile)
                "halo": (head.get("halo") if (head.get("halo") or {}).get("posts_per_stage") else None),
                "valu_floor": valu_floor(blk, args.nx, plain) if head.get("group_calls") else None,
                "whole_stage": {"achieved": round(whole, 1), "frac": round(whole/HBM_PEAK_GBS, 4)},
                "note": "two roofs: HBM (algorithmic bytes / 8 TB/s) and fp64 issue (valu_floor.ms)%s: DESIGN.md 3" % (
                    "; the measured traffic is %.2fx the algorithmic bytes (intermediates between the kernels of the stage, "
                    "tile halos counted per XCD)" % (traffic/float(stage_bytes*ncell_rank)) if traffic else "")}

    rname = (args.recon or "plm").upper()
    halo = "none (single periodic block: same-rank gather)" if world == 1 else (
        "RCCL called from the C++ host: grouped ncclSend/ncclRecv per variable class on the communicator's stream, "
        "ncclAllReduce(min) for dt; per-stage U and B messages posted under CornerE/CT and the interior c2p"
        if cpp_ok else "RCCL send/recv through torch.distributed, per-stage U and B messages posted under "
                       "CornerE/CT and the interior c2p")
    out = {"metric": "Mcell-updates/s (3D MHD %s+HLLD+CT RK2, %d^3 cells per GPU)" % (rname, args.nx)
           if blk == "mhd" else "Mcell-updates/s (3D hydro %s+HLLC RK2, %d^3 per GPU)" % (rname, args.nx),
           "value": round(head["value"], 2), "unit": "Mcell-updates/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(head["ms_per_step"], 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
           "data": "synthetic (closed-form %s initial condition)" % args.problem,
           "lib_sha16": lib_sha16(), "src_sha16": src_sha16(),
           "config": {"workload": "%s 3D, %s, %d^3 cells per GPU, mesh %dx%dx%d in %dx%dx%d "
                                  "MeshBlocks, cfl 0.3, RK2, ng=%d" % (
                                      args.problem, ("ideal MHD %s+HLLD+CT" % rname) if blk == "mhd" else
                                      ("ideal hydro %s+HLLC" % rname), args.nx, args.nx*nblk[0],
                                      args.nx*nblk[1], args.nx*nblk[2],
                                      *[b*args.nx//(args.mb or args.nx) for b in nblk], py["ng"]),
                      "path": "fused stage" if py["fused"] else
                              ("task-granular" if args.split else
                               "task-granular (chosen by the hosts for a small 3-D pack: one thread per face beats the "
                               "marching kernels there)"),
                      "host": head["host"] + ("" if world == 1 or not cpp_ok else ", one child process per rank"),
                      "halo": halo},
           "roofline": roofline}
    if why:
        out["config"]["host_note"] = why
    if cpp_ok:
        out["config"]["host_check"] = check
    elif why:
        out["config"]["host_check"] = "C++ host not the headline: " + why
    out["other_host"] = None if other is None else {
        "host": other["host"], "value": round(other["value"], 2), "ms_per_step": round(other["ms_per_step"], 4),
        "stage_group_ms": round(other["group_ms"]/nst, 4) if other.get("group_calls") else None}
    if args.set:
        out["config"]["workload"] += " + " + " ".join(args.set)
    if world == 1 and plain and args.nx == 256 and args.problem == "orszag_tang" and not args.no_other_configs \
            and not args.no_cpu_baseline:        # (the profiling tools run with --no-cpu-baseline: headline kernels only)
        out["other_configs"] = other_configs(args)
    if world == 1 and not args.no_cpu_baseline and not args.set:
        out["cpu_baseline"] = cpu_baseline(args, blk)
    print(json.dumps(out), flush=True)


def other_configs(args):
    """Side measurements of the default one-GPU run, AFTER the timed region of the headline and never part of
    `value`: the other single-GPU workloads of BASELINE.json through the C++ host, whole-run Mcell-updates/s --
    configs[1] (sod 3-D, 128^3, one MeshBlock, PLM+HLLC) and configs[4]'s mesh on one GPU (3-D MHD blast, two levels
    of static refinement, PPM4+HLLD+CT, ng = 4: the deck as shipped, 120 x 16^3, and at production size, 960 x 32^3).
    A failure here is recorded, it never takes the line down."""
    import copy
    import torch
    from athenak_amd import native
    from athenak_amd.main import load_deck
    res = []

    def run(label, pin, warm, steps):
        try:
            sim = native.NativeSimulation(pin)
            pm = sim.pmesh
            ncell = pm.nmb_total*pm.NumberOfMeshBlockCells()
            sim.Execute(max_cycles=warm)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            done = sim.Execute(max_cycles=steps)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            sim.close()
            torch.cuda.empty_cache()
            # whole-step roofline of this workload: SURVEY 8(d)'s algorithmic bytes per cell and stage (hydro 240 B,
            # MHD 384 B) x cells x 2 stages (RK2) / wall time of the timed cycles -- every kernel, exchange and boundary
            # condition of the cycle included (kernel-time figures belong to a rocprof profile of the same lib_sha16:
            # profiles/r05_hydro128_kernel_stats.txt, r05_config5.txt)
            blk_ = "mhd" if pin.DoesBlockExist("mhd") else "hydro"
            sb = BYTES_PASS_A[blk_] + BYTES_PASS_B[blk_]
            ach = sb*ncell*2*done/el/1e9
            res.append({"config": label, "value": round(ncell*done/el/1e6, 1), "unit": "Mcell-updates/s",
                        "meshblocks": int(pm.nmb_total), "cells": int(ncell), "steps": int(done),
                        "ms_per_step": round(el/max(done, 1)*1e3, 4),
                        "roofline": {"bound": "hbm", "basis": "whole step (wall clock of the timed cycles)",
                                     "algorithmic_bytes_per_cell_stage": sb, "stages_per_step": 2,
                                     "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(ach/HBM_PEAK_GBS, 4)}})
        except Exception as e:
            res.append({"config": label, "value": None, "error": repr(e)[:300]})

    a = copy.copy(args)
    a.problem, a.nx = "sod", 128
    pin, _ = make_pin(a, (1, 1, 1))
    run("configs[1]: sod 3D, 128^3 single MeshBlock, ideal hydro PLM+HLLC, RK2, C++ host", pin, 5, 40)
    # the same hydro scheme at the headline's size, and config 5's numerics (PPM4 + HLLD, ng = 4) on the headline's uniform mesh
    a = copy.copy(args)
    a.problem, a.nx = "sod", 256
    pin, _ = make_pin(a, (1, 1, 1))
    run("hydro at the headline's size: sod 3D, 256^3 single MeshBlock, ideal hydro PLM+HLLC, RK2, C++ host", pin, 3, 20)
    a = copy.copy(args)
    a.problem, a.nx, a.recon = "orszag_tang", 256, "ppm4"
    pin, _ = make_pin(a, (1, 1, 1))
    run("config 5's numerics on the headline's mesh: orszag_tang 3D, 256^3 single MeshBlock, MHD PPM4+HLLD+CT, ng=4, RK2, "
        "C++ host", pin, 3, 15)
    ov = ["time/nlim=-1", "time/tlim=1.0e9"]
    run("configs[4] mesh on one GPU, deck size: blast 3D MHD, 2-level static refinement, 120 MeshBlocks of 16^3, "
        "PPM4+HLLD+CT, ng=4, C++ host", load_deck("blast_mhd_smr.athinput", ov), 5, 40)
    prod = ["mesh/nx%d=256" % q for q in (1, 2, 3)] + ["meshblock/nx%d=32" % q for q in (1, 2, 3)]
    run("configs[4] mesh on one GPU, production size: 960 MeshBlocks of 32^3 (256^3 root grid), C++ host",
        load_deck("blast_mhd_smr.athinput", ov + prod), 2, 8)
    return res


def native_check(args, rank, world):
    """The same workload and the same K timed cycles through the C++ host with RCCL called directly
    (ncclSend/ncclRecv groups on the communicator stream, ncclAllReduce for dt).  Every rank starts a
    CHILD process for it (same GPU, communicator bootstrapped over TCP by akmi_comm_init_env): this
    transport could never be run on more than one GPU where the library was built, and neither an abort
    inside it nor a stall may cost the measurement already taken.  The child gets 150 s; rank 0's child
    leaves its timing (barrier-bracketed, max over ranks) in a file.  Returns that dict on rank 0."""
    import subprocess
    import tempfile
    res = os.path.join(tempfile.gettempdir(), "akmi_bench_native_%d_%d.json" % (os.getpid(), rank))
    cmd = [sys.executable, os.path.abspath(__file__), "--native-child", res, "--gpus", str(world),
           "--nx", str(args.nx), "--problem", args.problem, "--steps", str(args.steps),
           "--warmup", str(args.warmup)]
    if args.mb:
        cmd += ["--mb", str(args.mb)]
    if args.split:
        cmd += ["--split"]
    if args.recon:
        cmd += ["--recon", args.recon]
    if args.ng:
        cmd += ["--ng", str(args.ng)]
    for s_ in (args.set or []):
        cmd += ["--set", s_]
    out = None
    # N cold RCCL initialisations side by side (bootstrap over TCP, one communicator of N ranks) + N library loads: the
    # limit grows with N; AKMI_BENCH_CHILD_TIMEOUT overrides it
    limit = float(os.environ.get("AKMI_BENCH_CHILD_TIMEOUT", 120 + 30*world))
    errf = res + ".stderr"
    note = None
    try:
        with open(errf, "w") as ef:
            p = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=ef)
            try:
                rc = p.wait(timeout=limit)
                if rc != 0:
                    note = "child ended with status %d" % rc
                elif rank == 0 and os.path.exists(res):
                    out = json.load(open(res))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
                note = "no result within %.0f s, child stopped" % limit
    except Exception as e:     # must never take the bench down
        note = "failed: %r" % (e,)
    tail = ""
    if os.path.exists(errf):
        txt = open(errf).read()
        sys.stderr.write(txt)                               # the child's messages stay visible in the bench's stderr
        tail = " | ".join(l.strip() for l in txt.strip().splitlines()[-4:])[-600:]
        os.remove(errf)
    if note:
        sys.stderr.write("[C++ host] rank %d: %s\n" % (rank, note))
        if rank == 0:
            out = {"failed": "rank 0 %s; its last messages: %s" % (note, tail or "(none)")}
    sys.stderr.flush()
    if os.path.exists(res):
        os.remove(res)
    return out


def native_child(args, pin, rank, world):
    """body of the child process of native_check: C++ host + RCCL, no torch.distributed"""
    import ctypes as C
    import torch
    from athenak_amd import capi, native
    L = capi.lib()
    capi.check(L.akmi_comm_init_env(), "comm_init_env")
    sim = native.NativeSimulation(pin)
    ncell_total = sim.pmesh.nmb_total*sim.pmesh.NumberOfMeshBlockCells()

    def allmin(x):
        v = (C.c_double*1)(x)
        capi.check(L.akmi_comm_allreduce_min(v, 1, capi._stream()), "comm_allreduce_min")
        return v[0]

    sim.Execute(max_cycles=args.warmup)
    time_w, dt_w = sim.time, sim.dt               # compared with the Python host's after the same cycles
    capi.check(L.akmi_sim_profile(sim.h, 1), "sim_profile")
    capi.check(L.akmi_comm_profile(1), "comm_profile")
    torch.cuda.synchronize()
    allmin(0.0)                                   # barrier
    t0 = time.perf_counter()
    n = sim.Execute(max_cycles=args.steps)
    torch.cuda.synchronize()
    el = -allmin(-(time.perf_counter() - t0))     # max over ranks
    ms, calls = C.c_double(0.0), C.c_longlong(0)
    capi.check(L.akmi_sim_profile_read(sim.h, C.byref(ms), C.byref(calls)), "sim_profile_read")
    nstage = {"rk1": 1, "rk2": 2, "rk3": 3, "rk4": 4}[pin.GetString("time", "integrator")]
    halo = comm_profile_read(L, n*nstage, n)
    if rank == 0:
        sys.stderr.write("[C++ host] exchange per stage on rank 0: pack %.4f ms, exposed wait %.4f ms, unpack %.4f ms, "
                         "%d bytes to %d peers; dt all-reduce + read-back %.4f ms per cycle; ncclCommCount = %d\n" % (
                             halo["pack_ms"], halo["exposed_wait_ms"], halo["unpack_ms"], halo["bytes_sent_per_stage"],
                             halo["peers"], halo["dt_reduce_ms"], halo["rccl_ranks"]))
        sys.stderr.write("[C++ host] RCCL called directly, %d GPUs: %.2f Mcell-updates/s, %.4f ms/step "
                         "(%d cycles, t=%.6e dt=%.6e)\n" % (world, ncell_total*n/el/1e6, el/n*1e3, n,
                                                            sim.time, sim.dt))
        sys.stderr.flush()
        with open(args.native_child, "w") as f:
            json.dump({"value": ncell_total*n/el/1e6, "ms_per_step": el/n*1e3, "steps": n, "el": el,
                       "time": sim.time, "dt": sim.dt, "time_w": time_w, "dt_w": dt_w, "group_ms": ms.value,
                       "group_calls": calls.value, "halo": halo,
                       "host": "C++ (akmi_sim_*: Mesh, TaskList, Driver in C++) + RCCL called directly"}, f)
    sim.close()
    native.finalize_comm()


def main_native(args, pin, blk, nblk, rank, world):
    """the same timed region with the C++ host: Mesh/TaskList/Driver in C++, one C-ABI call per task,
    RCCL called directly for the halos and the dt reduction when world > 1"""
    import torch
    from athenak_amd import native
    transport = "none"
    if world > 1:
        import torch.distributed as dist
        transport = native.init_comm_from_torch_distributed()
    sim = native.NativeSimulation(pin)
    pm = sim.pmesh
    ncell_rank = pm.pmb_pack.nmb_thispack*pm.NumberOfMeshBlockCells()
    ncell_total = pm.nmb_total*pm.NumberOfMeshBlockCells()
    nstage = {"rk1": 1, "rk2": 2, "rk3": 3, "rk4": 4}[pin.GetString("time", "integrator")]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    sim.Execute(max_cycles=args.warmup)
    barrier()
    t0 = time.perf_counter()
    done = sim.Execute(max_cycles=args.steps)
    barrier()
    el = time.perf_counter() - t0
    assert done == args.steps, (done, args.steps)
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    stage_bytes = BYTES_PASS_A[blk] + BYTES_PASS_B[blk]
    ach = stage_bytes*ncell_rank*nstage*args.steps/el/1e9
    if rank == 0:
        rname = (args.recon or "plm").upper()
        out = {"metric": "Mcell-updates/s (3D MHD %s+HLLD+CT RK2, %d^3 cells per GPU)" % (rname, args.nx)
               if blk == "mhd" else "Mcell-updates/s (3D hydro %s+HLLC RK2, %d^3 per GPU)" % (rname, args.nx),
               "value": round(ncell_total*args.steps/el/1e6, 2), "unit": "Mcell-updates/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el/args.steps*1e3, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic (closed-form %s initial condition)" % args.problem,
               "lib_sha16": lib_sha16(),
               "config": {"workload": "%s 3D, %s %s, %d^3 cells per GPU, %dx%dx%d MeshBlocks, cfl 0.3, RK2, ng=%d" % (
                              args.problem, blk, rname, args.nx, *[b*args.nx//(args.mb or args.nx) for b in nblk],
                              pm.mb_indcs.ng),
                          "path": "fused stage", "host": "C++ (akmi_sim_*)",
                          "halo": "none (single periodic block)" if world == 1 else
                                  {"rccl": "RCCL called from the C++ host: grouped ncclSend/ncclRecv per variable "
                                           "class on the communicator's stream, ncclAllReduce(min) for dt",
                                   "callbacks": "host-staged callbacks (functional check only)"}[transport]},
               "roofline": {"bound": "hbm", "kernel": "whole stage incl. halo exchange (C++ host)",
                            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(ach/HBM_PEAK_GBS, 4), "traffic": None,
                            "algorithmic_bytes_per_cell_stage": stage_bytes}}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, blk)
        print(json.dumps(out))
    sim.close()
    if world > 1:
        native.finalize_comm()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
