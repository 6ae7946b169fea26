"""main(): deck -> Mesh -> physics -> ProblemGenerator -> Driver (src/main.cpp:61-420)."""
import os

from .driver import Driver
from .mesh import Mesh
from .parameter_input import ParameterInput
from .pgen import ProblemGenerator

_DECKS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "inputs")


class Simulation:
    """Steps 3-8 of the reference's main() (src/main.cpp:246-375)."""

    def __init__(self, pin, my_rank=0, nranks=1, initialize=True, restart=None):
        """restart = (header, record reader) of outputs.read_restart(): Mesh::BuildTreeFromRestart
        + the restart constructor of ProblemGenerator (main.cpp:329,355-360)"""
        self.pin = pin
        self.pmesh = Mesh(pin, my_rank, nranks)
        self.pmesh.AddCoordinatesAndPhysics(pin)
        if restart is not None:
            self._load_restart(*restart)
        self.pmesh.pgen = ProblemGenerator(pin, self.pmesh, restart=restart is not None)
        self.pdriver = Driver(pin, self.pmesh)      # after pgen: linear_wave rescales tlim
        if initialize:
            self.pdriver.Initialize(self.pmesh, pin)

    def _load_restart(self, hdr, record):
        import numpy as np
        import torch
        pm = self.pmesh
        ind = pm.mb_indcs
        got = (hdr["nmb_total"], hdr["mb_indcs"][:4], hdr["mesh_indcs"][:4])
        want = (pm.nmb_total, (ind.ng, ind.nx1, ind.nx2, ind.nx3),
                (pm.mesh_indcs.ng, pm.mesh_indcs.nx1, pm.mesh_indcs.nx2, pm.mesh_indcs.nx3))
        if got != want:
            raise RuntimeError("### FATAL ERROR mesh in the restart file %r does not match the "
                               "parameters %r" % (got, want))
        lloc = [tuple(int(x) for x in l[:4]) for l in hdr["lloc"]]          # (lx1, lx2, lx3, level) per block
        if lloc != [tuple(l)[:3] + (pm.level_of(g),) for g, l in enumerate(pm.lloc_eachmb)]:
            raise RuntimeError("### FATAL ERROR MeshBlock order or refinement levels of the restart file differ")
        pm.time, pm.dt, pm.ncycle = hdr["time"], hdr["dt"], hdr["ncycle"]   # build_tree.cpp:365-369
        pk = pm.pmb_pack
        arrays = []
        if pk.phydro is not None:
            arrays.append(pk.phydro.u0)
        if pk.pmhd is not None:
            arrays += [pk.pmhd.u0, pk.pmhd.b0.x1f, pk.pmhd.b0.x2f, pk.pmhd.b0.x3f]
        if sum(int(a[0].numel())*8 for a in arrays) != hdr["data_size"]:
            raise RuntimeError("### FATAL ERROR CC data size read from restart file not equal to size "
                               "of Hydro and/or MHD arrays, restart file is broken.")
        for m in range(pk.nmb_thispack):
            rec = record(pk.gids + m)
            off = 0
            for a in arrays:
                n = int(a[m].numel())
                a[m].copy_(torch.from_numpy(np.array(rec[off:off + n])).reshape(a[m].shape))
                off += n

    @property
    def phys(self):
        pk = self.pmesh.pmb_pack
        return pk.phydro if pk.phydro is not None else pk.pmhd

    def Execute(self, max_cycles=None):
        return self.pdriver.Execute(self.pmesh, self.pin, max_cycles)


def load_deck(name_or_path, overrides=()):
    path = name_or_path
    if not os.path.exists(path):
        path = os.path.join(_DECKS, name_or_path)
    pin = ParameterInput(filename=path)
    pin.ModifyFromCmdline(list(overrides))
    return pin


def load_restart(path, overrides=(), my_rank=0, nranks=1, initialize=True):
    """athena -r <file> [overrides]: main.cpp:248-293,329,355-365.  Outputs, if wanted, are created
    by the caller from sim.pin (their file_number/last_time continue from the file)."""
    from .outputs import read_restart
    text, hdr, record = read_restart(path)
    pin = ParameterInput(text=text)
    pin.ModifyFromCmdline(list(overrides))
    return Simulation(pin, my_rank, nranks, initialize=initialize, restart=(hdr, record))


def run_deck(name_or_path, overrides=(), my_rank=0, nranks=1, max_cycles=None):
    sim = Simulation(load_deck(name_or_path, overrides), my_rank, nranks)
    sim.Execute(max_cycles)
    return sim
