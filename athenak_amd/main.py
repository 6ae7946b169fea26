"""main(): deck -> Mesh -> physics -> ProblemGenerator -> Driver (src/main.cpp:61-420)."""
import os

from .driver import Driver
from .mesh import Mesh
from .parameter_input import ParameterInput
from .pgen import ProblemGenerator

_DECKS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "inputs")


class Simulation:
    """Steps 3-8 of the reference's main() (src/main.cpp:246-375)."""

    def __init__(self, pin, my_rank=0, nranks=1, initialize=True):
        self.pin = pin
        self.pmesh = Mesh(pin, my_rank, nranks)
        self.pmesh.AddCoordinatesAndPhysics(pin)
        self.pmesh.pgen = ProblemGenerator(pin, self.pmesh)
        self.pdriver = Driver(pin, self.pmesh)      # after pgen: linear_wave rescales tlim
        if initialize:
            self.pdriver.Initialize(self.pmesh, pin)

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


def run_deck(name_or_path, overrides=(), my_rank=0, nranks=1, max_cycles=None):
    sim = Simulation(load_deck(name_or_path, overrides), my_rank, nranks)
    sim.Execute(max_cycles)
    return sim
