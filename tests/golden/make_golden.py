#!/usr/bin/env python
"""Regenerate tests/golden/*.npz and known_answers.json.

Two kinds of data live here:
 * known_answers.json -- numbers that come from the REFERENCE: thresholds of its regression
   suite (tst/test_suite/nr/*.py, file:line recorded per entry) and the outputs of the
   reference binary recorded in BASELINE.md section 2b.  They pin the oracle
   (tests/test_oracle_pins.py).
 * *.npz -- small whole-run snapshots produced by the pinned oracle (inputs: deck overrides;
   expected outputs: conserved variables and face fields after N cycles).  They freeze the
   oracle against regressions and give the GPU tests a fixture that does not depend on
   re-running the oracle.  The reference itself cannot be built in the authoring container
   (Kokkos submodule empty), so no array-valued reference output exists to be stored.
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

KNOWN = {
    "lwave1d_thresholds": {
        "source": "tst/test_suite/nr/test_nr_lwave1d_cpu.py:15-96 (error(64), error(64)/error(32))",
        "run_arguments": "tst/test_suite/nr/test_nr_lwave1d_cpu.py:109-131",
        "values": {"hydro,rk2,plm,0": [2.1e-08, 0.28], "hydro,rk2,plm,4": [2.1e-08, 0.28],
                   "hydro,rk2,plm,3": [1.2e-08, 0.29], "hydro,rk3,plm,0": [1.8e-08, 0.28],
                   "hydro,rk2,ppm4,0": [1.7e-08, 0.35], "hydro,rk3,ppm4,0": [4.7e-09, 0.23],
                   "mhd,rk2,plm,0": [2.5e-08, 0.28], "mhd,rk2,plm,6": [2.5e-08, 0.28],
                   "mhd,rk2,plm,5": [1.7e-08, 0.29], "mhd,rk2,plm,1": [1.7e-08, 0.29],
                   "mhd,rk2,plm,4": [2.8e-08, 0.32], "mhd,rk2,plm,2": [2.8e-08, 0.32],
                   "mhd,rk2,ppm4,0": [2e-08, 0.35]}},
    "lwave1d_left_right_equal": {"source": "tst/test_suite/nr/test_nr_lwave1d_cpu.py:155-160"},
    "sod_convergence": {"source": "tst/test_suite/nr/test_nr_sod_cpu.py:20-46,65-86",
                        "max_ratio_256_over_128": 0.6},
    "divb": {"source": "tst/test_suite/nr/test_nr_divb_amr_mpicpu.py:38-40", "max": 2e-11},
    "reference_binary_outputs": {
        "source": "BASELINE.md section 2b",
        "hydro_lwave1d_rk2_plm_hllc": {"N32": "7.390252e-08", "N64": "2.052777e-08"},
        "mhd_lwave1d_rk2_plm_hlld": {"N32": "8.812266e-08", "N64": "2.448591e-08"},
        "c1_deck": {"ncycle": 855, "RMS-L1": "5.939209e-06", "L-inf": "2.880281e-08",
                    "d_L1": "2.880995e-06", "M1_L1": "2.880532e-06", "E_L1": "4.321644e-06"}},
}

SNAPSHOTS = [
    ("ot3d_16_mb8_plm_rk2_c3", ("orszag_tang", 16, 3, 8), 3, dict(cfl=0.3)),
    ("sod3d_16_plm_rk2_c4", ("sod", 16, 3, 16), 4, dict(cfl=0.3)),
    ("blast2d_24_ppm4_rk3_c3", ("blast", 24, 2, 12), 3, dict(integrator="rk3")),
    ("lwave_hydro1d_64_c10", ("linear_wave_hydro", 64, 1, 16), 10, dict(extra=["problem/along_x1=true"])),
]


def main():
    import parity_util as pu
    from oracle import akref
    from athenak_amd.main import load_deck
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(KNOWN, f, indent=1, sort_keys=True)
    for name, (problem, n, dims, mb), cycles, kw in SNAPSHOTS:
        deck, ov = pu.deck_overrides(problem, n, dims, mb, **kw)
        pin = load_deck(deck, ov)
        o = akref.Sim(**pu.oracle_kwargs(pin))
        o.initialize()
        init = pu.oracle_arrays(o, bool(o.params.is_mhd))
        for _ in range(cycles):
            o.step()
        fin = pu.oracle_arrays(o, bool(o.params.is_mhd))
        out = {"cycles": cycles, "time": o.time, "dt": o.dt, "deck": deck, "overrides": "\n".join(ov)}
        out.update({"init_" + k: v for k, v in init.items()})
        out.update({"final_" + k: v for k, v in fin.items()})
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: v.shape for k, v in fin.items()})


if __name__ == "__main__":
    main()
