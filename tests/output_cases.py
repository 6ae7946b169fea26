"""Shared by the output-format tests and tests/golden/make_output_fixtures.py: two small decks
with <outputN> blocks, and a runner that executes them through the command-line entry
(python -m athenak_amd -i deck -d dir), i.e. Outputs + Driver.Initialize/Execute/Finalize."""
import os

SOD_DECK = """
<job>
basename = Sod
<mesh>
nghost = 2
nx1 = 64
x1min = -0.5
x1max = 0.5
ix1_bc = outflow
ox1_bc = outflow
nx2 = 1
x2min = -0.5
x2max = 0.5
ix2_bc = periodic
ox2_bc = periodic
nx3 = 1
x3min = -0.5
x3max = 0.5
ix3_bc = periodic
ox3_bc = periodic
<meshblock>
nx1 = 32
nx2 = 1
nx3 = 1
<time>
evolution = dynamic
integrator = rk2
cfl_number = 0.8
nlim = -1
tlim = 0.1
<hydro>
eos = ideal
reconstruct = plm
rsolver = hllc
gamma = 1.4
fused_stage = FUSED
<problem>
pgen_name = shock_tube
shock_dir = 1
xshock = 0.0
dl = 1.0
pl = 1.0
ul = 0.0
vl = 0.0
wl = 0.0
dr = 0.125
pr = 0.1
ur = 0.0
vr = 0.0
wr = 0.0
<output1>
file_type = tab
variable = hydro_w
data_format = %12.5e
dt = 0.05
slice_x2 = 0.0
slice_x3 = 0.0
<output2>
file_type = hst
dt = 0.025
<output3>
file_type = bin
variable = hydro_u
dt = 0.1
"""

OT_DECK = """
<job>
basename = OrszagTang
<mesh>
nghost = 2
nx1 = 16
x1min = -0.5
x1max = 0.5
ix1_bc = periodic
ox1_bc = periodic
nx2 = 16
x2min = -0.5
x2max = 0.5
ix2_bc = periodic
ox2_bc = periodic
nx3 = 8
x3min = -0.5
x3max = 0.5
ix3_bc = periodic
ox3_bc = periodic
<meshblock>
nx1 = 8
nx2 = 8
nx3 = 8
<time>
evolution = dynamic
integrator = rk2
cfl_number = 0.3
nlim = 6
tlim = 1.0
<mhd>
eos = ideal
reconstruct = plm
rsolver = hlld
gamma = 1.666666667
fused_stage = FUSED
<problem>
pgen_name = orszag_tang
<output1>
file_type = hst
dcycle = 2
data_format = %20.12e
<output2>
file_type = bin
variable = mhd_bcc
dcycle = 6
<output3>
file_type = tab
variable = mhd_u
slice_x2 = 0.1
slice_x3 = -0.2
dcycle = 6
"""

LWAVE_ARGS = ["mesh/nx1=32", "mesh/nx2=1", "mesh/nx3=1", "meshblock/nx1=16", "meshblock/nx2=1",
              "meshblock/nx3=1", "mesh/nghost=3", "time/cfl_number=0.4", "time/tlim=1.0",
              "problem/along_x1=true", "problem/amp=1.0e-6", "problem/wave_flag=0"]


def run_case(name, workdir, fused):
    """runs one case in workdir through athenak_amd.__main__.main; returns sorted relative
    paths of the files it wrote"""
    from athenak_amd.__main__ import main
    os.makedirs(workdir, exist_ok=True)
    here = os.getcwd()
    try:
        if name in ("sod", "ot"):
            deck = os.path.join(workdir, name + ".athinput")
            text = (SOD_DECK if name == "sod" else OT_DECK).replace("FUSED", "true" if fused else "false")
            with open(deck, "w") as f:
                f.write(text)
            rc = main(["-i", deck, "-d", workdir])
        else:
            blk = "hydro" if name == "lwave_hydro" else "mhd"
            rc = main(["-i", "linear_wave_%s.athinput" % blk, "-d", workdir] + LWAVE_ARGS +
                      ["%s/fused_stage=%s" % (blk, "true" if fused else "false")])
        assert rc == 0
    finally:
        os.chdir(here)
    out = []
    for root, _, files in os.walk(workdir):
        for fn in files:
            if not fn.endswith(".athinput"):
                out.append(os.path.relpath(os.path.join(root, fn), workdir))
    return sorted(out)
