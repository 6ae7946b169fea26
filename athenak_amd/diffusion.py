"""Diffusion objects of Hydro/MHD: constant isotropic viscosity, constant thermal diffusivity,
Ohmic resistivity.  Mirror of src/diffusion/{viscosity,conduction,resistivity}.cpp: each object adds
its fluxes (EMFs) inside the Fluxes (EField) task and owns a `dtnew` that Mesh::NewTimeStep folds
into the time step (src/mesh/mesh.cpp:586-612).  The kernels are csrc/akmi_diffusion.hip.
"""
import ctypes as C

import torch

from . import capi

FLT_MAX = 3.4028234663852886e+38


def _fac(pm):
    """stability factor of an explicit diffusion step: viscosity.cpp:236-243"""
    return 1.0/6.0 if pm.three_d else (0.25 if pm.two_d else 0.5)


class _Diffusion:
    def __init__(self, phys):
        self.phys = phys
        self.pmy_pack = phys.pmy_pack
        self.L = phys.L
        self.dtnew = FLT_MAX

    def _const_dt(self, coeff):
        """min over the MeshBlocks of fac*dx^2/coeff (viscosity.cpp:244-250, resistivity.cpp:299-309)"""
        pm = self.pmy_pack.pmesh
        fac = _fac(pm)
        dt = FLT_MAX
        for dx in self.pmy_pack.pmb.dx:
            dt = min(dt, fac*(dx[0]*dx[0])/coeff)
            if pm.multi_d:
                dt = min(dt, fac*(dx[1]*dx[1])/coeff)
            if pm.three_d:
                dt = min(dt, fac*(dx[2]*dx[2])/coeff)
        return float(dt)


class Viscosity(_Diffusion):
    """src/diffusion/viscosity.cpp"""

    def __init__(self, block, phys, pin):
        super().__init__(phys)
        self.nu_iso = pin.GetOrAddReal(block, "nu_iso", 0.0)
        self.nu_aniso = pin.GetOrAddReal(block, "nu_aniso", 0.0)
        if self.nu_aniso != 0.0:
            raise RuntimeError("### FATAL ERROR <%s>/nu_aniso: anisotropic viscosity is a no-op in "
                               "the reference and not on this path" % block)

    def AddViscousFluxes(self, w0, flx, face_shaped):
        if self.nu_iso != 0.0:                                   # viscosity.cpp:52-54
            capi.check(self.L.akmi_viscous_fluxes(
                C.byref(self.phys.pack_c), C.c_double(self.nu_iso), capi._p(w0), capi._p(flx.x1f),
                capi._p(flx.x2f), capi._p(flx.x3f), face_shaped, capi._stream()), "viscous_fluxes")

    def NewTimeStep(self):
        self.dtnew = self._const_dt(self.nu_iso) if self.nu_iso != 0.0 else FLT_MAX


class Conduction(_Diffusion):
    """src/diffusion/conduction.cpp (constant alpha_iso)"""

    def __init__(self, block, phys, pin):
        super().__init__(phys)
        self.alpha_iso = pin.GetOrAddReal(block, "alpha_iso", 0.0)
        if pin.GetOrAddReal(block, "alpha_aniso", 0.0) != 0.0 or \
                pin.GetOrAddBoolean(block, "alpha_spitzer", False):
            raise RuntimeError("### FATAL ERROR <%s>: only constant isotropic thermal conduction "
                               "(alpha_iso) is on this path" % block)
        if not phys.peos.eos_data.is_ideal:                      # hydro.cpp:89-95
            raise RuntimeError("### FATAL ERROR Thermal conduction requires ideal gas EOS")
        self.dtmin = torch.zeros(1, dtype=torch.float64, device=phys.device)

    def AddHeatFluxes(self, w0, flx, face_shaped):
        if self.alpha_iso != 0.0:                                # conduction.cpp:89-91
            capi.check(self.L.akmi_heat_fluxes(
                C.byref(self.phys.pack_c), C.c_double(self.alpha_iso), capi._p(w0), capi._p(flx.x1f),
                capi._p(flx.x2f), capi._p(flx.x3f), face_shaped, capi._stream()), "heat_fluxes")

    def NewTimeStep(self, w0):
        if self.alpha_iso == 0.0:
            self.dtnew = FLT_MAX*_fac(self.pmy_pack.pmesh)      # SQR(dx)/0 = inf never wins the min
            return
        capi.check(self.L.akmi_conduction_newdt(
            C.byref(self.phys.pack_c), C.c_double(self.alpha_iso), capi._p(w0), capi._p(self.dtmin),
            capi._stream()), "conduction_newdt")
        self.dtnew = float(self.dtmin.item())*_fac(self.pmy_pack.pmesh)


class Resistivity(_Diffusion):
    """src/diffusion/resistivity.cpp (constant Ohmic eta_ohm)"""

    def __init__(self, phys, pin):
        super().__init__(phys)
        self.eta_ohm = pin.GetOrAddReal("mhd", "eta_ohm", 0.0)
        self.eta_ad = pin.GetOrAddReal("mhd", "eta_ad", 0.0)
        self.dtmin = torch.zeros(1, dtype=torch.float64, device=phys.device)

    def AddResistiveEMFs(self, b0, efld):
        if self.eta_ohm != 0.0:                                  # resistivity.cpp:49-51
            capi.check(self.L.akmi_resistive_emfs(
                C.byref(self.phys.pack_c), C.c_double(self.eta_ohm), capi._p(b0.x1f), capi._p(b0.x2f),
                capi._p(b0.x3f), capi._p(efld.x1e), capi._p(efld.x2e), capi._p(efld.x3e),
                capi._stream()), "resistive_emfs")
        if self.eta_ad != 0.0:                                   # resistivity.cpp:52-54
            capi.check(self.L.akmi_ambipolar_emfs(
                C.byref(self.phys.pack_c), C.c_double(self.eta_ad), capi._p(self.phys.bcc0),
                capi._p(b0.x1f), capi._p(b0.x2f), capi._p(b0.x3f), capi._p(efld.x1e), capi._p(efld.x2e),
                capi._p(efld.x3e), capi._stream()), "ambipolar_emfs")

    def AddResistiveFluxes(self, b0, flx):
        if self.eta_ohm != 0.0:                                  # resistivity.cpp:64-66
            capi.check(self.L.akmi_resistive_fluxes(
                C.byref(self.phys.pack_c), C.c_double(self.eta_ohm), capi._p(b0.x1f), capi._p(b0.x2f),
                capi._p(b0.x3f), capi._p(flx.x1f), capi._p(flx.x2f), capi._p(flx.x3f),
                capi._stream()), "resistive_fluxes")
        if self.eta_ad != 0.0:                                   # resistivity.cpp:67-69
            capi.check(self.L.akmi_ambipolar_fluxes(
                C.byref(self.phys.pack_c), C.c_double(self.eta_ad), capi._p(self.phys.bcc0),
                capi._p(b0.x1f), capi._p(b0.x2f), capi._p(b0.x3f), capi._p(flx.x1f), capi._p(flx.x2f),
                capi._p(flx.x3f), capi._stream()), "ambipolar_fluxes")

    def NewTimeStep(self):
        if self.eta_ad == 0.0:                                   # resistivity.cpp:298-311
            self.dtnew = self._const_dt(self.eta_ohm) if self.eta_ohm > 0.0 else FLT_MAX
            return
        capi.check(self.L.akmi_resistive_newdt(                  # resistivity.cpp:313-345
            C.byref(self.phys.pack_c), C.c_double(self.eta_ohm), C.c_double(self.eta_ad),
            capi._p(self.phys.bcc0), capi._p(self.dtmin), capi._stream()), "resistive_newdt")
        self.dtnew = float(self.dtmin.item())*_fac(self.pmy_pack.pmesh)


def make_diffusion(phys, pin, blk):
    """objects exist only when their parameters are in the input file (hydro.cpp:77-98,
    mhd.cpp:104-130); any of them moves the run to the task-granular kernels"""
    phys.pvisc = phys.pcond = phys.presist = None
    if pin.DoesParameterExist(blk, "nu_iso") or pin.DoesParameterExist(blk, "nu_aniso"):
        phys.pvisc = Viscosity(blk, phys, pin)
    if pin.DoesParameterExist(blk, "alpha_iso") or pin.DoesParameterExist(blk, "alpha_aniso") or \
            pin.DoesParameterExist(blk, "alpha_spitzer"):
        phys.pcond = Conduction(blk, phys, pin)
    if blk == "mhd" and (pin.DoesParameterExist("mhd", "eta_ohm") or
                         pin.DoesParameterExist("mhd", "eta_ad")):
        phys.presist = Resistivity(phys, pin)
    return phys.pvisc is not None or phys.pcond is not None or phys.presist is not None
