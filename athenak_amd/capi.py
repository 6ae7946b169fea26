"""ctypes binding of the C ABI in include/akmi.h (athenak_amd/lib/libakmi.so).

PyTorch is used here only as plumbing: device memory (tensor.data_ptr()) and HIP streams
(torch.cuda.current_stream().cuda_stream).  There is NO CPU fallback: if the HIP library is
missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AKMI_LIB: developer override used by tools/kbench.py to A/B kernel variants
LIB_PATH = os.environ.get("AKMI_LIB") or os.path.join(_HERE, "lib", "libakmi.so")

RECON = {"dc": 0, "plm": 1, "ppm4": 2, "ppmx": 3, "wenoz": 4, "teno": 5}
RSOLVER = {"llf": 0, "hlle": 1, "hllc": 2, "hlld": 3, "roe": 4, "advect": 5}
BC = {"block": -1, "periodic": 0, "outflow": 1, "reflect": 2, "user": 3, "inflow": 4, "diode": 5,
      "vacuum": 6}

COMPLETE, INCOMPLETE, FAIL = 0, 1, -1


class Pack(C.Structure):
    """struct akmi_pack"""
    _fields_ = [("nmb", C.c_int), ("nvar", C.c_int), ("nx1", C.c_int), ("nx2", C.c_int),
                ("nx3", C.c_int), ("ng", C.c_int), ("dx", C.c_void_p),
                ("gamma", C.c_double), ("dfloor", C.c_double), ("pfloor", C.c_double),
                ("tfloor", C.c_double), ("sfloor", C.c_double), ("sigma_max", C.c_double),
                ("iso_cs", C.c_double), ("is_ideal", C.c_int)]


PHASE_SWEEPS, PHASE_EMF_CT, PHASE_C2P, PHASE_ALL = 1, 2, 4, 7     # AKMI_PHASE_* of include/akmi.h
SMALL_PACK_CELLS = 375000       # AKMI_SMALL_PACK_CELLS of include/akmi.h (tests/test_capi_symbols.py compares them)

# every symbol include/akmi.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "akmi_last_error", "akmi_version", "akmi_build_flags", "akmi_copy_cons", "akmi_rk4_copy_cons", "akmi_hydro_fluxes_fofc", "akmi_hydro_fofc", "akmi_mhd_fluxes_fofc", "akmi_mhd_fofc", "akmi_kinematic_newdt", "akmi_ambipolar_emfs", "akmi_ambipolar_fluxes", "akmi_resistive_newdt", "akmi_restrict_cc", "akmi_restrict_fc", "akmi_restrict_cc_masked", "akmi_restrict_fc_masked", "akmi_restrict_flux_cc", "akmi_restrict_emf", "akmi_prim2cons", "akmi_prolong_cc", "akmi_prolong_fc_shared", "akmi_prolong_fc_internal", "akmi_hydro_bcs_inflow", "akmi_bfield_bcs_inflow", "akmi_hydro_bcs_dirs", "akmi_bfield_bcs_dirs", "akmi_viscous_fluxes", "akmi_heat_fluxes", "akmi_conduction_newdt", "akmi_resistive_emfs", "akmi_resistive_fluxes", "akmi_hydro_fluxes", "akmi_rk_update", "akmi_rk_update_oop", "akmi_mhd_ct_oop",
    "akmi_hydro_c2p", "akmi_hydro_newdt", "akmi_mhd_fluxes", "akmi_mhd_corner_e", "akmi_mhd_ct",
    "akmi_mhd_c2p", "akmi_mhd_newdt", "akmi_bvals_cc_local", "akmi_bvals_cc_local_bcs", "akmi_bvals_fc_local_bcs", "akmi_bvals_cc_pack",
    "akmi_bvals_cc_unpack", "akmi_bvals_cc_segsize", "akmi_bvals_fc_local", "akmi_bvals_fc_pack",
    "akmi_bvals_fc_unpack", "akmi_bvals_fc_segsize", "akmi_hydro_bcs", "akmi_bfield_bcs",
    "akmi_stage_workspace_bytes", "akmi_hydro_stage_update", "akmi_mhd_stage_update",
    "akmi_hydro_c2p_newdt", "akmi_mhd_c2p_newdt", "akmi_calib_copy", "akmi_hydro_stage_fused", "akmi_mhd_stage_fused",
    "akmi_hydro_stage_phase", "akmi_mhd_stage_phase", "akmi_hydro_stage_phase_dt", "akmi_mhd_stage_phase_dt", "akmi_history_sums",
    "akmi_hydro_c2p_shell", "akmi_mhd_c2p_shell", "akmi_sim_create", "akmi_sim_initialize",
    "akmi_sim_execute", "akmi_sim_profile", "akmi_sim_profile_read", "akmi_sim_destroy", "akmi_sim_time", "akmi_sim_dt", "akmi_sim_tlim",
    "akmi_sim_ncycle", "akmi_sim_nmb", "akmi_sim_array", "akmi_sim_lloc", "akmi_sim_gids", "akmi_sim_nmb_thisrank",
    "akmi_comm_unique_id", "akmi_comm_init_rccl", "akmi_comm_init_env", "akmi_comm_init_callbacks", "akmi_hydro_stage_fused_dt", "akmi_mhd_stage_fused_dt", "akmi_comm_finalize", "akmi_comm_allreduce_min",
    "akmi_hydro_stage_w_eligible", "akmi_hydro_stage_w", "akmi_hydro_ghost_uw", "akmi_comm_rank", "akmi_comm_nranks", "akmi_comm_profile", "akmi_comm_profile_read", "akmi_host_exchange_plan",
    "akmi_smr_exchange_cc", "akmi_smr_exchange_fc", "akmi_smr_fill_coarse_cc", "akmi_smr_fill_coarse_fc",
    "akmi_smr_prolong_cc", "akmi_smr_prolong_fc", "akmi_smr_c2p_coarse", "akmi_smr_p2c_fine", "akmi_smr_build_lists", "akmi_smr_flux_cc", "akmi_smr_emf_exchange", "akmi_smr_pack_cc", "akmi_smr_unpack_cc", "akmi_smr_pack_fc",
    "akmi_selftest_fp64",
    "akmi_smr_fc_map", "akmi_smr_fc_copy", "akmi_smr_cc_map", "akmi_smr_cc_copy",
    "akmi_smr_unpack_fc", "akmi_smr_pack_flux_cc", "akmi_smr_unpack_flux_cc", "akmi_smr_pack_emf", "akmi_smr_unpack_emf",
]

_LIB = None


class Smr(C.Structure):
    """struct akmi_smr (include/akmi.h)"""
    _fields_ = [("nnghbr", C.c_int), ("multilevel", C.c_int), ("nghbr", C.c_void_p),
                ("mblev", C.c_void_p), ("cc_tab", C.c_void_p), ("fc_tab", C.c_void_p),
                ("ndat", C.c_void_p), ("slot_ox", C.c_void_p), ("layout", C.c_void_p),
                ("soff", C.c_void_p), ("roff", C.c_void_p), ("direct_same", C.c_int), ("needs_coarse", C.c_void_p),
                ("lists", C.c_void_p), ("list_cnt", C.c_int*6)]


class AkmiError(RuntimeError):
    pass


def lib():
    """Load libakmi.so (fails loudly when it has not been built)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise AkmiError(
                "HIP library %s not found: run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.akmi_last_error.restype = C.c_char_p
        L.akmi_bvals_cc_segsize.restype = C.c_longlong
        L.akmi_bvals_fc_segsize.restype = C.c_longlong
        L.akmi_stage_workspace_bytes.restype = C.c_longlong
        L.akmi_smr_fc_map.restype = C.c_longlong
        L.akmi_smr_cc_map.restype = C.c_longlong
        L.akmi_sim_create.restype = C.c_void_p
        L.akmi_sim_array.restype = C.c_void_p
        L.akmi_sim_lloc.restype = C.POINTER(C.c_int)
        L.akmi_host_exchange_plan.restype = C.c_longlong
        L.akmi_host_exchange_plan.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                              C.c_longlong]
        for f in ("akmi_sim_time", "akmi_sim_dt", "akmi_sim_tlim"):
            getattr(L, f).restype = C.c_double
        _LIB = L
    return _LIB


def _p(t):
    """device pointer of a torch tensor (None -> NULL)"""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


DEVICE = "cuda"      # device of all field tensors; CPU-only unit tests of the host logic
                     # override it together with _LIB (see tests/cpu_backend.py)


def _stream():
    if DEVICE != "cuda":
        return None
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc < 0:
        raise AkmiError("%s failed: %s" % (what, lib().akmi_last_error().decode()))
    return rc


def d(x):
    return C.c_double(float(x))
