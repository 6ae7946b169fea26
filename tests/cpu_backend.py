"""TEST INFRASTRUCTURE: run the product's HOST logic (Mesh, TaskList, Driver, bvals, multi-rank
exchange) on CPU tensors by standing the oracle's akref_* functions in for the akmi_* C ABI.
Used only by the not-gpu tests of the distributed path (gloo); the shipped package never
does this -- without libakmi.so it raises (tests/test_capi_symbols.py)."""
import ctypes as C

from oracle import akref


class OracleAsAkmi:
    """object with akmi_* attributes that forward to akref_* (dropping the stream argument)"""

    def __init__(self):
        self.R = akref.lib()

    def akmi_last_error(self):
        return b"(cpu test backend)"

    def akmi_version(self):
        return 100

    def __getattr__(self, name):
        if not name.startswith("akmi_"):
            raise AttributeError(name)
        # handle-based oracle functions own the plain names of these: their ABI twins end in _t
        twin = {"akmi_smr_fill_coarse_cc", "akmi_smr_fill_coarse_fc", "akmi_smr_prolong_cc",
                "akmi_smr_prolong_fc", "akmi_smr_flux_cc", "akmi_smr_c2p_coarse", "akmi_smr_p2c_fine"}
        fn = getattr(self.R, "akref_" + name[5:] + ("_t" if name in twin else ""))
        if name.endswith("segsize") or name.endswith("_doubles"):
            fn.restype = C.c_longlong
            return fn

        def call(*args):
            return fn(*args[:-1])          # last argument is the HIP stream
        return call


def install():
    from athenak_amd import capi
    capi._LIB = OracleAsAkmi()
    capi.DEVICE = "cpu"


def uninstall():
    from athenak_amd import capi
    capi._LIB = None
    capi.DEVICE = "cuda"
