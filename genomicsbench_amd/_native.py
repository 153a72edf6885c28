"""ctypes binding of libgbx.so (include/gbx.h).  Fails loudly when the library is absent."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GBX_LIB") or os.path.join(_HERE, "libgbx.so")      # GBX_LIB: a tuning build of the same library

GBX_OK = 0
GBX_ERR_ARG = -1
GBX_ERR_NO_DEVICE = -2
GBX_ERR_HIP = -3
GBX_ERR_NOMEM = -4
GBX_ERR_UNSUPPORTED = -5


class GbxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("gbx error %d: %s" % (code, msg))
        self.code = code


class BswParams(C.Structure):
    _fields_ = [("o_del", C.c_int32), ("e_del", C.c_int32), ("o_ins", C.c_int32), ("e_ins", C.c_int32),
                ("zdrop", C.c_int32), ("end_bonus", C.c_int32), ("w", C.c_int32),
                ("mat", C.c_int8 * 25), ("pad_", C.c_int8 * 3)]


class ChainCall(C.Structure):
    _fields_ = [("avg_qspan", C.c_float), ("max_dist_x", C.c_int32), ("max_dist_y", C.c_int32),
                ("bw", C.c_int32), ("n_segs", C.c_int32)]


CHAIN_CALL_DTYPE = np.dtype([("avg_qspan", "<f4"), ("max_dist_x", "<i4"), ("max_dist_y", "<i4"),
                             ("bw", "<i4"), ("n_segs", "<i4")])
BSW_RESULT_FIELDS = ("score", "tle", "gtle", "qle", "gscore", "max_off")
_SP_NAMES = ["idr", "idq", "id", "len1", "len2", "h0", "seqid", "regid", "score", "tle", "gtle", "qle", "gscore",
             "max_off"]
_SP_FMTS = ["<i8"] * 3 + ["<i4"] * 11
_SP_OFFS = [0, 8, 16] + [24 + 4 * k for k in range(11)]
# 68 bytes of fields, padded to 72 by the 8-byte alignment of the C struct (bandedSWA.h:91-100)
SEQPAIR_DTYPE = np.dtype({"names": _SP_NAMES, "formats": _SP_FMTS, "offsets": _SP_OFFS, "itemsize": 72})
assert SEQPAIR_DTYPE.itemsize == 72

_lib = None


def lib():
    """Returns the loaded libgbx.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libgbx.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(expected at %s)" % LIB_PATH)
        # PyTorch-ROCm bundles its own HIP/HSA runtime.  Two copies of the runtime in one process cannot
        # both own the GPU, so when torch is installed it must be loaded first; libgbx.so then binds to the
        # runtime already in the process (same SONAME).  The C++ drivers never load torch.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(L):
    vp, i64, sz, i32 = C.c_void_p, C.c_int64, C.c_size_t, C.c_int
    L.gbx_version.restype = C.c_char_p
    L.gbx_last_error.restype = C.c_char_p
    L.gbx_device_count.restype = C.c_int
    L.gbx_set_device.argtypes = [C.c_int]
    L.gbx_device_name.argtypes = [C.c_char_p, sz]
    L.gbx_host_set_devices.argtypes = [C.c_int]
    L.gbx_host_devices.argtypes = []
    L.gbx_split_by_cost.argtypes = [i64, vp, C.c_int, vp]
    L.gbx_host_prepare.argtypes = []
    L.gbx_host_release.argtypes = []
    L.gbx_host_reserve.argtypes = [sz]
    if hasattr(L, "gbx_host_combine_stats"):
        L.gbx_host_combine_stats.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.c_int]
    L.gbx_timer_create.argtypes = [C.POINTER(vp)]
    L.gbx_timer_start.argtypes = [vp, vp]
    L.gbx_timer_stop.argtypes = [vp, vp]
    L.gbx_timer_elapsed_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.gbx_timer_destroy.argtypes = [vp]
    L.gbx_timer_destroy.restype = None
    L.gbx_profile_end.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_int),
                                  C.POINTER(C.c_int)]
    L.gbx_malloc_device.argtypes = [C.POINTER(vp), sz]
    L.gbx_free_device.argtypes = [vp]
    L.gbx_memcpy_h2d.argtypes = [vp, vp, sz, vp]
    L.gbx_memcpy_d2h.argtypes = [vp, vp, sz, vp]
    L.gbx_stream_synchronize.argtypes = [vp]
    L.gbx_bsw_default_params.argtypes = [C.POINTER(BswParams)]
    L.gbx_bsw_default_params.restype = None
    L.gbx_bsw_fill_scmat.argtypes = [i32, i32, i32, C.POINTER(C.c_int8)]
    L.gbx_bsw_fill_scmat.restype = None
    L.gbx_bsw_workspace_bytes.argtypes = [i64]
    L.gbx_bsw_workspace_bytes.restype = sz
    L.gbx_bsw_extend_host.argtypes = [C.POINTER(BswParams), i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp]
    L.gbx_bsw_extend_seqpairs.argtypes = [C.POINTER(BswParams), vp, i64, vp, i64, vp, i64]
    L.gbx_bsw_extend_device.argtypes = [C.POINTER(BswParams), i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    if hasattr(L, "gbx_poa_consensus_host"):
        L.gbx_poa_default_params.argtypes = [vp]
        L.gbx_poa_default_params.restype = None
        L.gbx_poa_plan_host.argtypes = [i64, vp, vp, vp]
        L.gbx_poa_workspace_bytes.argtypes = [vp]
        L.gbx_poa_workspace_bytes.restype = sz
        L.gbx_poa_consensus_host.argtypes = [vp, i64, vp, i64, vp, vp, vp, i64, vp, vp, i64]
        L.gbx_poa_cells.argtypes = [vp, vp, C.POINTER(C.c_int64), vp]
        L.gbx_poa_consensus_device.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, i64, vp, sz, vp]
    if hasattr(L, "gbx_phmm_forward_host"):
        L.gbx_phmm_workspace_bytes.argtypes = [i64, i64, C.c_int32]
        L.gbx_phmm_workspace_bytes.restype = sz
        L.gbx_phmm_forward_host.argtypes = [i64, vp, vp, i64, vp, vp, i64, vp, vp, vp, vp, vp, i64, vp, vp, i64, vp, vp]
        L.gbx_phmm_forward_device.argtypes = [i64, vp, vp, i64] + [vp] * 10 + [C.c_int32, vp, vp, sz, vp]
    if hasattr(L, "gbx_abea_align_host"):
        L.gbx_abea_plan_host.argtypes = [i64, vp, vp, vp, vp, vp]
        L.gbx_abea_workspace_bytes.argtypes = [i64, i64, i64]
        L.gbx_abea_workspace_bytes.restype = sz
        L.gbx_abea_align_host.argtypes = [i64, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp]
        L.gbx_abea_align_device.argtypes = [i64] + [vp] * 11 + [i64, i64, vp, vp, vp, sz, vp]
        L.gbx_abea_cells.argtypes = [vp, C.POINTER(C.c_int64), vp]
    if hasattr(L, "gbx_fmi_smem_host"):
        L.gbx_fmi_default_params.argtypes = [vp, C.c_int32]
        L.gbx_fmi_default_params.restype = None
        L.gbx_fmi_index_bytes.argtypes = [i64]
        L.gbx_fmi_index_bytes.restype = sz
        L.gbx_fmi_index_build.argtypes = [vp, vp, sz, vp]
        L.gbx_fmi_workspace_bytes.argtypes = [i64, C.c_int32, C.c_int32]
        L.gbx_fmi_workspace_bytes.restype = sz
        L.gbx_fmi_smem_host.argtypes = [vp, vp, i64, vp, i64, vp, vp, vp, i64, vp, C.POINTER(C.c_int64)]
        L.gbx_fmi_smem_device.argtypes = [vp, vp, vp, i64, C.c_int32, vp, vp, vp, vp, i64, vp, vp, vp, sz, vp]
        L.gbx_fmi_extensions.argtypes = [vp, C.POINTER(C.c_int64), vp]
        L.gbx_fmi_overflow.argtypes = [vp, C.POINTER(C.c_int64), vp]
    if hasattr(L, "gbx_chain_host"):
        L.gbx_chain_job_stats.argtypes = [vp, i64, i64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), vp]
        L.gbx_chain_workspace_bytes.argtypes = [i64, i64]
        L.gbx_chain_workspace_bytes.restype = sz
        L.gbx_chain_host.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp, vp]
        L.gbx_chain_device.argtypes = [i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
        L.gbx_chain_evaluated_pairs.argtypes = [vp, C.POINTER(C.c_int64), vp]


def check(rc):
    if rc != 0:
        raise GbxError(rc, lib().gbx_last_error().decode())


def ptr(a):
    """Raw pointer of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def combine_stats(kernel, reset=False):
    """Counters of the host entries' call combiner for one kernel ("bsw", "phmm", "poa") ->
    dict(calls, device_calls, shared, largest)."""
    out = (C.c_uint64 * 4)()
    check(lib().gbx_host_combine_stats({"bsw": 1, "phmm": 3, "poa": 4}[kernel], out, 1 if reset else 0))
    return dict(calls=int(out[0]), device_calls=int(out[1]), shared=int(out[2]), largest=int(out[3]))


def device_count():
    return lib().gbx_device_count()


def profile_begin():
    check(lib().gbx_profile_begin())


def profile_end(cap=64):
    """-> {kernel name: (summed ms, launches)} for everything launched since profile_begin()."""
    names = (C.c_char_p * cap)()
    ms = (C.c_float * cap)()
    cnt = (C.c_int * cap)()
    n = C.c_int(0)
    check(lib().gbx_profile_end(cap, names, ms, cnt, C.byref(n)))
    return {names[k].decode(): (ms[k], cnt[k]) for k in range(n.value)}


class StreamTimer:
    """HIP-event timer on an explicit stream (wraps gbx_timer_*)."""

    def __init__(self):
        self._t = C.c_void_p()
        check(lib().gbx_timer_create(C.byref(self._t)))

    def start(self, stream=None):
        check(lib().gbx_timer_start(self._t, stream))

    def stop(self, stream=None):
        check(lib().gbx_timer_stop(self._t, stream))

    def elapsed_ms(self):
        ms = C.c_float()
        check(lib().gbx_timer_elapsed_ms(self._t, C.byref(ms)))
        return ms.value

    def __del__(self):
        if self._t:
            lib().gbx_timer_destroy(self._t)
            self._t = None
