"""ctypes binding of libnupgcm_hip.so (include/nupgcm_hip.h) - the same symbols ext/nuPGCMHIPExt.jl `ccall`s.

There is no CPU fallback for GPU(): if the shared library is missing this module raises, and creating a context without a
gfx950 device raises `DeviceError` (NPG_ENODEV).  The reference's CPU() architecture is a different library behind the same names
(libnupgcm_host.so, csrc_host/), selected explicitly by creating a CPU() context - see select()."""
from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnupgcm_hip.so")
# NPG_AB_LIB: ANOTHER build of the library for same-box A/B timing (tools only).  It must live outside the product directory (under
# tools/ or gpurun_out/) and every process that loads it says so on stderr - a run can never pick up a stale binary silently.
_AB = os.environ.get("NPG_AB_LIB")
if _AB:
    _ab = os.path.abspath(_AB)
    _root = os.path.abspath(os.path.join(_HERE, ".."))
    if not any(_ab.startswith(os.path.join(_root, d) + os.sep) for d in ("tools", "gpurun_out")):
        raise ImportError(f"NPG_AB_LIB={_AB}: an A/B build must live under tools/ or gpurun_out/, not beside the product library")
    import sys as _sys
    print(f"[npg] NPG_AB_LIB: loading the A/B build {_ab} instead of {LIB_PATH}", file=_sys.stderr)
    LIB_PATH = _ab
HEADER_PATH = os.path.join(_HERE, "..", "include", "nupgcm_hip.h")


class DeviceError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libnupgcm_hip error {code}: {msg}")
        self.code = code


class SolveStats(C.Structure):
    _fields_ = [("solved", C.c_int32), ("niter", C.c_int32), ("npass", C.c_int32), ("status", C.c_int32),
                ("nreorth", C.c_int32), ("nflagged", C.c_int32), ("rnorm0", C.c_double), ("rnorm", C.c_double),
                ("seconds", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class FeDesc(C.Structure):
    _fields_ = [("ncell", C.c_int64), ("nq", C.c_int32), ("nloc_b", C.c_int32),
                ("grad_lambda", C.c_void_p), ("wdet", C.c_void_p), ("qw", C.c_void_p), ("N2", C.c_void_p),
                ("dN2", C.c_void_p), ("Nb", C.c_void_p), ("dNb", C.c_void_p), ("N1", C.c_void_p),
                ("cell_u", C.c_void_p), ("cell_p", C.c_void_p), ("cell_b", C.c_void_p),
                ("u_diri", C.c_void_p), ("n_u_diri", C.c_int64), ("b_diri", C.c_void_p), ("n_b_diri", C.c_int64),
                ("n_inv", C.c_int64), ("n_b", C.c_int64)]


def declared_symbols(header=HEADER_PATH):
    """Every function name declared in include/nupgcm_hip.h (used by the CPU test that the library exports them all)."""
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(npg_[a-z0-9_]+)\s*\(", txt)))


_lib = None
# One process runs on ONE architecture.  "hip" (default): libnupgcm_hip.so, the GPU() architecture.  "host": libnupgcm_host.so, the
# same C ABI (the subset Model(CPU(), ...) drives) in plain C++ / OpenMP for the reference's CPU() architecture - chosen explicitly
# by creating a CPU() context (architectures.CPU.ctx -> select("host")), never a fallback: GPU() without the HIP library or without a
# gfx950 device still raises.
HOST_LIB_PATH = os.path.join(_HERE, "libnupgcm_host.so")
_kind = "hip"


def select(kind):
    """choose the library behind lib(): "hip" or "host".  Switching drops the loaded library object, so the caller
    (architectures.py) refuses it while handles of the other architecture are alive."""
    global _lib, _kind
    if kind not in ("hip", "host"):
        raise ValueError(kind)
    if kind != _kind:
        _lib, _kind = None, kind


def kind():
    return _kind


def lib():
    global _lib
    if _lib is None:
        path = LIB_PATH if _kind == "hip" else HOST_LIB_PATH
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              f"(make -C nupgcm_amd/{'csrc' if _kind == 'hip' else 'csrc_host'}). "
                              + ("nupgcm_amd has no CPU fallback for GPU()." if _kind == "hip" else ""))
        _lib = C.CDLL(path)
        _declare(_lib, partial=_kind == "host")
    return _lib


def _declare(L, partial=False):
    P, I64, I32, D, VP = C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_void_p
    PP = C.POINTER(C.c_void_p)
    sig = {
        "npg_ctx_create": [C.c_int, PP], "npg_ctx_destroy": [P], "npg_ctx_sync": [P],
        "npg_mem_status": [P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)],
        "npg_device_name": [P, C.c_char_p, C.c_size_t],
        "npg_timer_start": [P], "npg_timer_stop": [P, C.POINTER(D)],
        "npg_vec_create": [P, I64, PP], "npg_vec_view": [P, I64, I64, PP], "npg_vec_destroy": [P], "npg_vec_upload": [P, VP], "npg_vec_download": [P, VP],
        "npg_vec_upload_perm": [P, VP, VP], "npg_vec_download_perm": [P, VP, VP], "npg_vec_fill": [P, D],
        "npg_vec_copy": [P, P], "npg_vec_axpby": [P, D, P, D], "npg_vec_dot": [P, P, C.POINTER(D)],
        "npg_vec_nrm2": [P, C.POINTER(D)], "npg_vec_maxabs": [P, C.POINTER(D), C.POINTER(C.c_int)],
        "npg_vec_is_constant": [P, C.POINTER(D), C.POINTER(C.c_int)],
        "npg_vec_lincomb": [P, C.c_int, C.POINTER(D), PP], "npg_vec_mul": [P, P, P],
        "npg_csr_create_from_csc": [P, I64, I64, VP, VP, VP, C.c_int, PP],
        "npg_csr_create": [P, I64, I64, VP, VP, VP, PP], "npg_csr_destroy": [P],
        "npg_csr_shape": [P, C.POINTER(I64), C.POINTER(I64), C.POINTER(I64)],
        "npg_csr_to_csc": [P, VP, VP, VP], "npg_csr_download": [P, VP, VP, VP], "npg_csr_clone": [P, PP],
        "npg_csr_zero_values": [P], "npg_csr_pair_xy": [P, I64, D, C.POINTER(C.c_int)],
        "npg_csr_block_nodes": [P, I64, I64, D, C.POINTER(C.c_int)],
        "npg_csr_storage": [P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
        "npg_csr_coupling_records": [P, C.POINTER(C.c_int64)], "npg_csr_pack_nodes": [P, I64, I64, C.POINTER(C.c_int)], "npg_csr_set_ghost_nodes": [P, I64, VP, VP], "npg_csr_spmv_bytes": [P, C.POINTER(C.c_int64)],
        "npg_spmv_gather32": [P, P, P, C.c_int, C.c_int],
        "npg_csr_block_nodes_dofs": [P, C.c_void_p, C.c_void_p, D, C.POINTER(C.c_int)],
        "npg_csr_window_info": [P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)], "npg_csr_set_lanes": [P, C.c_int], "npg_csr_product": [P, P, P], "npg_csr_values_to_vec": [P, P], "npg_csr_values_from_vec": [P, P, P], "npg_csr_line_block_inverse": [P, P, P, P], "npg_csr_line_schur": [P] * 12, "npg_precond_mg_set_scaled_gradient": [P, C.c_int, P], "npg_csr_combine": [P, D, P, D, P, P], "npg_csr_inv_diag": [P, P],
        "npg_csr_node_block_inverse": [P, P, I64, I64], "npg_csr_triple_product": [P, P, P, P],
        "npg_index_create": [P, I64, VP, I64, PP], "npg_index_destroy": [P], "npg_csr_gather_values": [P, P, P],
        "npg_spmv": [P, P, P, D, D],
        "npg_gmres_create": [P, I64, C.c_int, PP], "npg_gmres_destroy": [P],
        "npg_gmres_solve": [P, P, C.c_int, D, P, P, P, D, D, I64, D, C.POINTER(SolveStats)],
        "npg_gmres_set_profile": [P, C.c_int], "npg_gmres_set_split": [P, C.c_int], "npg_gmres_set_basis": [P, C.c_int], "npg_gmres_set_gather": [P, C.c_int], "npg_gmres_get_profile": [P, C.POINTER(D), C.POINTER(I64)],
        "npg_cg_create": [P, I64, PP], "npg_cg_destroy": [P],
        "npg_cg_solve": [P, P, C.c_int, D, P, P, P, D, D, I64, C.POINTER(SolveStats)],
        "npg_precond_create": [P, C.c_int, C.c_int, PP], "npg_precond_destroy": [P],
        "npg_precond_blockdiag_set": [P, C.c_int, I64, P, P, I64, D, D], "npg_precond_blockdiag_set_ilu0": [P, C.c_int, P],
        "npg_ilu0_create": [P, P, C.POINTER(P)], "npg_ilu0_destroy": [P], "npg_ilu0_refactor": [P, P], "npg_ilu0_apply": [P, P, P],
        "npg_ilu0_info": [P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)], "npg_ilu0_factors": [P, P],
        "npg_cg_ilu0_solve": [P, P, P, P, D, D, I64, C.POINTER(SolveStats)],
        "npg_precond_mg_set_level": [P, C.c_int, P, I64, P, P, P, P, P, P],
        "npg_precond_mg_update_level": [P, C.c_int, P, P, P, P, P],
        "npg_precond_mg_set_params": [P, D, D, C.c_int, C.c_int, C.c_int, C.c_int],
        "npg_precond_mg_set_cycle": [P, C.c_int], "npg_precond_mg_set_mixed": [P, C.c_int], "npg_precond_dense_set": [P, P, C.c_int],
        "npg_precond_mg_set_coarse_dense": [P, C.c_int],
        "npg_precond_apply": [P, P, P], "npg_precond_counters": [P, C.POINTER(I64), C.POINTER(I64)],
        "npg_precond_cycle_bytes": [P, C.POINTER(I64)],
        "npg_fgmres_create": [P, I64, C.c_int, PP], "npg_fgmres_destroy": [P], "npg_fgmres_set_halo": [P, P],
        "npg_precond_mg_set_level_dist": [P, C.c_int, P, I64, P, P, P, P, P, P, P, P, P],
        "npg_precond_mg_set_transfer_dist": [P, C.c_int, P, P, P, P],
        "npg_fgmres_solve": [P, P, P, P, P, D, D, D, I64, C.POINTER(SolveStats)],
        "npg_fe_set_precision": [P, C.c_int], "npg_fe_get_precision": [P],
        "npg_fe_create": [P, C.POINTER(FeDesc), PP], "npg_fe_destroy": [P], "npg_fe_set_coeff": [P, C.c_char_p, VP],
        "npg_fe_evolution_rhs": [P, C.c_int, D, D, D, P, P, P, P, P, P, P, P, P, P],
        "npg_fe_advection_rhs": [P, C.c_int, D, D, P, P, P, P, P],
        "npg_fe_assemble_matrix": [P, C.c_int, D, C.c_int, P, P], "npg_fe_assemble_rhs_diff": [P, D, P],
        "npg_fe_update_kappa_convection": [P, VP, D, D, D, D, P],
        "npg_fe_update_nu_eddy": [P, D, D, D, D, D, P], "npg_fe_restrict_coeff": [P, P, C.c_char_p], "npg_fe_coeff_cell_mean": [P, C.c_char_p, P], "npg_fe_cfl_ratio": [P, VP, D, P, C.POINTER(D)],
        "npg_comm_unique_id": [VP], "npg_comm_init": [P, VP, C.c_int, C.c_int],
        "npg_comm_allreduce_sum": [P, C.POINTER(D), C.c_int], "npg_comm_info": [P, C.c_char_p, C.c_size_t], "npg_comm_disable_peer": [P], "npg_comm_allreduce_vec": [P, P],
        "npg_comm_allgather_segments": [P, P, C.c_int, VP, VP, VP, VP, P],
        "npg_halo_create": [P, I64, I64, C.c_int, VP, VP, VP, VP, PP], "npg_halo_destroy": [P],
        "npg_halo_exchange": [P, P], "npg_gmres_set_halo": [P, P], "npg_gmres_set_dist_options": [P, C.c_int, C.c_int], "npg_cg_set_halo": [P, P],
    }
    for name, args in sig.items():
        if (_AB or partial) and not hasattr(L, name):
            continue            # (an older build timed beside the current one: entry points it lacks stay unbound)
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int
    L.npg_last_error.restype = C.c_char_p
    L.npg_last_error.argtypes = []
    L.npg_ctx_stream.restype = C.c_void_p
    L.npg_ctx_stream.argtypes = [P]
    L.npg_vec_len.restype = I64
    L.npg_vec_len.argtypes = [P]
    L.npg_gmres_history.restype = I64
    L.npg_gmres_history.argtypes = [P, VP, I64]
    L.npg_cg_history.restype = I64
    L.npg_cg_history.argtypes = [P, VP, I64]
    if hasattr(L, "npg_fgmres_history"):           # (the host library has no flexible GMRES: general preconditioners are device work)
        L.npg_fgmres_history.restype = I64
        L.npg_fgmres_history.argtypes = [P, VP, I64]


def check(rc):
    if rc != 0:
        raise DeviceError(rc, lib().npg_last_error().decode(errors="replace"))


def ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def as_i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def as_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


NPG_PRECOND_NONE, NPG_PRECOND_SCALAR, NPG_PRECOND_DIAG = 0, 1, 2
NPG_PC_BLOCKDIAG, NPG_PC_MG, NPG_PC_DENSE = 1, 2, 3
NPG_BDF1, NPG_BDF2 = 1, 2
NPG_FE_FP64, NPG_FE_FP32 = 0, 1
NPG_MAT_M, NPG_MAT_KH, NPG_MAT_KV, NPG_MAT_A, NPG_MAT_B = 1, 2, 3, 4, 5
