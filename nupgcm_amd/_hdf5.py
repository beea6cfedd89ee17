"""Minimal HDF5 access through libhdf5 + ctypes (no h5py in the image): flat files of Float64 datasets, written the way
JLD2's `jldsave(ofile; u, p, b, t)` lays them out (/root/reference/src/IO.jl:8) - a 512-byte user block carrying the JLD2
header line, superblock version 2, one little-endian Float64 dataset per name (1-D, or scalar) - so that the reference's
`jldopen(ifile)["u"]` (src/IO.jl:12-23) reads what `save_state` wrote and vice versa.  Host-side, off the hot path."""
from __future__ import annotations

import ctypes as C
import ctypes.util
import os

import numpy as np

_JLD2_HEADER = b"HDF5-based Julia Data Format, version 0.1.1\x00 (nupgcm_amd, libhdf5 writer)\x00"
_USERBLOCK = 512
_lib = None


def lib():
    global _lib
    if _lib is None:
        names = [os.environ.get("NPG_HDF5_LIB"), ctypes.util.find_library("hdf5"), "/opt/conda/lib/libhdf5.so",
                 "libhdf5.so", "libhdf5_serial.so"]
        err = None
        for nm in names:
            if not nm:
                continue
            try:
                _lib = C.CDLL(nm)
                break
            except OSError as e:
                err = e
        if _lib is None:
            raise ImportError(f"libhdf5 not found (set NPG_HDF5_LIB); last error: {err}")
        L = _lib
        hid, herr = C.c_int64, C.c_int
        L.H5open.restype = herr
        for fn, res, args in (("H5Pcreate", hid, [hid]), ("H5Pset_userblock", herr, [hid, C.c_uint64]),
                              ("H5Pset_libver_bounds", herr, [hid, C.c_int, C.c_int]), ("H5Pclose", herr, [hid]),
                              ("H5Fcreate", hid, [C.c_char_p, C.c_uint, hid, hid]), ("H5Fopen", hid, [C.c_char_p, C.c_uint, hid]),
                              ("H5Fclose", herr, [hid]), ("H5Screate_simple", hid, [C.c_int, C.c_void_p, C.c_void_p]),
                              ("H5Screate", hid, [C.c_int]), ("H5Sclose", herr, [hid]),
                              ("H5Sget_simple_extent_npoints", C.c_int64, [hid]),
                              ("H5Dcreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]),
                              ("H5Dopen2", hid, [hid, C.c_char_p, hid]), ("H5Dget_space", hid, [hid]),
                              ("H5Dwrite", herr, [hid, hid, hid, hid, hid, C.c_void_p]),
                              ("H5Dread", herr, [hid, hid, hid, hid, hid, C.c_void_p]), ("H5Dclose", herr, [hid]),
                              ("H5Lexists", C.c_int, [hid, C.c_char_p, hid]), ("H5Eset_auto2", herr, [hid, C.c_void_p, C.c_void_p])):
            f = getattr(L, fn)
            f.restype, f.argtypes = res, args
        L.H5open()
        L.H5Eset_auto2(0, None, None)           # errors are reported through return codes, not printed
    return _lib


def _g(name):
    return C.c_int64.in_dll(lib(), name).value


def write_flat(path, arrays):
    """arrays: name -> 1-D float array or scalar"""
    L = lib()
    fcpl = L.H5Pcreate(_g("H5P_CLS_FILE_CREATE_ID_g"))
    fapl = L.H5Pcreate(_g("H5P_CLS_FILE_ACCESS_ID_g"))
    if fcpl < 0 or fapl < 0 or L.H5Pset_userblock(fcpl, _USERBLOCK) < 0 or L.H5Pset_libver_bounds(fapl, 1, 1) < 0:
        raise OSError("HDF5: could not set up the file creation properties")
    f = L.H5Fcreate(os.fsencode(path), 2, fcpl, fapl)                  # H5F_ACC_TRUNC
    if f < 0:
        raise OSError(f"HDF5: cannot create {path}")
    f64, nat = _g("H5T_IEEE_F64LE_g"), _g("H5T_NATIVE_DOUBLE_g")
    try:
        for name, val in arrays.items():
            scalar = np.ndim(val) == 0
            a = np.ascontiguousarray(val, dtype=np.float64)
            if scalar:
                sp = L.H5Screate(0)                                    # H5S_SCALAR
            else:
                dims = (C.c_uint64 * 1)(a.size)
                sp = L.H5Screate_simple(1, dims, None)
            d = L.H5Dcreate2(f, name.encode(), f64, sp, 0, 0, 0)
            if d < 0 or L.H5Dwrite(d, nat, 0, 0, 0, a.ctypes.data_as(C.c_void_p)) < 0:
                raise OSError(f"HDF5: cannot write dataset {name}")
            L.H5Dclose(d)
            L.H5Sclose(sp)
    finally:
        L.H5Fclose(f)
        L.H5Pclose(fcpl)
        L.H5Pclose(fapl)
    with open(path, "r+b") as fh:                                       # the user block is ours: JLD2's header line
        fh.write(_JLD2_HEADER.ljust(_USERBLOCK, b"\x00")[:_USERBLOCK])
    return path


def read_flat(path, names):
    """name -> float64 array (shape (n,) or () for a scalar dataset); works on the reference's own .jld2 state files"""
    L = lib()
    f = L.H5Fopen(os.fsencode(path), 0, 0)
    if f < 0:
        raise OSError(f"HDF5: cannot open {path}")
    out = {}
    nat = _g("H5T_NATIVE_DOUBLE_g")
    try:
        for name in names:
            if L.H5Lexists(f, name.encode(), 0) <= 0:
                continue
            d = L.H5Dopen2(f, name.encode(), 0)
            sp = L.H5Dget_space(d)
            n = L.H5Sget_simple_extent_npoints(sp)
            a = np.empty(max(int(n), 1))
            if L.H5Dread(d, nat, 0, 0, 0, a.ctypes.data_as(C.c_void_p)) < 0:
                raise OSError(f"HDF5: cannot read dataset {name} of {path}")
            out[name] = a
            L.H5Sclose(sp)
            L.H5Dclose(d)
    finally:
        L.H5Fclose(f)
    return out
