#!/usr/bin/env python3
"""Regenerate tests/golden/* from the reference tree's DATA files (run in the build container only).

Reads (never executes) /root/reference:
  * meshes/*.msh                 -> mesh_*.npz      (arrays parsed by nupgcm_amd.gmsh_io.read_msh)
  * test/data/<state>.jld2       -> state_*.npz     (u, p, b Float64 vectors in native Gridap free-DoF order, t)
  * test/data/A_bowl_mixing_2D.jld2 -> A_bowl_mixing_2D.npz (CSC arrays m,n,colptr,rowval,nzval + iperm)

The .jld2 files are HDF5; `h5dump -b` extracts plain datasets, and a 40-line C program against libhdf5 follows the
object references JLD2 hides the SparseMatrixCSC arrays behind.  Only data is copied - no reference source text.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from nupgcm_amd import gmsh_io  # noqa: E402

REF = "/root/reference"
H5DUMP = "/opt/conda/bin/h5dump"

C_READER = r"""
#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef struct { int64_t m, n; hobj_ref_t colptr, rowval, nzval; } csc_t;
static void dump(hid_t file, hobj_ref_t *r, hid_t memtype, size_t esz, const char *out) {
    hid_t d = H5Rdereference2(file, H5P_DEFAULT, H5R_OBJECT, r);
    hid_t s = H5Dget_space(d);
    hssize_t n = H5Sget_simple_extent_npoints(s);
    void *buf = malloc((size_t)n * esz);
    H5Dread(d, memtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf);
    FILE *f = fopen(out, "wb"); fwrite(buf, esz, (size_t)n, f); fclose(f);
    free(buf); H5Sclose(s); H5Dclose(d);
}
int main(int argc, char **argv) {
    hid_t file = H5Fopen(argv[1], H5F_ACC_RDONLY, H5P_DEFAULT);
    hid_t d = H5Dopen2(file, "A_inversion", H5P_DEFAULT);
    hid_t t = H5Tcreate(H5T_COMPOUND, sizeof(csc_t));
    H5Tinsert(t, "m", HOFFSET(csc_t, m), H5T_NATIVE_INT64);
    H5Tinsert(t, "n", HOFFSET(csc_t, n), H5T_NATIVE_INT64);
    H5Tinsert(t, "colptr", HOFFSET(csc_t, colptr), H5T_STD_REF_OBJ);
    H5Tinsert(t, "rowval", HOFFSET(csc_t, rowval), H5T_STD_REF_OBJ);
    H5Tinsert(t, "nzval", HOFFSET(csc_t, nzval), H5T_STD_REF_OBJ);
    csc_t a;
    if (H5Dread(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, &a) < 0) return 1;
    printf("%lld %lld\n", (long long)a.m, (long long)a.n);
    char p[4096];
    snprintf(p, sizeof p, "%s/colptr.bin", argv[2]); dump(file, &a.colptr, H5T_NATIVE_INT64, 8, p);
    snprintf(p, sizeof p, "%s/rowval.bin", argv[2]); dump(file, &a.rowval, H5T_NATIVE_INT64, 8, p);
    snprintf(p, sizeof p, "%s/nzval.bin", argv[2]);  dump(file, &a.nzval, H5T_NATIVE_DOUBLE, 8, p);
    return 0;
}
"""


def h5_dataset(path, name, dtype):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "d.bin")
        subprocess.run([H5DUMP, "-d", "/" + name, "-b", "LE", "-o", out, path], check=True,
                       stdout=subprocess.DEVNULL)
        return np.fromfile(out, dtype=dtype)


def main():
    for msh, out in [("bowl3D_1.000000e-01_5.000000e-01", "mesh_bowl3D_h0.1"),
                     ("bowl3D_8.000000e-02_5.000000e-01", "mesh_bowl3D_h0.08"),
                     ("bowl2D_1.000000e-01_5.000000e-01", "mesh_bowl2D_h0.1")]:
        m = gmsh_io.read_msh(f"{REF}/meshes/{msh}.msh")
        gmsh_io.save_npz(m, os.path.join(HERE, out + ".npz"))
        print(out, "nodes", len(m.coords), "cells", len(m.cells), "facets", len(m.facets), "ridges", len(m.ridges),
              m.phys_names)

    for st in ["bowl_mixing_3D", "bowl_mixing_2D", "bowl_diri", "bowl_wind", "bowl_surface_flux"]:
        p = f"{REF}/test/data/{st}.jld2"
        d = {k: h5_dataset(p, k, np.float64) for k in ("u", "p", "b", "t")}
        np.savez_compressed(os.path.join(HERE, f"state_{st}.npz"), **d)
        print(st, {k: v.shape for k, v in d.items()}, "t=", d["t"])

    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "r.c")
        open(src, "w").write(C_READER)
        exe = os.path.join(td, "r")
        subprocess.run(["gcc", "-O1", "-I/opt/conda/include", src, "-o", exe, "-L/opt/conda/lib", "-lhdf5",
                        "-Wl,-rpath,/opt/conda/lib"], check=True)
        p = f"{REF}/test/data/A_bowl_mixing_2D.jld2"
        mn = subprocess.run([exe, p, td], check=True, capture_output=True, text=True).stdout.split()
        colptr = np.fromfile(os.path.join(td, "colptr.bin"), dtype=np.int64)
        rowval = np.fromfile(os.path.join(td, "rowval.bin"), dtype=np.int64)
        nzval = np.fromfile(os.path.join(td, "nzval.bin"), dtype=np.float64)
        iperm = h5_dataset(p, "iperm", np.int64)
        np.savez_compressed(os.path.join(HERE, "A_bowl_mixing_2D.npz"), m=np.int64(mn[0]), n=np.int64(mn[1]),
                            colptr=colptr, rowval=rowval, nzval=nzval, iperm=iperm)
        print("A_2D", mn, colptr.shape, rowval.shape, nzval.shape, iperm[:5])


if __name__ == "__main__":
    main()
