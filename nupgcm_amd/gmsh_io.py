"""Gmsh 4.1 ASCII reader -> plain arrays.

Stands in for `GmshDiscreteModel(ifile)` at /root/reference/src/meshes.jl:30 (GridapGmsh 0.7.4, not vendored
in the reference).  Only what the hot path's setup needs is kept: node coordinates, the top-dimensional cells in file
order, the lower-dimensional boundary elements, and for every vertex / boundary element the *set of physical names*
of the Gmsh entity it belongs to, stored as a bitmask over `phys_names` (names are merged across dimensions the way
GridapGmsh's face labelling does: "bottom" of dims 0,1,2 is one tag).

The parsed model can be written to / read from a compact `.npz` (`save_npz` / `load_npz`), which is how the committed
mesh fixtures under tests/golden/ travel to the GPU box (the reference tree does not).
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field

import numpy as np

_NODES_PER_TYPE = {15: 1, 1: 2, 2: 3, 4: 4}   # point, line, triangle, tetrahedron
_DIM_OF_TYPE = {15: 0, 1: 1, 2: 2, 4: 3}


@dataclass
class GmshModel:
    dim: int                     # topological dimension of the cells (2 or 3)
    coords: np.ndarray           # (nn, 3) float64
    node_phys: np.ndarray        # (nn,) uint32 bitmask over phys_names: names of the entity that owns the node
    cells: np.ndarray            # (nc, dim+1) int64, 0-based node ids, raw Gmsh order, file order
    facets: np.ndarray           # (nf, dim) int64: (dim-1)-dimensional boundary elements
    facets_phys: np.ndarray      # (nf,) uint32 bitmask
    ridges: np.ndarray           # (nr, dim-1) int64: (dim-2)-dimensional elements (lines in 3-D, points in 2-D)
    ridges_phys: np.ndarray      # (nr,) uint32 bitmask
    phys_names: list = field(default_factory=list)
    # periodic meshes (`gmsh.model.mesh.setPeriodic`, /root/reference/meshes/channel_basin.jl:103-108): periodic[n] = master
    # node of n (n itself when it has none).  GridapGmsh glues the vertices of paired nodes in the grid topology; cells keep
    # their own node coordinates.  None = not periodic.
    periodic: np.ndarray = None
    # set by refine.refine_once: for every cell of the PARENT mesh the diagonal (0, 1, 2) its inner octahedron was cut along
    child_variant: np.ndarray = None

    def tag_mask(self, names) -> int:
        m = 0
        for nm in names:
            m |= 1 << self.phys_names.index(nm)
        return m


def read_msh(path: str) -> GmshModel:
    with open(path, "r") as fh:
        lines = fh.read().split("\n")
    pos = 0

    def seek(section):
        nonlocal pos
        while lines[pos].strip() != section:
            pos += 1
        pos += 1

    seek("$MeshFormat")
    ver = lines[pos].split()
    if not ver[0].startswith("4.1") or ver[1] != "0":
        raise ValueError(f"only Gmsh 4.1 ASCII is supported, got {lines[pos]!r}")

    # physical names: (dim, tag) -> name ; names merged across dims
    seek("$PhysicalNames")
    nphys = int(lines[pos]); pos += 1
    names: list[str] = []
    name_of: dict[tuple[int, int], int] = {}
    for _ in range(nphys):
        d, t, nm = lines[pos].split(maxsplit=2); pos += 1
        nm = nm.strip().strip('"')
        if nm not in names:
            names.append(nm)
        name_of[(int(d), int(t))] = names.index(nm)

    # entities: (dim, tag) -> bitmask of physical names
    seek("$Entities")
    counts = [int(x) for x in lines[pos].split()]; pos += 1
    ent_mask: dict[tuple[int, int], int] = {}
    for d, cnt in enumerate(counts):
        for _ in range(cnt):
            tok = lines[pos].split(); pos += 1
            tag = int(tok[0])
            off = 4 if d == 0 else 7
            nph = int(tok[off])
            m = 0
            for k in range(nph):
                pt = abs(int(tok[off + 1 + k]))
                if (d, pt) in name_of:
                    m |= 1 << name_of[(d, pt)]
            ent_mask[(d, tag)] = m

    seek("$Nodes")
    nblocks, nn, _mn, mx = (int(x) for x in lines[pos].split()); pos += 1
    if mx != nn:
        raise ValueError("non-contiguous node tags are not supported")
    coords = np.zeros((nn, 3))
    node_phys = np.zeros(nn, dtype=np.uint32)
    for _ in range(nblocks):
        ed, et, _par, nb = (int(x) for x in lines[pos].split()); pos += 1
        tags = [int(lines[pos + k]) for k in range(nb)]; pos += nb
        for k in range(nb):
            coords[tags[k] - 1] = [float(x) for x in lines[pos + k].split()[:3]]
            node_phys[tags[k] - 1] = ent_mask.get((ed, et), 0)
        pos += nb

    seek("$Elements")
    nblocks, _ne, _mn, _mx = (int(x) for x in lines[pos].split()); pos += 1
    by_dim: dict[int, list] = {0: [], 1: [], 2: [], 3: []}
    phys_by_dim: dict[int, list] = {0: [], 1: [], 2: [], 3: []}
    for _ in range(nblocks):
        ed, et, ety, nb = (int(x) for x in lines[pos].split()); pos += 1
        if ety not in _NODES_PER_TYPE:
            raise ValueError(f"unsupported Gmsh element type {ety}")
        k = _NODES_PER_TYPE[ety]
        m = ent_mask.get((ed, et), 0)
        for j in range(nb):
            tok = lines[pos + j].split()
            by_dim[_DIM_OF_TYPE[ety]].append([int(x) - 1 for x in tok[1:1 + k]])
            phys_by_dim[_DIM_OF_TYPE[ety]].append(m)
        pos += nb

    periodic = None
    if "$Periodic" in lines[pos:]:
        seek("$Periodic")
        nlinks = int(lines[pos]); pos += 1
        periodic = np.arange(nn, dtype=np.int64)
        for _ in range(nlinks):
            pos += 1                                   # entityDim entityTag entityTagMaster
            pos += 1                                   # numAffine value ...
            ncorr = int(lines[pos]); pos += 1
            for k in range(ncorr):
                a, b = lines[pos + k].split()
                periodic[int(a) - 1] = int(b) - 1
            pos += ncorr
        while True:                                    # masters of masters (corner nodes of several links)
            nxt = periodic[periodic]
            if np.array_equal(nxt, periodic):
                break
            periodic = nxt

    dim = 3 if by_dim[3] else 2

    def arr(d, k):
        a = np.asarray(by_dim[d], dtype=np.int64).reshape(-1, k)
        return a, np.asarray(phys_by_dim[d], dtype=np.uint32)

    cells, _ = arr(dim, dim + 1)
    facets, facets_phys = arr(dim - 1, dim)
    ridges, ridges_phys = arr(dim - 2, dim - 1)
    return GmshModel(dim, coords, node_phys, cells, facets, facets_phys, ridges, ridges_phys, names, periodic=periodic)


def save_npz(model: GmshModel, path: str) -> None:
    np.savez_compressed(
        path, dim=np.int64(model.dim), coords=model.coords, node_phys=model.node_phys, cells=model.cells,
        facets=model.facets, facets_phys=model.facets_phys, ridges=model.ridges, ridges_phys=model.ridges_phys,
        phys_names=np.frombuffer(json.dumps(model.phys_names).encode(), dtype=np.uint8),
        **({} if model.periodic is None else {"periodic": np.asarray(model.periodic, dtype=np.int64)}))


def load_npz(path: str) -> GmshModel:
    z = np.load(path)
    names = json.loads(bytes(z["phys_names"]).decode())
    return GmshModel(int(z["dim"]), z["coords"], z["node_phys"], z["cells"], z["facets"], z["facets_phys"],
                     z["ridges"], z["ridges_phys"], names, periodic=z["periodic"] if "periodic" in z.files else None)


def load_model(path: str) -> GmshModel:
    return load_npz(path) if path.endswith(".npz") else read_msh(path)
