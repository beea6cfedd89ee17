"""nupgcm_amd - MI355X-native hot path of nuPGCM (assembly -> Krylov inversion -> buoyancy evolution) behind the
reference's Architecture / InversionToolkit / EvolutionToolkit / invert! / evolve! surface.  GPU(): the device layer is
libnupgcm_hip.so (hand-written HIP for gfx950) - no fallback: without the library or a gfx950 device it raises.  CPU() (round 5;
BASELINE configs[0]): the same C ABI built for the host (libnupgcm_host.so, plain C++ / OpenMP) with the reference's CPU() solver
branches (sparse LU / backslash / host Krylov) - an architecture chosen explicitly, one per process, never a stand-in for GPU()."""
from .architectures import (CPU, GPU, AbstractArchitecture, DeviceCSR, DeviceILU0, DeviceVector, architecture, on_architecture,
                            print_memory_status, vector_type)
from .evolution import EvolutionToolkit, collect_evolution_LHS, evolution_parameter
from .fe import DoFHandler, FEData, Mesh, Spaces, get_n_dofs
from .inputs import (ConvectionParameterization, EddyParameterization, Forcings, Parameters, SurfaceDirichletBC,
                     SurfaceFluxBC)
from .inversion import InversionToolkit, build_A_inversion, build_B_inversion, build_b_inversion
from .io import save_checkpoint, save_state, save_vtk, set_out_dir, set_state_from_file
from .iterative_solvers import CgWorkspace, Diagonal, GmresWorkspace, IterativeSolverToolkit, MgsGmresWorkspace, iterative_solve
from .multigrid import BlockDiagonalPreconditioner, DenseInversePreconditioner, FgmresWorkspace, GeneralPreconditioner, MultigridPreconditioner
from .model import BlowUp, Model, State, evolve, invert, run, set_b, sync_flow
from .timesteppers import BDF1, BDF2, update_dt, update_t

__all__ = [n for n in dir() if not n.startswith("_")]
