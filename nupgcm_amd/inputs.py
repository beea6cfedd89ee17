"""Parameters, Forcings, surface boundary conditions and the two parameterisation structs - mirrors
/root/reference/src/inputs.jl:3-189.  Functions of x take an array (..., 3) and return an array (...)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any


@dataclass
class Parameters:
    """src/inputs.jl:3-15"""
    eps: float      # Ekman number
    alpha: float    # aspect ratio
    mu_rho: float   # Prandtl times Burger number
    N2: float       # background stratification
    f: Any          # Coriolis parameter, function of x (or a number)
    H: Any          # depth, function of x

    def __repr__(self):
        return (f"Parameters\n├── ε  = {self.eps:1.1e}\n├── α  = {self.alpha:1.1e}\n├── μϱ = {self.mu_rho:1.1e}\n"
                f"├── N² = {self.N2:1.1e}\n├── f: {self.f}\n└── H: {self.H}")


@dataclass
class SurfaceDirichletBC:
    """src/inputs.jl:35-37"""
    value: Any


@dataclass
class SurfaceFluxBC:
    """src/inputs.jl:48-50"""
    flux: Any


@dataclass
class ConvectionParameterization:
    """src/inputs.jl:64-85"""
    kappa_c: float = 0.0
    N2min: float = 0.0
    is_on: bool = True


@dataclass
class EddyParameterization:
    """src/inputs.jl:95-114"""
    f: Any = 0.0
    N2min: float = 0.0
    is_on: bool = True


@dataclass
class Forcings:
    """Forcings(nu, kappa_h, kappa_v, tau_x, tau_y, b_surface_bc; conv_param, eddy_param) - src/inputs.jl:141-189"""
    nu: Any
    kappa_h: Any
    kappa_v: Any
    tau_x: Any
    tau_y: Any
    b_surface_bc: Any
    conv_param: ConvectionParameterization = field(default_factory=lambda: ConvectionParameterization(0, 0, False))
    eddy_param: EddyParameterization = field(default_factory=lambda: EddyParameterization(0, 0, False))
