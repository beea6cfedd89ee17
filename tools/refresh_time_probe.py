import os, sys, time
sys.path.insert(0, "/root/repo")
import cProfile, pstats, io
import nupgcm_amd as npg
from nupgcm_amd import workloads
from nupgcm_amd.inversion import build_A_inversion
arch = npg.GPU(0)
m = workloads.channel_basin_model(arch, h=0.01, levels=2, itmax=0)
npg.run(m, n_steps=2)
sol = m.inversion.solver
fe = m.evolution.fe
ep = m.forcings.eddy_param
arch.ctx.sync()
for rep in range(2):
    t0 = time.perf_counter()
    fe.update_nu_eddy(ep.N2min, m.params.alpha, m.params.N2, m.b_vec)
    arch.ctx.sync(); t1 = time.perf_counter()
    build_A_inversion(arch, m.fe_data, m.params, None, A=sol.A)
    arch.ctx.sync(); t2 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable()
    sol.P.refresh(sol.A, m)
    arch.ctx.sync(); pr.disable(); t3 = time.perf_counter()
    print(f"rep {rep}: update_nu_eddy {1e3*(t1-t0):.1f} ms, re-assemble A (+repack) {1e3*(t2-t1):.1f} ms, P.refresh {1e3*(t3-t2):.1f} ms", flush=True)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14); print(s.getvalue()[:2500])
