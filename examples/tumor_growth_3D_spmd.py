"""
The reference's *_mpi.py scripts run unchanged under `mpirun -np N` because DOLFIN distributes the mesh
(README.md:142-183).  The counterpart here: run the SAME script under

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 examples/tumor_growth_3D_spmd.py

(one rank per GPU, halos over RCCL); without torchrun it runs on one GPU.  Every rank ends up with the global solution.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import torch.distributed as dist
from glimslib_amd.simulation import TumorGrowthBrain
from glimslib_amd import fenics_local as fenics
from _brain_like import brain_like_mesh

world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    # GLIMS_FORCE_DEVICE=<ordinal> + GLIMS_TRANSPORT=gloo: rehearsal with all ranks on one GPU (RCCL needs one GPU per rank)
    forced = os.environ.get("GLIMS_FORCE_DEVICE")
    torch.cuda.set_device(int(forced if forced is not None else os.environ["LOCAL_RANK"]))
    dist.init_process_group(backend="gloo" if forced is not None else "cpu:gloo,cuda:nccl")
rank = dist.get_rank() if world > 1 else 0


class Boundary(fenics.SubDomain):
    def inside(self, x, on_boundary):
        return on_boundary


mesh, subdomains = brain_like_mesh(int(sys.argv[1]) if len(sys.argv) > 1 else 32)
sim = TumorGrowthBrain(mesh)
sim.setup_global_parameters(subdomains=subdomains, domain_names={1: 'CSF', 3: 'WM', 2: 'GM', 4: 'Ventricles'},
                            boundaries={'boundary_all': Boundary()},
                            dirichlet_bcs={'clamped_0': {'bc_value': fenics.Constant((0.0, 0.0, 0.0)),
                                                         'named_boundary': 'boundary_all', 'subspace_id': 0}})
iv = fenics.Expression('exp(-a*pow(x[0]-x0, 2) - a*pow(x[1]-y0, 2) - a*pow(x[2]-z0,2))', degree=1, a=0.005,
                       x0=118, y0=-109, z0=72)
sim.setup_model_parameters(iv_expression={0: fenics.Constant((0., 0., 0.)), 1: iv}, sim_time=20, sim_time_step=1,
                           E_GM=3000E-6, E_WM=3000E-6, E_CSF=1000E-6, E_VENT=1000E-6, nu_GM=0.45, nu_WM=0.45,
                           nu_CSF=0.45, nu_VENT=0.3, D_GM=0.01, D_WM=0.05, rho_GM=0.05, rho_WM=0.05, coupling=0.1)
sol = sim.run(keep_nth=10, save_method=None, plot=False)
if rank == 0:
    print("ranks %d: max c %.4f, max |u| %.4e, stats %s" %
          (world, sol.components[1].max(), abs(sol.components[0]).max(), sim.solver_statistics()))
sim.close()
if world > 1:
    dist.destroy_process_group()
