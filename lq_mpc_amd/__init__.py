"""lq_mpc_amd -- MI355X-native batched LQ-MPC box-QP solver behind the reference's class API.

Hot path of lcrekko/lq_mpc (LQ_MPC_Controller.solve / LQ_MPC_Simulator.simulate and the batch
loops that drive them) as hand-written HIP kernels for gfx950; see DESIGN.md.
"""
from .mpc import BatchSolver, LQ_MPC_Controller, LQ_MPC_Simulator, box_from_Fu, default_solver  # noqa: F401
from ._lib import LqmpcError, KERNEL_AUTO, KERNEL_GENERIC, KERNEL_SPECIALIZED, KERNEL_WORKGROUP  # noqa: F401

__version__ = "0.2.0"
