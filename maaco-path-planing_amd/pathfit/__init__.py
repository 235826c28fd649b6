"""pathfit: MI355X-native population-fitness engine behind the solver classes of
dvnam1605/MAACO-path-planing.  The classes keep the reference's constructor and
solve() / solve_path_planning() surface (MAACO.py:10, MPA.py:9, ga_solver.py:8,
pso.py:8, astar.py:10); the per-agent decode -> A*-stitch -> score -> update
hot path runs in hand-written HIP kernels (csrc/) behind a C-ABI
(include/pathfit.h).  No CPU fallback exists."""
from .env import FREE_SPACE, OBSTACLE, START_NODE_VAL, TARGET_NODE_VAL  # noqa: F401
from .rng import AgentRandom  # noqa: F401
from ._lib import PathfitError  # noqa: F401
from .engine import Engine, DevBuf, score_params  # noqa: F401
from .paths import CellPath  # noqa: F401
from .solvers import AStarSolver, DijkstraSolver, GASolver, PSOSolver, BasePathfinder  # noqa: F401
from .maaco import MAACO  # noqa: F401
from .mpa import MPA  # noqa: F401
