"""pathfit: MI355X-native population-fitness engine behind the solver classes of
dvnam1605/MAACO-path-planing (MAACO, MPA, GASolver, PSOSolver, AStarSolver)."""
from .rng import AgentRandom  # noqa: F401
