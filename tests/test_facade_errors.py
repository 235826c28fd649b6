"""Constructor error conventions of the reference (astar.py:19-20, MAACO.py:35-38,
MPA.py:36-39, ga_solver.py:19-20, pso.py:19-20): raised before any GPU work."""
import numpy as np
import pytest

import pathfit


@pytest.mark.parametrize("make,msg", [
    (lambda g: pathfit.AStarSolver(g), "AStar: Start node not found in grid."),
    (lambda g: pathfit.MAACO(g, 4, 2, 1, 7, .1, 2.5, 1, .9, .2, .9, .5), "MAACO: Start node not found."),
    (lambda g: pathfit.MPA(g, 4, 2), "MPA: Start node not found in grid."),
    (lambda g: pathfit.GASolver(g, 2, 4, 3, .1, .8), "GA: Start node not found."),
    (lambda g: pathfit.PSOSolver(g, 2, 4, 3, .7, 1.5, 1.5), "PSO: Start node not found."),
])
def test_missing_start(make, msg):
    g = np.zeros((6, 6), int); g[5, 5] = 3
    with pytest.raises(ValueError) as ei:
        make(g)
    assert str(ei.value) == msg


def test_missing_target():
    g = np.zeros((6, 6), int); g[0, 0] = 2
    with pytest.raises(ValueError) as ei:
        pathfit.MPA(g, 4, 2)
    assert str(ei.value) == "MPA: Target node not found in grid."
    with pytest.raises(ValueError) as ei:
        pathfit.GASolver(g, 2, 4, 3, .1, .8)
    assert str(ei.value) == "GA: Target node not found."


def test_cellpath_behaves_like_list_of_tuples():
    p = pathfit.CellPath(np.array([0, 21, 42]), 20)
    assert len(p) == 3 and p[0] == (0, 0) and p[-1] == (2, 2) and list(p) == [(0, 0), (1, 1), (2, 2)]
    assert p == [(0, 0), (1, 1), (2, 2)] and bool(p) and not pathfit.CellPath(np.zeros(0, np.int32), 20)
