"""CellPath: a path kept as an int32 cell array that behaves like the reference's
list of (r, c) tuples (materialised lazily)."""
import numpy as np


class CellPath:
    __slots__ = ("cells", "C", "_lst")

    def __init__(self, cells, C):
        self.cells = np.asarray(cells, np.int32)
        self.C = int(C)
        self._lst = None

    def tolist(self):
        if self._lst is None:
            C = self.C
            self._lst = [(int(x) // C, int(x) % C) for x in self.cells]
        return self._lst

    def __len__(self):
        return int(self.cells.size)

    def __bool__(self):
        return self.cells.size > 0

    def __iter__(self):
        return iter(self.tolist())

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self.tolist()[i]
        x = int(self.cells[i])
        return (x // self.C, x % self.C)

    def __eq__(self, other):
        if isinstance(other, CellPath):
            return self.C == other.C and np.array_equal(self.cells, other.cells)
        return self.tolist() == list(other)

    def __repr__(self):
        return f"CellPath({self.tolist()!r})"


def cells_of(path, C):
    """list of (r,c) / CellPath / array of cells -> int32 cell array."""
    if isinstance(path, CellPath):
        return path.cells
    if isinstance(path, np.ndarray) and path.ndim == 1:
        return path.astype(np.int32)
    return np.array([int(r) * C + int(c) for r, c in path], np.int32)
