"""Cell values of the reference's env.py (env.py:4-7) and grid generators.

The reference ships only literal grids (largest 256x256); the 128/512/1024
grids named in BASELINE.json are synthesised here from a recipe.
"""
import hashlib

import numpy as np

FREE_SPACE = 0
OBSTACLE = 1
START_NODE_VAL = 2
TARGET_NODE_VAL = 3


def mark(grid01, start, target):
    """Copy of an occupancy grid with the start/target markers set."""
    g = np.array(grid01, dtype=np.int64)
    g[g > 1] = FREE_SPACE
    g[tuple(start)] = START_NODE_VAL
    g[tuple(target)] = TARGET_NODE_VAL
    return g


def upsample(grid, k, start=None, target=None):
    """k x nearest-neighbour upsample of the obstacle mask (np.kron), S=(0,0), T=(R-1,C-1) by default.
    Preserves connectivity under the corner-cut rule."""
    occ = np.kron((np.asarray(grid) == OBSTACLE).astype(np.int64), np.ones((k, k), np.int64))
    R, C = occ.shape
    return mark(occ, start or (0, 0), target or (R - 1, C - 1))


def random_blocks(R, C, density=0.2, seed=0, block=(2, 6), start=None, target=None):
    """Seeded random rectangular blocks until `density` of the cells are obstacles; the
    start/target corners and a 2-cell margin around them stay free."""
    rng = np.random.default_rng(seed)
    occ = np.zeros((R, C), np.int64)
    s = start or (0, 0)
    t = target or (R - 1, C - 1)
    goal = int(density * R * C)
    guard = 0
    while occ.sum() < goal and guard < 100000:
        guard += 1
        h, w = rng.integers(block[0], block[1] + 1, 2)
        r, c = rng.integers(0, R - h + 1), rng.integers(0, C - w + 1)
        occ[r:r + h, c:c + w] = 1
    for (r, c) in (s, t):
        occ[max(0, r - 2):r + 3, max(0, c - 2):c + 3] = 0
    return mark(occ, s, t)


def grid_from_image(image, size=None, obstacle_below=0.5, invert=False, start=None, target=None):
    """An occupancy grid from an image, the way the reference's `grid_map_from_image_data*` literals (env.py:46-114) were
    made: dark pixels are obstacles.  `image`: a 2-D / 3-D array (grey, RGB or RGBA; uint8 or float), or a file name --
    `.npy`, binary / ASCII PGM or PPM are read with numpy alone, anything else through matplotlib.image.imread if
    matplotlib is importable.  `size` = (R, C) resamples by block-wise minimum for downscaling (an obstacle anywhere in a
    block keeps the cell blocked) or nearest neighbour for upscaling.  Returns the marked int64 grid (S = (0,0), T =
    (R-1,C-1) by default; both cells are forced free)."""
    if isinstance(image, str):
        image = _read_image(image)
    a = np.asarray(image)
    if a.ndim == 3:
        a = a[..., :3].astype(np.float64).mean(axis=2)
    a = a.astype(np.float64)
    if a.max() > 1.0:
        a = a / 255.0
    if invert:
        a = 1.0 - a
    if size is not None:
        R, C = int(size[0]), int(size[1])
        ri = (np.arange(R + 1) * a.shape[0]) // R
        ci = (np.arange(C + 1) * a.shape[1]) // C
        if R <= a.shape[0] and C <= a.shape[1]:
            a = np.array([[a[ri[i]:max(ri[i + 1], ri[i] + 1), ci[j]:max(ci[j + 1], ci[j] + 1)].min() for j in range(C)] for i in range(R)])
        else:
            a = a[np.minimum(ri[:-1], a.shape[0] - 1)][:, np.minimum(ci[:-1], a.shape[1] - 1)]
    occ = (a < obstacle_below).astype(np.int64)
    R, C = occ.shape
    s = start or (0, 0)
    t = target or (R - 1, C - 1)
    occ[tuple(s)] = 0
    occ[tuple(t)] = 0
    return mark(occ, s, t)


def _read_image(path):
    if path.endswith(".npy"):
        return np.load(path, allow_pickle=False)
    with open(path, "rb") as f:
        head = f.read(2)
        if head in (b"P2", b"P3", b"P5", b"P6"):
            data = head + f.read()
            toks, pos = [], 2
            while len(toks) < 3:                                   # width, height, maxval (comments allowed)
                while data[pos:pos + 1].isspace():
                    pos += 1
                if data[pos:pos + 1] == b"#":
                    pos = data.index(b"\n", pos) + 1
                    continue
                end = pos
                while not data[end:end + 1].isspace():
                    end += 1
                toks.append(int(data[pos:end])); pos = end
            w, h_, mx = toks
            ch = 3 if head in (b"P3", b"P6") else 1
            if head in (b"P5", b"P6"):
                arr = np.frombuffer(data[pos + 1:], np.uint8 if mx < 256 else ">u2", count=w * h_ * ch)
            else:
                arr = np.array(data[pos:].split(), np.int64)[: w * h_ * ch]
            arr = arr.reshape((h_, w, ch) if ch == 3 else (h_, w)).astype(np.float64) / mx
            return arr
    try:
        import matplotlib.image as mpimg
    except Exception as e:                                         # pragma: no cover
        raise ValueError(f"grid_from_image: cannot read {path!r} without matplotlib ({e})")
    return mpimg.imread(path)


def grid_hash(grid):
    g = np.ascontiguousarray(np.asarray(grid), np.uint8)
    return hashlib.sha256(bytes(g.shape[0].to_bytes(4, "little")) + bytes(g.shape[1].to_bytes(4, "little")) + g.tobytes()).hexdigest()


def find_marker(grid, val, who):
    """np.argwhere(grid == val)[0] with the reference's ValueError (astar.py:17-22 etc.)."""
    found = np.argwhere(np.asarray(grid) == val)
    if not found.size > 0:
        what = "Start" if val == START_NODE_VAL else "Target"
        raise ValueError(f"{who}: {what} node not found in grid." if who in ("AStar", "MPA") else f"{who}: {what} node not found.")
    return (int(found[0][0]), int(found[0][1]))


def g256():
    """The 256x256 benchmark map G256: occupancy of the reference's
    grid_map_from_image_data5 (env.py:114), shipped bit-packed as data; S=(0,0), T=(255,255)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "g256.npz"), allow_pickle=False)
    R, C = (int(v) for v in z["shape"])
    occ = np.unpackbits(z["bits"])[: R * C].reshape(R, C).astype(np.int64)
    sr, sc, tr, tc = (int(v) for v in z["st"])
    return mark(occ, (sr, sc), (tr, tc))


def bench_grid(size):
    """G256 / G512 / G1024 (np.kron upsample of G256's obstacle mask) or a seeded random-blocks G128."""
    if size == 256:
        return g256()
    if size in (512, 1024):
        return upsample(g256(), size // 256)
    if size == 128:
        return random_blocks(128, 128, 0.2, seed=128, block=(3, 8))   # 78 % of MAACO ants succeed (oracle-checked)
    raise ValueError("bench grids: 128, 256, 512, 1024")
