"""HIP kernels (through the C-ABI) == CPU oracle == golden vectors of the
unmodified reference.  Cell-index paths and fp64 stats are compared bit for bit
(the north-star tolerance for fp32 fitness, 1e-5 relative, is met with zero
error)."""
import math

import numpy as np
import pytest

import golden_io as gio

pytestmark = pytest.mark.gpu

_eng = {}
SEQ = True        # the closed-set searches run on the sequential pop loop: pop / push counters are the reference's


@pytest.fixture(autouse=True, params=["sequential", "settle"])
def closed_set_engine(request):
    """Every test of this module runs twice: with the closed-set connector on the sequential pop loop (paths AND the
    reference's pop / push counts are compared) and with the parallel label-settling engine in front of it (pf_settle.h:
    paths and statuses are compared; its expansion counts are its own)."""
    global SEQ
    from pathfit.engine import Engine
    import golden_io
    e = eng("fig7")[0]
    SEQ = request.param == "sequential"
    e.set_option("astar_settle", 0 if SEQ else 1)
    yield
    e.set_option("astar_settle", -1)
    SEQ = True


def eng(name):
    from pathfit.engine import Engine
    import pf_oracle as po
    if name not in _eng:
        if name.startswith("up"):          # "up2:g256" -> np.kron upsample
            k, base = name[2:].split(":")
            g0, _, _ = gio.grid(base)
            g = gio.upsample(g0, int(k))
            s, t = 0, g.size - 1
        else:
            g, s, t = gio.grid(name)
        _eng[name] = (Engine(g), po.Oracle(g), s, t, g)
    return _eng[name]


def test_device_sqrt_is_correctly_rounded():
    e, _, _, _, _ = eng("fig7")
    # every dr^2+dc^2 reachable on a 1024^2 grid is < 2^21; test all n <= 2^21 + large sample up to 2*4095^2
    n1 = np.arange(0, 1 << 21, dtype=np.int64)
    rng = np.random.default_rng(0)
    n2 = rng.integers(0, 2 * 4095 ** 2, size=1 << 20, dtype=np.int64)
    for arr in (n1, n2):
        d_in, d_out = e.put(arr), e.buf(arr.size, np.float64)
        e._ck(e.L.pf_selftest_sqrt(e.h, arr.size, d_in.ptr, d_out.ptr))
        assert np.array_equal(d_out.download(), np.sqrt(arr.astype(np.float64)))


def test_device_rng_matches_cpython():
    e, _, _, _, _ = eng("fig7")
    z = gio.load("rng")
    for i, key in enumerate(z["keys"]):
        k = [int(v) for v in key]
        du, df, di = e.buf(8, np.uint64), e.buf(36, np.float64), e.buf(37, np.int64)
        e._ck(e.L.pf_selftest_rng(e.h, k[0], k[1], k[2], k[3], du.ptr, df.ptr, di.ptr))
        u, f, ii = du.download(), df.download(), di.download()
        assert np.array_equal(u, z[f"k{i}_next64"])
        assert np.array_equal(f[:8], z[f"k{i}_random"])
        assert np.array_equal(f[8:28], z[f"k{i}_normal"])
        assert np.array_equal(f[28:36], z[f"k{i}_uniform"])
        assert np.array_equal(ii[:24], z[f"k{i}_randint"])
        assert np.array_equal(ii[24:34], z[f"k{i}_choice"])
        assert [int(ii[34]), int(ii[35]), int(ii[36])] == [int(z[f"k{i}_randint_draws"]), int(z[f"k{i}_normal_draws"]),
                                                           int(z[f"k{i}_choice_draws"])]


def test_astar_golden_both_variants():
    z = gio.load("astar_cases")
    names = [str(s) for s in z["grid_names"]]
    for gid, gname in enumerate(names):
        e, o, _, _, _ = eng(gname)
        for variant in (0, 1):
            idx = [i for i in range(len(z["start"])) if z["grid_id"][i] == gid and z["variant"][i] == variant]
            avoid = [gio.csr_get(z["avoid_off"], z["avoid"], i) if z["has_avoid"][i] else None for i in idx]
            paths, st, cnt = e.astar_host(variant, z["start"][idx], z["target"][idx], avoid, want_counters=True)
            for j, i in enumerate(idx):
                want = gio.csr_get(z["path_off"], z["path"], i)
                assert st[j] != 3
                assert np.array_equal(paths[j], want), (gname, variant, i)
                if len(want) > 1:
                    assert (not SEQ and variant == 0) or cnt[j, 0] == z["pops"][i], (gname, variant, i, cnt[j], z["pops"][i])
                    # oracle cross-check of the counter definitions
                    _, ost = o.astar(int(z["start"][i]), int(z["target"][i]), avoid[j], variant)
                    assert (not SEQ and variant == 0) or (cnt[j, 1] == ost[1] == z["pushes"][i] and cnt[j, 3] == ost[4])


def test_big_cases_at_bench_sizes():
    """HIP == the UNMODIFIED reference on the bench grids themselves (tests/golden/big_cases.npz, oracle/capture_golden_big.py):
    both connectors on G512 and G1024 incl. the corner-to-corner searches, a path-prefix avoid set, the reference's heap pop / push
    counts (sequential engine); GA chained decodes + fp64 stats on G512."""
    from pathfit.engine import score_params
    z = gio.load("big_cases")
    names = [str(s) for s in z["grid_names"]]
    for gid, gname in enumerate(names):
        e, o, _, _, g = eng(f"up{int(gname[1:]) // 256}:g256")
        for variant in (0, 1):
            idx = [i for i in range(len(z["start"])) if z["grid_id"][i] == gid and z["variant"][i] == variant]
            avoid = [gio.csr_get(z["avoid_off"], z["avoid"], i) if z["has_avoid"][i] else None for i in idx]
            paths, st, cnt = e.astar_host(variant, z["start"][idx], z["target"][idx], avoid, path_cap=16 * 1024, want_counters=True)
            for j, i in enumerate(idx):
                want = gio.csr_get(z["path_off"], z["path"], i)
                assert st[j] != 3 and np.array_equal(paths[j], want), (gname, variant, i)
                if len(want) > 1 and (SEQ or variant == 1):
                    assert cnt[j, 0] == z["pops"][i] and cnt[j, 1] == z["pushes"][i], (gname, variant, i, cnt[j], z["pops"][i], z["pushes"][i])
    e = eng("up2:g256")[0]
    sp = score_params(0, True, 0.3, 0.8, 1.8, 100.0)
    for W in (3, 5):
        js = [j for j in range(len(z["dec_wp"])) if int((z["dec_wp"][j] >= 0).sum()) == W]
        wp = np.ascontiguousarray(np.asarray(z["dec_wp"], np.int32)[js, :W])
        paths, st, stats = e.decode_host(0, 512 * 512 - 1, wp_cells=wp, sp=sp, path_cap=16 * 1024)
        for k, j in enumerate(js):
            assert np.array_equal(paths[k], gio.csr_get(z["dec_path_off"], z["dec_path"], j))
            assert np.array_equal(stats[k], z["dec_stats"][j])
    # MPA._reconstruct_path_segment of the reference's own initial path at 512^2 (main.py:44-52 parameters)
    from pathfit._lib import MpaParams
    base = np.asarray(z["reb_base"], np.int32)
    seed, it = (int(v) for v in z["reb_seed_it"])
    n = len(z["reb_idx"])
    spm = score_params(1, True, 0.1, 0.8, 1.8, 100.0)
    e.mpa_setup(MpaParams(0.5, 2.0, float(z["reb_sigma"][0]), 0.2, n, 0, 512 * 512 - 1, 1, 1), spm)
    cap = 8192
    pop = np.zeros((n, cap), np.int32); pop[:, :len(base)] = base
    plen = np.full(n, len(base), np.int32)
    pstats = np.tile(np.asarray(z["reb_base_stats"], np.float64), (n, 1))
    dpop, dlen, dstats, del_ = e.put(pop), e.put(plen), e.put(pstats), e.put(base)
    oc, ol, os_, ost = e.buf((n, cap), np.int32), e.buf(n, np.int32), e.buf((n, 5), np.float64), e.buf(n, np.int32)
    d_idx, d_lv = e.put(z["reb_idx"], np.int32), e.put(z["reb_is_levy"], np.int32)
    d_sc, d_ag = e.put(z["reb_scale"], np.float64), e.put(z["reb_agent"], np.int32)
    e._ck(e.L.pf_mpa_rebuild_batch(e.h, it, seed, n, cap, dpop.ptr, dlen.ptr, dstats.ptr, del_.ptr, len(base),
                                   d_idx.ptr, d_lv.ptr, d_sc.ptr, d_ag.ptr, oc.ptr, ol.ptr, os_.ptr, ost.ptr))
    cells, lens, stats, st = oc.download(), ol.download(), os_.download(), ost.download()
    for i in range(n):
        want = gio.csr_get(z["reb_out_off"], z["reb_out"], i)
        assert st[i] != 3 and np.array_equal(cells[i, :lens[i]], want), (i, st[i])
        assert np.array_equal(stats[i], z["reb_stats"][i]), i


def test_dijkstra_golden_and_facade():
    """DijkstraSolver (dijkstra.py:32-97): HIP variant 2 == golden paths / pops / pushes of the unmodified reference; the
    facade returns the reference's 6-tuple."""
    import pathfit
    z = gio.load("dijkstra_cases")
    names = [str(s) for s in z["grid_names"]]
    for gid, gname in enumerate(names):
        e, o, _, _, g = eng(gname)
        idx = [i for i in range(len(z["start"])) if z["grid_id"][i] == gid]
        avoid = [gio.csr_get(z["avoid_off"], z["avoid"], i) if z["has_avoid"][i] else None for i in idx]
        paths, st, cnt = e.astar_host(2, z["start"][idx], z["target"][idx], avoid, want_counters=True)
        for j, i in enumerate(idx):
            want = gio.csr_get(z["path_off"], z["path"], i)
            assert st[j] != 3 and np.array_equal(paths[j], want), (gname, i)
            if len(want) > 1:
                assert not SEQ or (cnt[j, 0] == z["pops"][i] and cnt[j, 1] == z["pushes"][i]), (gname, i, cnt[j], z["pops"][i], z["pushes"][i])
        if g.shape[0] <= 20:
            C = g.shape[1]
            d = pathfit.DijkstraSolver(g, engine=e)
            for j, i in enumerate(idx[:12]):
                s_, t_ = int(z["start"][i]), int(z["target"][i])
                av = {(int(a) // C, int(a) % C) for a in avoid[j]} if avoid[j] is not None else None
                res = d.solve((s_ // C, s_ % C), (t_ // C, t_ % C), av)
                want = gio.csr_get(z["path_off"], z["path"], i)
                assert [r * C + c for r, c in res[0]] == list(want)
                ws = z["stats"][i]
                assert all((a == b) or (math.isinf(a) and math.isinf(b)) for a, b in zip(res[1:6], ws)), (gname, i, res[1:], ws)


def test_astar_random_512_vs_oracle():
    e, o, s, t, g = eng("up2:g256")
    rnd = np.random.default_rng(5)
    free = np.flatnonzero(g.reshape(-1) != 1)
    n = 48
    starts = rnd.choice(free, n); targets = rnd.choice(free, n)
    starts[0], targets[0] = s, t                       # corner to corner: 91 044 pops in the reference
    avoid = [rnd.choice(free, 200) if i % 2 else None for i in range(n)]
    for variant in (0, 1):
        paths, st, cnt = e.astar_host(variant, starts, targets, avoid, path_cap=8192, want_counters=True)
        for i in range(n):
            want, ost = o.astar(int(starts[i]), int(targets[i]), avoid[i], variant)
            assert st[i] != 3 and np.array_equal(paths[i], want), (variant, i)
            if len(want) > 1:
                assert (not SEQ and variant == 0) or cnt[i, 0] == ost[0]
        if variant == 0 and SEQ:
            assert cnt[0, 0] == 91044


def test_astar_sealed_rooms_vs_oracle():
    """Goals/starts in free rooms no move can enter (larger than the 512-cell pocket flood): the static component
    test must give the reference's answer (no path), under every move policy and with avoid sets."""
    from pathfit.engine import Engine
    import pf_oracle as po
    rnd = np.random.default_rng(9)
    g = (rnd.random((96, 96)) < 0.08).astype(np.uint8)
    g[20:62, 30] = 1; g[20:62, 71] = 1; g[20, 30:72] = 1; g[61, 30:72] = 1          # sealed 40x40 room
    g[70:90, 5:8] = 1; g[70, 5:30] = 1; g[89, 5:30] = 1; g[70:90, 29] = 1             # second room, diagonal leak at a corner
    g[89, 29] = 0; g[88, 29] = 1; g[89, 28] = 1
    e, o = Engine(g), po.Oracle(g)
    free = np.flatnonzero(g.reshape(-1) != 1)
    inside = np.array([c for c in free if 20 < c // 96 < 61 and 30 < c % 96 < 71])
    room2 = np.array([c for c in free if 70 < c // 96 < 89 and 7 < c % 96 < 29])
    n = 40
    starts = np.concatenate([rnd.choice(inside, 10), rnd.choice(free, 10), rnd.choice(room2, 10), rnd.choice(free, 10)])
    targets = np.concatenate([rnd.choice(free, 10), rnd.choice(inside, 10), rnd.choice(free, 10), rnd.choice(room2, 10)])
    avoid = [rnd.choice(free, 60) if i % 3 == 0 else None for i in range(n)]
    for variant in (0, 1):
        for restrict in (1, 0):
            paths, st, _ = e.astar_host(variant, starts, targets, avoid, path_cap=4096, restrict_corner=restrict, want_counters=True)
            fails = 0
            for i in range(n):
                o.restrict = restrict
                want, _ = o.astar(int(starts[i]), int(targets[i]), avoid[i], variant)
                assert st[i] != 3 and np.array_equal(paths[i], want), (variant, restrict, i)
                fails += len(want) == 0
            assert fails >= 10
    e.close()


def test_astar_clustered_heads_vs_oracle():
    """The pop loop takes up to seven window heads per trip and replays, per cell, what the earlier heads of the trip
    do to the records they share.  Small, nearly open maps keep consecutive pops adjacent (every trip has overlapping
    3x3 neighbourhoods), thin walls and avoid sets make cells get improved twice within a trip, and start == target /
    adjacent / unreachable pairs cover the short trips.  Every search is compared with the oracle: path, pops, pushes;
    all three connector variants, both corner policies."""
    from pathfit.engine import Engine
    import pf_oracle as po
    rnd = np.random.default_rng(11)
    maps = []
    g = np.zeros((40, 40), np.uint8); maps.append(g)
    g = (rnd.random((48, 48)) < 0.06).astype(np.uint8); maps.append(g)
    g = np.zeros((33, 57), np.uint8); g[5:28, 20] = 1; g[5, 20:40] = 1; g[16, 0:14] = 1; g[16, 15:20] = 1; maps.append(g)
    for g in maps:
        e, o = Engine(g), po.Oracle(g)
        free = np.flatnonzero(g.reshape(-1) != 1)
        n = 120
        starts = rnd.choice(free, n); targets = rnd.choice(free, n)
        targets[:6] = starts[:6]                                           # start == target
        targets[6:12] = np.clip(starts[6:12] + 1, 0, g.size - 1)           # neighbours (or an obstacle / the next row)
        avoid = [rnd.choice(free, int(rnd.integers(1, 80))) if i % 3 else None for i in range(n)]
        for variant in (0, 1, 2):
            for restrict in (1, 0):
                o.restrict = restrict
                paths, st, cnt = e.astar_host(variant, starts, targets, avoid, path_cap=2048, restrict_corner=restrict,
                                              want_counters=True)
                for i in range(n):
                    want, ost = o.astar(int(starts[i]), int(targets[i]), avoid[i], variant)
                    assert st[i] != 3 and np.array_equal(paths[i], want), (g.shape, variant, restrict, i)
                    if len(want) > 1:
                        assert (not SEQ and variant != 1) or (cnt[i, 0] == ost[0] and cnt[i, 1] == ost[1]), (g.shape, variant, restrict, i, cnt[i], ost)
        e.close()


@pytest.mark.parametrize("plateau_kernels", [1, 0])
def test_astar_open_map_plateaus_vs_oracle(plateau_kernels):
    """(Both builds of the pop loop: the separately compiled kernels with the plateau refills that open maps are dispatched
    to, and the plain ones.)  An empty 1024 x 1024 map: thousands of open entries within 1/64 of f (near-ties along straight runs) overflow
    single buckets of the open-list pool, so the spill list and its re-offer at refill time are exercised; nothing
    may overflow and the searches with the largest open lists must match the oracle."""
    from pathfit.engine import Engine
    import pf_oracle as po
    g = np.zeros((1024, 1024), np.uint8)
    e, o = Engine(g), po.Oracle(g)
    e.set_option("plateau_kernels", plateau_kernels)
    rnd = np.random.default_rng(7)
    n = 256
    starts = rnd.integers(0, g.size, n).astype(np.int32); targets = rnd.integers(0, g.size, n).astype(np.int32)
    starts[:4] = [0, 0, g.size - 1, 1023]; targets[:4] = [g.size - 1, 1023 * 1024, 0, 1023 * 1024 + 511]
    for variant in (0, 1):
        paths, st, cnt = e.astar_host(variant, starts, targets, None, path_cap=16 * 2048, want_counters=True)
        assert (st == 0).all()
        if SEQ or variant == 1:
            assert e.counters()["candidates"] > 0      # for A* calls: entries that went through the spill list
        check = set(np.argsort(-cnt[:, 2])[:6].tolist()) | {0, 1, 2, 3}
        for i in sorted(check):
            want, ost = o.astar(int(starts[i]), int(targets[i]), None, variant)
            assert np.array_equal(paths[i], want), (variant, i)
            assert (not SEQ and variant != 1) or cnt[i, 0] == ost[0]
    e.set_option("plateau_kernels", -1)
    e.close()


def test_decode_and_score_golden():
    from pathfit.engine import score_params
    z = gio.load("decode_cases")
    names = [str(s) for s in z["grid_names"]]
    for gid, gname in enumerate(names):
        e, o, s, t, _ = eng(gname)
        for wname, w in (("main", z["main_w"]), ("def", z["def_w"])):
            sp = score_params(0, True, w[0], w[1], w[2], w[3])
            for kind in (0, 1):
                idx = [i for i in range(len(z["kind"])) if z["grid_id"][i] == gid and str(z["weights"][i]) == wname
                       and z["kind"][i] == kind]
                byW = {}
                for i in idx:
                    wp = gio.csr_get(z["wp_off"], z["wp"], i)
                    byW.setdefault(len(wp) if kind == 0 else len(wp) // 2, []).append(i)
                for W, ids in byW.items():
                    if kind == 0:
                        wp = np.array([gio.csr_get(z["wp_off"], z["wp"], i) for i in ids]).astype(np.int32)
                        paths, st, stats = e.decode_host(s, t, wp_cells=wp, sp=sp)
                    else:
                        wp = np.array([gio.csr_get(z["wp_off"], z["wp"], i) for i in ids]).reshape(len(ids), W, 2)
                        paths, st, stats = e.decode_host(s, t, wp_pos=wp, sp=sp)
                    for j, i in enumerate(ids):
                        want = gio.csr_get(z["path_off"], z["path"], i)
                        assert st[j] != 3 and np.array_equal(paths[j], want), (gname, wname, kind, i)
                        assert np.array_equal(stats[j], z["stats"][i]), (gname, i, stats[j], z["stats"][i])
    # stand-alone scoring incl. the hand-built corner-cutting path
    e, o, s, t, _ = eng("fig7")
    i = int(np.flatnonzero(z["kind"] == 2)[0])
    w = z["main_w"]
    got = e.score_host([gio.csr_get(z["path_off"], z["path"], i), np.zeros(0, np.int32)],
                       score_params(0, True, w[0], w[1], w[2], w[3]))
    assert np.array_equal(got[0], z["stats"][i]) and got[0][3] > 0
    assert got[1][0] == math.inf and got[1][4] == math.inf and got[1][1] == 0


def test_decode_random_512_vs_oracle():
    from pathfit.engine import score_params
    e, o, s, t, g = eng("up2:g256")
    rnd = np.random.default_rng(11)
    free = np.flatnonzero(g.reshape(-1) != 1)
    n, W = 24, 5
    wp = rnd.choice(free, (n, W)).astype(np.int32)
    sp = score_params(0, True, 0.3, 0.8, 1.8, 100.0)
    paths, st, stats = e.decode_host(s, t, wp_cells=wp, sp=sp, path_cap=16384)
    feas = 0
    for i in range(n):
        want, _ = o.decode(s, t, wp[i])
        assert st[i] != 3 and np.array_equal(paths[i], want), i
        assert np.array_equal(stats[i], o.score(want, 0, 0.3, 0.8, 1.8, True, 100.0)), i
        feas += len(want) > 0
    assert feas >= 3


def test_pso_update_golden():
    e, _, _, _, _ = eng("fig7")
    z = gio.load("pso_update")
    for i in range(len(z["seed"])):
        w, c1, c2, mv = z["params"][i]
        dp, dv = e.put(z["pos0"][i]), e.put(z["vel0"][i])
        dpb, dgb = e.put(z["pbest"][i]), e.put(z["gbest"][i])
        e.pso_update(1, 5, w, c1, c2, mv, dp, dv, dpb, dgb, int(z["seed"][i]), int(z["it"][i]), 0)
        assert np.array_equal(dp.download(), z["pos1"][i]) and np.array_equal(dv.download(), z["vel1"][i]), i


def test_maaco_golden_walks_and_pheromone():
    from pathfit._lib import MaacoParams
    z = gio.load("maaco_cases")
    bp = z["base_params"]
    for ri, gname in enumerate(z["runs_grid"]):
        beta, n_ants, n_it, K, seed = z["runs_num"][ri]
        n_ants, n_it, K, seed = int(n_ants), int(n_it), int(K), int(seed)
        e, o, s, t, _ = eng(str(gname))
        e.maaco_setup(MaacoParams(bp[0], beta, bp[1], bp[2], bp[3], bp[4], bp[5], bp[6], bp[7], bp[8], K, s, t))
        assert np.array_equal(e.maaco_get_pheromone(), z[f"r{ri}_tau0"])
        cap = 4 * (e.R + e.C) + 64 if e.R > 20 else 400
        dc, dl, dp, dt, ds = e.buf((n_ants, cap), np.int32), e.buf(n_ants, np.int32), e.buf(n_ants, np.float64), \
            e.buf(n_ants, np.int32), e.buf(n_ants, np.int32)
        best, k, dead = math.inf, 0, 0
        for it in range(1, n_it + 1):
            e.maaco_walk(it, seed, 0, n_ants, cap, dc, dl, dp, dt, ds)
            cells, lens, plen, turns, st = dc.download(), dl.download(), dp.download(), dt.download(), ds.download()
            for ant in range(n_ants):
                want = gio.csr_get(z[f"r{ri}_path_off"], z[f"r{ri}_path"], k)
                assert st[ant] != 3 and np.array_equal(cells[ant, :lens[ant]], want), (ri, it, ant)
                assert plen[ant] == z[f"r{ri}_len"][k] and turns[ant] == z[f"r{ri}_turns"][k], (ri, it, ant)
                # MAACO.py:287-288: an ant with no candidate left dies where it stands (88 of the 376 golden ants do,
                # all of them after at least one step: the start always has a free neighbour on these maps)
                assert st[ant] == (1 if len(want) == 0 else 0), (ri, it, ant, st[ant])
                dead += len(want) == 0
                k += 1
            best = min(best, plen.min())
            e.maaco_evaporate(); e.maaco_deposit(n_ants, cap, dc, dl, dp); e.maaco_clip(best)
            if f"r{ri}_tau" in z:
                assert np.array_equal(e.maaco_get_pheromone(), z[f"r{ri}_tau"][it - 1]), (ri, it)
        assert dead > 0, ri
        if f"r{ri}_tau_sum" in z:
            m = e.maaco_get_pheromone()
            assert np.array_equal(m[:4], z[f"r{ri}_tau_last_rows"])
            assert np.array_equal(np.array([m.sum(), m.max(), m.min()]), z[f"r{ri}_tau_sum"])


@pytest.mark.parametrize("pack8_min", [1, 1 << 30])
def test_big_maaco_golden_at_bench_sizes(pack8_min):
    """HIP == the UNMODIFIED reference's MAACO (MAACO.py:278-332) on G512 (8 ants x 2 iterations, beta 7 and beta 2) and G1024
    (4 ants): every ant's walk, length and turns, and the WHOLE pheromone matrix after each update -- with eight ants per
    wavefront (k_maaco_walk8) and one (k_maaco_walk), through the one-pass update (k_tau_update)."""
    from pathfit._lib import MaacoParams
    z = gio.load("big_cases")
    bp = z["maaco_base_params"]
    e0 = eng("fig7")[0]
    e0.set_option("maaco_pack8_min", pack8_min)
    try:
        for ri in range(int(z["maaco_runs"])):
            R, beta, n_ants, n_it, K, seed = z[f"maaco{ri}_cfg"]
            R, n_ants, n_it, K, seed = int(R), int(n_ants), int(n_it), int(K), int(seed)
            e = eng(f"up{R // 256}:g256")[0]
            s, t = 0, R * R - 1
            e.maaco_setup(MaacoParams(bp[0], beta, bp[1], bp[2], bp[3], bp[4], bp[5], bp[6], bp[7], bp[8], K, s, t))
            cap = 6 * (R + R) + 64
            dc, dl, dp, dt, ds = e.buf((n_ants, cap), np.int32), e.buf(n_ants, np.int32), e.buf(n_ants, np.float64), \
                e.buf(n_ants, np.int32), e.buf(n_ants, np.int32)
            best, k = math.inf, 0
            for it in range(1, n_it + 1):
                e.maaco_walk(it, seed, 0, n_ants, cap, dc, dl, dp, dt, ds)
                cells, lens, plen, turns, st = dc.download(), dl.download(), dp.download(), dt.download(), ds.download()
                for ant in range(n_ants):
                    want = gio.csr_get(z[f"maaco{ri}_path_off"], z[f"maaco{ri}_path"], k)
                    assert st[ant] == (1 if len(want) == 0 else 0) and np.array_equal(cells[ant, :lens[ant]], want), (ri, it, ant)
                    assert plen[ant] == z[f"maaco{ri}_len"][k] and turns[ant] == z[f"maaco{ri}_turns"][k], (ri, it, ant)
                    k += 1
                best = min(best, plen.min())
                e.maaco_update(n_ants, cap, dc, dl, dp, best)
                assert np.array_equal(e.maaco_get_pheromone(), z[f"maaco{ri}_tau"][it - 1]), (ri, it)
    finally:
        e0.set_option("maaco_pack8_min", 2048)


def test_mpa_rebuild_golden():
    from pathfit._lib import MpaParams
    from pathfit.engine import score_params
    z = gio.load("mpa_cases")
    names = [str(s) for s in z["grid_names"]]
    seed, it = (int(v) for v in z["seed_it"])
    for gid, gname in enumerate(names):
        e, o, s, t, _ = eng(gname)
        for bi, beta in enumerate((1.5, 2.0)):
            ids = [i for i in range(len(z["idx"])) if z["grid_id"][i] == gid and z["beta"][i] == beta]
            sp = score_params(1, True, 0.1, 0.05, 1.5, 1000.0)
            e.mpa_setup(MpaParams(0.5, beta, float(z["sigma"][bi]), 0.2, len(ids), s, t, 1, 1), sp)
            # group by elite path (the batch call takes one elite)
            groups = {}
            for i in ids:
                groups.setdefault(gio.csr_get(z["el_off"], z["el_path"], i).tobytes(), []).append(i)
            for _, gi in groups.items():
                n = len(gi)
                cap = 4 * (e.R + e.C) + 64 if e.R > 20 else 400
                pop = np.zeros((n, cap), np.int32); plen = np.zeros(n, np.int32)
                for j, i in enumerate(gi):
                    p = gio.csr_get(z["in_off"], z["in_path"], i)
                    pop[j, :len(p)] = p; plen[j] = len(p)
                el = gio.csr_get(z["el_off"], z["el_path"], gi[0])
                pstats = np.array([o.score(pop[j, :plen[j]], 1, 0.1, 0.05, 1.5, True, 1000.0) for j in range(n)])
                dpop, dlen, dstats, del_ = e.put(pop), e.put(plen), e.put(pstats), e.put(el)
                oc, ol, os_, ost = e.buf((n, cap), np.int32), e.buf(n, np.int32), e.buf((n, 5), np.float64), e.buf(n, np.int32)
                # keep the DevBufs alive across the call (a temporary would be freed before the launch)
                d_idx, d_lv = e.put(z["idx"][gi], np.int32), e.put(z["is_levy"][gi], np.int32)
                d_sc, d_ag = e.put(z["scale"][gi], np.float64), e.put(z["agent"][gi], np.int32)
                e._ck(e.L.pf_mpa_rebuild_batch(e.h, it, seed, n, cap, dpop.ptr, dlen.ptr, dstats.ptr, del_.ptr, len(el),
                                               d_idx.ptr, d_lv.ptr, d_sc.ptr, d_ag.ptr, oc.ptr, ol.ptr, os_.ptr, ost.ptr))
                cells, lens, stats, st = oc.download(), ol.download(), os_.download(), ost.download()
                for j, i in enumerate(gi):
                    want = gio.csr_get(z["out_off"], z["out_path"], i)
                    assert st[j] != 3 and np.array_equal(cells[j, :lens[j]], want), (gname, beta, i, st[j])
                    assert np.array_equal(stats[j], z["stats"][i]), (gname, beta, i)


def test_maaco_eight_ants_per_wave_path_matches():
    """k_maaco_walk8 (8 ants per wavefront, in-loop refetch) == goldens: force it on for tiny batches too."""
    e, _, _, _, _ = eng("fig7")
    e.set_option("maaco_pack8_min", 1)
    try:
        test_maaco_golden_walks_and_pheromone()
    finally:
        e.set_option("maaco_pack8_min", 2048)


@pytest.mark.parametrize("pack8", [1, 0])
def test_maaco_tabu_epoch_wrap(pack8):
    """The packed tabu sets carry a 16-bit epoch per word: start every slot just below the wrap so that the ants of one
    batch cross it (slot wipe + epoch restart) and compare with the goldens again."""
    e, _, _, _, _ = eng("fig7")
    e.set_option("maaco_pack8_min", 1 if pack8 else 1 << 30)
    try:
        for ep in (0xFFF0 - 1, 0xFFF0 - 2):     # (descending: a slot never revisits an epoch between two wipes)
            e.set_option("maaco_tabu_epoch", ep)
            test_maaco_golden_walks_and_pheromone()
    finally:
        e.set_option("maaco_pack8_min", 2048)
        e.set_option("maaco_tabu_epoch", -1)


def test_maaco_packed_and_single_ant_kernels_agree_1024():
    """Long walks on G1024 (~1700 steps, ants that wander back into cells they left long ago): the 8-ants-per-wave kernel
    and the one-ant-per-wave kernel must emit identical walks, also across an epoch wrap in the middle of the batch."""
    import pathfit
    from pathfit import env
    g = env.bench_grid(1024)
    m = pathfit.MAACO(g, 4096, 100, 1.0, 7.0, 0.1, 2.5, 1.0, 0.9, 0.2, 0.9, 0.5, 0.1, seed=11)
    res = []
    for pack8_min, ep in ((1, -1), (1 << 30, -1), (1, 0xFFF0 - 1), (1, 0xFFF0 - 2)):
        m.engine.set_option("maaco_pack8_min", pack8_min)
        m.engine.set_option("maaco_tabu_epoch", ep)
        m.walk_iteration_dev(3)
        dc, dl, dp, dt, ds = m.walk_bufs()
        res.append((dc.download(), dl.download(), dp.download(), dt.download(), ds.download()))
    m.engine.set_option("maaco_pack8_min", 2048)
    m.engine.set_option("maaco_tabu_epoch", -1)
    a = res[0]
    for b in res[1:]:
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
        for i in range(4096):
            assert np.array_equal(a[0][i, :a[1][i]], b[0][i, :b[1][i]]), i
    assert (a[1] > 0).mean() > 0.3


def _levy_sigma(beta):
    num = math.gamma(1 + beta) * math.sin(math.pi * beta / 2)
    den = math.gamma((1 + beta) / 2) * beta * (2 ** ((beta - 1) / 2))
    return (num / den) ** (1 / beta)


def _targets(e, seed, is_levy, beta, scale, cur, elite):
    import ctypes as C
    d_cur, d_el, d_out = e.put(cur, np.int32), e.put(elite, np.int32), e.buf(cur.size, np.int32)
    nd = C.c_int64(0)
    e._ck(e.L.pf_selftest_mpa_targets(e.h, seed, cur.size, int(is_levy), beta, _levy_sigma(beta), scale, d_cur.ptr, d_el.ptr,
                                      d_out.ptr, C.byref(nd)))
    return d_out.download(), int(nd.value)


def test_mpa_target_cells_two_million_draws_vs_oracle():
    """MPA.py:250-282 on the device (ocml log / pow / sin / cos, decisions near a boundary handed to the host's glibc)
    == the oracle's libm arithmetic on 2.5 M keyed draws: Levy with beta 1.5 and scales that make the step actually
    move (the main.py beta = 2 setting almost never does), Brownian with both branches."""
    import pf_oracle as po
    e, o, s, t, g = eng("up2:g256")
    rnd = np.random.default_rng(11)
    n = 500_000
    cur = rnd.integers(0, g.size, n).astype(np.int32)
    elite = rnd.integers(0, g.size, n).astype(np.int32)
    elite[::7] = -1                                       # elite_node_on_path is None
    elite[1::11] = cur[1::11]                             # dist <= 1e-6 -> return the elite node
    doubts = 0
    for is_levy, beta, scale in ((1, 1.5, 40.0), (1, 1.5, 800.0), (1, 2.0, 300.0), (0, 1.5, 0.5), (0, 1.5, 6.0)):
        seed = 1000 + int(scale)
        got, nd = _targets(e, seed, is_levy, beta, scale, cur, elite)
        want = po.mpa_targets_batch(seed, is_levy, e.R, e.C, cur, elite, scale, beta, _levy_sigma(beta))
        assert np.array_equal(got, want), (is_levy, beta, scale, int((got != want).sum()))
        if is_levy and beta == 1.5:
            assert (got != cur).mean() > 0.5              # the Levy branch really moves at these scales (with beta = 2 the
                                                          # Mantegna sigma is sin(pi) ~ 1e-16: the step is always 0, MPA.py:251)
        doubts += nd
    assert doubts < 100                                   # the host route is the exception (expected ~0)


def test_mpa_doubtful_proposals_take_the_host_route():
    """Widen the margins so that EVERY proposal counts as doubtful: all of them are then recomputed by the host's libm
    (mpa_resolve_doubts) and the golden rebuilds / the target sweep must still come out bit for bit."""
    import pf_oracle as po
    e, o, s, t, g = eng("fig7")
    before = e.L.pf_mpa_doubts_resolved(e.h)
    for ee in {id(v[0]): v[0] for v in _eng.values()}.values():
        ee.set_option("mpa_doubt_round_e15", 600_000_000_000_000)      # 0.6 > any |frac - 0.5|
        ee.set_option("mpa_doubt_log_e15", 10 ** 18)
    try:
        test_mpa_rebuild_golden()
        rnd = np.random.default_rng(3)
        cur = rnd.integers(0, g.size, 2000).astype(np.int32); elite = rnd.integers(0, g.size, 2000).astype(np.int32)
        for is_levy, scale in ((1, 30.0), (0, 2.0)):
            got, nd = _targets(e, 5, is_levy, 1.5, scale, cur, elite)
            assert nd >= 1900                          # (a Brownian proposal that returns the elite node draws no normal deviate)
            assert np.array_equal(got, po.mpa_targets_batch(5, is_levy, e.R, e.C, cur, elite, scale, 1.5, _levy_sigma(1.5)))
    finally:
        for ee in {id(v[0]): v[0] for v in _eng.values()}.values():
            ee.set_option("mpa_doubt_round_e15", -1)
            ee.set_option("mpa_doubt_log_e15", -1)
    assert e.L.pf_mpa_doubts_resolved(e.h) > before or any(v[0].L.pf_mpa_doubts_resolved(v[0].h) > 0 for v in _eng.values())


def test_step_cap_path_vs_oracle():
    """astar.py:58 / MPA.py:118: `while open_set and steps < max_steps`.  The reference's caps (3RC / 2RC loop
    iterations) cannot be reached on a grid (every cell is popped once, re-pops are a few per cent), so
    the cap path is exercised with the cap lowered on both sides by the same test hook: same status, same pop count,
    no path -- including caps that fall in the middle of a seven-head trip."""
    import pf_oracle as po
    e, o, s, t, g = eng("up2:g256")
    rnd = np.random.default_rng(9)
    free = np.flatnonzero(g.reshape(-1) != 1)
    n = 24
    starts = rnd.choice(free, n); targets = rnd.choice(free, n)
    starts[0], targets[0] = s, t
    try:
        for cap in (1, 2, 5, 7, 8, 13, 1000, 1003, 20011):
            e.set_option("astar_step_cap", cap); po.set_step_cap(cap)
            for variant in (0, 1, 2):
                paths, st, cnt = e.astar_host(variant, starts, targets, None, path_cap=8192, want_counters=True)
                for i in range(n):
                    want, ost = o.astar(int(starts[i]), int(targets[i]), None, variant)
                    assert np.array_equal(paths[i], want), (cap, variant, i)
                    if st[i] == 1 and ost[5] == 2:
                        # the engine proves "no path" without searching (component / pocket proofs): the uncapped
                        # reference must agree that there is none
                        po.set_step_cap(0)
                        assert len(o.astar(int(starts[i]), int(targets[i]), None, variant)[0]) == 0
                        po.set_step_cap(cap)
                        continue
                    assert st[i] == ost[5], (cap, variant, i, st[i], ost[5])
                    if ost[5] == 2:
                        assert cnt[i, 0] == cap and len(want) == 0
    finally:
        e.set_option("astar_step_cap", 0); po.set_step_cap(0)
