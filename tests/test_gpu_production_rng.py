"""The device-side generator that production runs use in place of the reference's three random sources
(numpy.random.randint for transform ids, random.gammavariate for the root noise, random.choices for the move;
search/mcts.py:4-6, math_helper.py:13-23).  Replay mode is held to the reference draw for draw elsewhere; here the
generator's distributions are checked statistically, through the C ABI, at 4.5 sigma of the sampling error."""
import numpy as np
import pytest
from scipy import stats

pytestmark = pytest.mark.gpu


def _noise_draws(alpha, G=48, seed=1):
    """Gamma draws as the engine applies them: P <- (1 - eps) * P + eps * eta on the children of a fresh root
    (mcts.py:285-291).  float64 search, known uniform priors, eps = 0.5 -> eta recovered to ~1e-16."""
    from alphazero_amd.engine import Engine
    H = 19
    eng = Engine("go", H, H, n_slots=G, max_playout=0, transform_types=0, d_alpha=alpha, d_epsilon=0.5, seed=seed, search_f64=True)
    O = eng.O
    eng.search_begin(np.arange(G), np.zeros(G, np.int32), np.ones(G, np.int32))
    assert eng.step_select() == G
    eng.step_finish(np.full((G, O), 1.0 / O), np.zeros(G))
    etas = []
    for s in range(G):
        a, n, w, p, rn = eng.root_stats(s, 0)
        assert len(a) == O
        etas.append((p - 0.5 / O) / 0.5)
    eng.close()
    return np.concatenate(etas)


@pytest.mark.parametrize("alpha", [0.03, 0.5, 1.0, 2.5])
def test_device_gamma_has_the_distribution_of_gammavariate(alpha):
    x = _noise_draws(alpha)
    n = x.size
    assert n == 48 * 362 and np.all(x >= -1e-15)
    dist = stats.gamma(alpha)
    # distribution function at points that matter for a heavy-headed gamma: binomial error bound per point
    for q in (1e-9, 1e-4, 1e-2, 0.1, 0.5, 1.0, 2.0, 4.0):
        p = dist.cdf(q)
        if p * n < 50 or (1 - p) * n < 50:
            continue
        got = np.mean(x <= q)
        assert abs(got - p) <= 4.5 * np.sqrt(p * (1 - p) / n), (q, got, p)
    # Kolmogorov-Smirnov distance over the part float64 resolves (eta below 1e-13 is lost in P + eta)
    big = x[x > 1e-12]
    frac = dist.sf(1e-12)
    grid = np.sort(big)
    emp = (np.arange(1, grid.size + 1) / grid.size)
    cond = (dist.cdf(grid) - dist.cdf(1e-12)) / frac
    assert np.abs(emp - cond).max() <= 1.95 / np.sqrt(grid.size)          # p ~ 1e-3 for a true sample
    # first two moments: mean alpha, variance alpha (standard errors from the 2nd / 4th central moments)
    m2, m4 = dist.var(), dist.moment(4) - 4 * dist.moment(3) * alpha + 6 * dist.moment(2) * alpha ** 2 - 3 * alpha ** 4
    assert abs(x.mean() - alpha) <= 4.5 * np.sqrt(m2 / n)
    assert abs(x.var() - m2) <= 4.5 * np.sqrt((m4 - m2 ** 2) / n) + 1e-12
    # independent draws: neighbouring children of one root are uncorrelated
    r = np.corrcoef(np.minimum(x[:-1], 1.0), np.minimum(x[1:], 1.0))[0, 1]
    assert abs(r) <= 4.5 / np.sqrt(n)


def test_engine_rejects_bad_noise_parameters():
    from alphazero_amd.engine import Engine, EngineError
    with pytest.raises(EngineError):
        Engine("go", 9, 9, d_alpha=0.0)
    with pytest.raises(EngineError):
        Engine("go", 9, 9, d_epsilon=1.5)
    Engine("go", 9, 9, d_alpha=0.0, rng_replay=True).close()    # replayed draws: alpha is not used


def test_device_transform_ids_are_uniform():
    """One draw per playout (mcts.py:208): the leaf planes of an asymmetric position tell which of the 8 ids was drawn."""
    from alphazero_amd.engine import Engine
    H, G = 9, 4096
    eng = Engine("go", H, H, n_slots=G, max_playout=4, transform_types=8, seed=7)
    for mv in (2, 12, 40, 31):                               # same asymmetric position in every slot
        eng.do_move(np.arange(G), np.full(G, mv))
    forms = eng.features([0] * 8, list(range(8)))            # the position under each id
    assert len({f.tobytes() for f in forms}) == 8
    eng.search_begin(np.arange(G), np.zeros(G, np.int32), np.zeros(G, np.int32))
    counts = np.zeros(8)
    rounds = 0
    while rounds < 3 and eng.step_select() > 0:              # root expansion + two playouts: 3 draws per slot
        planes, valid = eng.leaf_batch()
        if rounds == 0:
            key = {f.tobytes(): i for i, f in enumerate(forms)}
            tid = np.array([key[p.tobytes()] for p in planes])
            counts += np.bincount(tid, minlength=8)
            first = tid
        eng.step_finish(np.full((G, eng.O), 1.0 / eng.O, np.float32), np.zeros(G, np.float32))
        rounds += 1
    chi2 = ((counts - G / 8) ** 2 / (G / 8)).sum()
    assert chi2 <= stats.chi2(7).isf(1e-5), (counts, chi2)
    # neighbouring slots draw independently
    both = np.zeros((8, 8)); np.add.at(both, (first[:-1], first[1:]), 1)
    assert ((both - (G - 1) / 64) ** 2 / ((G - 1) / 64)).sum() <= stats.chi2(63).isf(1e-5)
    eng.close()


def test_device_move_choice_follows_visit_counts():
    """Game.start picks the move with random.choices over pi = N / sum N (mcts.py:331-334, math_helper.py:20-23): with
    identical trees in 4096 slots the chosen first moves are a sample of pi."""
    from alphazero_amd.engine import Engine
    G, P = 4096, 60
    eng = Engine("mnk", 3, 3, 3, n_slots=G, max_playout=P, transform_types=0, dirichlet_before=-1, seed=11, record_trajectories=True)
    pri = np.tile(np.array([[.30, .05, .10, .05, .20, .05, .10, .05, .10]], np.float32), (G, 1))
    eng.selfplay_start()
    for _ in range(P + 1):
        assert eng.selfplay_select() == G
        eng.selfplay_finish(pri, np.zeros(G, np.float32))
    plies, done = eng.selfplay_status()
    assert np.all(plies == 1)
    # every slot has searched the same tree; its root counts were logged with the move
    eng2 = Engine("mnk", 3, 3, 3, n_slots=1, max_playout=P, transform_types=0, dirichlet_before=-1, seed=1)
    eng2.search_begin([0], [0], [0])
    while eng2.step_select() > 0:
        eng2.step_finish(pri[:1], np.zeros(1, np.float32))
    a, n, w, p, rn = eng2.root_stats(0, 0)
    pi = np.zeros(9); pi[a] = n / n.sum()
    st = [eng.state_get(s) for s in range(0, G, 1)]
    moved = np.array([int(np.flatnonzero(np.frombuffer(b.board, np.int8, 9))[0]) for b in st])
    counts = np.bincount(moved, minlength=9)
    live = pi > 0
    chi2 = ((counts[live] - G * pi[live]) ** 2 / (G * pi[live])).sum()
    assert counts[~live].sum() == 0 and chi2 <= stats.chi2(live.sum() - 1).isf(1e-5), (counts, G * pi)
    eng.close(); eng2.close()


def test_new_weights_are_used_from_the_next_round_on():
    """The model update of train/parallel/selfplay.py:110-122 on the device: weights loaded between two rounds of
    running games serve every leaf of the next round (and none of the previous one)."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.network.model import fold_to_blob, random_net
    H, G, NB, C = 9, 32, 1, 128
    blob_a = fold_to_blob(random_net(H, H, 82, NB, C, seed=1, perturb_bn=True))
    blob_b = fold_to_blob(random_net(H, H, 82, NB, C, seed=2, perturb_bn=True))
    eng = Engine("go", H, H, n_slots=G, max_playout=20, transform_types=8, seed=3, num_blocks=NB, channels=C, tower_dtype="fp32",
                 record_trajectories=True)
    probe = {}
    for name, blob in (("a", blob_a), ("b", blob_b)):
        probe[name] = Engine("go", H, H, n_slots=G, num_blocks=NB, channels=C, tower_dtype="fp32")
        probe[name].tower_load(blob)
    eng.tower_load(blob_a)
    eng.selfplay_start()
    eng.selfplay_run(30)                                       # games are under way (ply 1, mid-search)
    for rnd, name in ((0, "a"), (1, "b"), (2, "b")):
        if rnd == 1:
            eng.tower_load(blob_b)                             # the update arrives while the slots are mid-game
        eng.selfplay_run(1)
        planes, valid = eng.leaf_batch()
        p, v = eng.leaf_outputs()
        for who in ("a", "b"):
            rp, rv = probe[who].tower_forward(planes)
            same = np.array_equal(p, rp) and np.array_equal(v, rv)
            assert same == (who == name), (rnd, who)
    plies, _ = eng.selfplay_status()
    assert np.all(plies >= 1)
    for e in (eng, probe["a"], probe["b"]):
        e.close()
