"""GPU parity and bookkeeping of the shared pool of tree edges: the trees of an engine take their edges in pages from
one pool (the reference's tree is a heap of objects without a limit, search/mcts.py:30-52).  Whole games must still
equal the C oracle's visit for visit whatever the page size and however scarce the pool, and every page must be
accounted for."""
import numpy as np
import pytest

from oracle import oracle as oz
from oracle.fake_eval import fake_eval_planes

pytestmark = pytest.mark.gpu


def _streams(rng, G, mt, P, O):
    out = []
    for _ in range(G):
        out.append((rng.randint(0, 8, size=(mt + 2) * (P + 1)).astype(np.int32), rng.gamma(0.03, 1.0, size=(mt + 2) * O), rng.rand(mt + 2)))
    return out


def _pages_held(eng):
    """Pages the trees need for the edges they hold, from the per-tree sizes."""
    page = eng.counters()["tree_page_edges"]
    return sum(max(1, -(-eng.tree_size(s, side)[0] // page)) for s in range(eng.G) for side in (0, 1))   # a tree's first page is its own for good


@pytest.mark.parametrize("game,H,P,mt,G,page,pool_frac", [
    ("go", 9, 60, 400, 6, 128, 0.30),      # smallest page a 9x9 board allows: most nodes' children lie across two pages
    ("go", 9, 60, 40, 6, 0, 0.0),          # automatic page and pool
    ("reversi", 8, 80, 400, 5, 128, 0.35),
    ("go", 19, 200, 6, 4, 512, 0.25),      # 362 children in 512-edge pages
])
def test_whole_games_in_a_scarce_pool_equal_the_oracle(game, H, P, mt, G, page, pool_frac):
    from alphazero_amd.engine import Engine
    rng = np.random.RandomState(77 + H + P)
    O = H * H + 1
    pool = int(pool_frac * 2 * G * 3 * (P + 1) * O)
    eng = Engine(game, H, H, n_slots=G, max_playout=P, transform_types=8, dirichlet_before=30, max_turn=mt, rng_replay=True,
                 record_trajectories=True, tree_page_edges=page, tree_pool_edges=pool)
    c0 = eng.counters()
    assert c0["tree_pool_pages_in_use"] == 2 * G and c0["tree_pool_pages"] > 2 * G
    if page:
        assert c0["tree_page_edges"] == page and c0["tree_pool_pages"] == -(-pool // page)
    streams = _streams(rng, G, mt, P, O)
    for s in range(G):
        eng.rng_inject(s, *streams[s])
    eng.selfplay_start()
    rounds = 0
    while eng.selfplay_select() > 0:
        planes, _ = eng.leaf_batch()
        eng.selfplay_finish(*fake_eval_planes(planes, O))
        rounds += 1
        if rounds % 97 == 0:   # every page in use belongs to exactly one tree, at any moment
            assert eng.counters()["tree_pool_pages_in_use"] == _pages_held(eng)
    c = eng.counters()
    assert c["expansions_skipped"] == 0
    assert c["tree_pool_pages_in_use"] == _pages_held(eng)
    assert c["tree_pool_pages_in_use"] <= c["tree_pool_pages_peak"] <= c["tree_pool_pages"]
    assert c["tree_pool_pages_peak"] * c["tree_page_edges"] >= c["peak_tree_edges"]
    for s in range(G):
        ref = oz.play_game(game, H, H, 0, 8, P, 30, mt, oz.fn_ptr("oz_fake_eval"), oz.Rng(*streams[s]))
        tr = eng.trajectory_pop(s)
        assert np.array_equal(tr["moves"], ref["moves"]) and np.array_equal(tr["visits"], ref["visits"]) and tr["winner"] == ref["winner"], s
    eng.reset(list(range(G)))
    assert eng.counters()["tree_pool_pages_in_use"] == 2 * G   # a new game hands every page but the trees' first ones back
    eng.close()


def _run_until_done(eng, evaluate):
    while eng.selfplay_select() > 0:
        planes, _ = eng.leaf_batch()
        eng.selfplay_finish(*evaluate(planes, eng.O))


def test_an_empty_pool_is_reported_like_a_full_tree():
    from alphazero_amd.engine import Engine, EngineError
    # 2 slots = 4 trees, 4 pages of 64 edges: a root's 50 children take a tree's only page, the first expansion below it finds none
    eng = Engine("go", 7, 7, n_slots=2, max_playout=30, transform_types=8, tree_page_edges=64, tree_pool_edges=4 * 64)
    assert eng.counters()["tree_pool_pages"] == 4
    eng.selfplay_start()
    with pytest.raises(EngineError) as ei:
        _run_until_done(eng, fake_eval_planes)
    assert ei.value.code == -4
    eng.close()


def test_an_empty_pool_with_the_skip_policy_keeps_playing():
    from alphazero_amd.engine import Engine
    eng = Engine("go", 7, 7, n_slots=2, max_playout=30, transform_types=8, max_turn=6, tree_page_edges=64, tree_pool_edges=4 * 64,
                 tree_full_policy=1, record_trajectories=True)
    eng.selfplay_start()
    _run_until_done(eng, fake_eval_planes)
    c = eng.counters()
    assert c["expansions_skipped"] > 0 and c["games_finished"] == 2 and c["tree_pool_pages_peak"] <= 4
    for s in range(2):
        tr = eng.trajectory_pop(s)
        assert np.all(tr["visits"].sum(axis=1) >= 30)   # every playout still counted once at the root (plus what a kept subtree brought)
    eng.close()


def test_an_empty_pool_with_the_grow_policy_stays_exact():
    """tree_full_policy 2 with a pool of one page per tree: pool and limits double as often as needed, and the games
    equal the oracle's (unlimited trees) visit for visit."""
    from alphazero_amd.engine import Engine
    game, H, P, mt, G = "go", 7, 40, 30, 3
    rng = np.random.RandomState(5)
    O = H * H + 1
    eng = Engine(game, H, H, n_slots=G, max_playout=P, transform_types=8, dirichlet_before=6, max_turn=mt, rng_replay=True,
                 record_trajectories=True, tree_page_edges=64, tree_pool_edges=2 * G * 64, tree_full_policy=2)
    pages0 = eng.counters()["tree_pool_pages"]
    streams = _streams(rng, G, mt, P, O)
    for s in range(G):
        eng.rng_inject(s, *streams[s])
    eng.selfplay_start()
    _run_until_done(eng, fake_eval_planes)
    c = eng.counters()
    assert c["tree_pool_pages"] > pages0 and c["expansions_skipped"] == 0
    assert c["tree_pool_pages_in_use"] == _pages_held(eng)
    for s in range(G):
        ref = oz.play_game(game, H, H, 0, 8, P, 6, mt, oz.fn_ptr("oz_fake_eval"), oz.Rng(*streams[s]))
        tr = eng.trajectory_pop(s)
        assert np.array_equal(tr["moves"], ref["moves"]) and np.array_equal(tr["visits"], ref["visits"]), s
    eng.close()


def test_explicit_growth_keeps_trees_and_accounts_for_the_new_pages():
    from alphazero_amd.engine import Engine
    eng = Engine("go", 9, 9, n_slots=2, max_playout=120, transform_types=8, tree_page_edges=128, seed=3)
    eng.search_begin([0, 1], [0, 1], [1, 0])
    for _ in range(70):
        assert eng.step_select() > 0
        planes, _ = eng.leaf_batch()
        eng.step_finish(*fake_eval_planes(planes, eng.O))
    before = [eng.root_stats(s, s) for s in (0, 1)]
    c0 = eng.counters()
    eng.grow_trees()
    c1 = eng.counters()
    assert c1["tree_edge_cap"] == 2 * c0["tree_edge_cap"] and c1["tree_pool_pages"] == 2 * c0["tree_pool_pages"]
    assert c1["tree_pool_pages_in_use"] == c0["tree_pool_pages_in_use"] == _pages_held(eng)
    assert c1["tree_pool_pages_peak"] == c0["tree_pool_pages_peak"]
    for s in (0, 1):
        for x, y in zip(before[s], eng.root_stats(s, s)):
            assert np.array_equal(x, y)
    while eng.step_select() > 0:   # and the searches go on in the enlarged pool
        planes, _ = eng.leaf_batch()
        eng.step_finish(*fake_eval_planes(planes, eng.O))
    for s in (0, 1):
        a, n, w, p, rootN = eng.root_stats(s, s)
        assert rootN == 121 and n.sum() == 120
    assert eng.counters()["tree_pool_pages_in_use"] == _pages_held(eng)
    eng.close()


def test_a_slot_listed_twice_in_a_reset_is_refused():
    from alphazero_amd.engine import Engine, EngineError
    eng = Engine("go", 5, 5, n_slots=2, max_playout=8)
    with pytest.raises(EngineError):
        eng.reset([1, 1])
    eng.close()


def test_bad_page_size_is_refused():
    from alphazero_amd.engine import Engine, EngineError
    for page in (100, 64):   # not a power of two; smaller than the 82 actions of a 9x9 board
        with pytest.raises(EngineError):
            Engine("go", 9, 9, n_slots=1, max_playout=8, tree_page_edges=page)


def test_growing_only_the_pool_keeps_limits_and_trees():
    from alphazero_amd.engine import Engine
    eng = Engine("go", 9, 9, n_slots=2, max_playout=100, transform_types=8, tree_page_edges=128, tree_pool_edges=6 * 128, seed=4)
    eng.search_begin([0, 1], [0, 0], [0, 1])
    for _ in range(3):
        assert eng.step_select() > 0
        planes, _ = eng.leaf_batch()
        eng.step_finish(*fake_eval_planes(planes, eng.O))
    before = [eng.root_stats(s, 0) for s in (0, 1)]
    c0 = eng.counters()
    assert c0["tree_pool_pages"] == 6
    eng.grow_pool()
    c1 = eng.counters()
    assert c1["tree_pool_pages"] == 12 and c1["tree_edge_cap"] == c0["tree_edge_cap"] and c1["tree_block_cap"] == c0["tree_block_cap"]
    assert c1["tree_pool_pages_in_use"] == c0["tree_pool_pages_in_use"] == _pages_held(eng)
    eng.grow_pool(10 ** 12)   # never beyond every tree at its limit
    most = 4 * -(-c0["tree_edge_cap"] // 128)
    assert eng.counters()["tree_pool_pages"] == most
    eng.grow_pool()           # already there: nothing to do
    assert eng.counters()["tree_pool_pages"] == most
    for s in (0, 1):
        for x, y in zip(before[s], eng.root_stats(s, 0)):
            assert np.array_equal(x, y)
    while eng.step_select() > 0:
        planes, _ = eng.leaf_batch()
        eng.step_finish(*fake_eval_planes(planes, eng.O))
    for s in (0, 1):
        a, n, w, p, rootN = eng.root_stats(s, 0)
        assert rootN == 101 and n.sum() == 100
    eng.close()


def test_batched_selfplay_grows_its_pool_before_a_page_is_missed():
    """BatchedSelfplay starts with whatever pool it is given and doubles it between polls once a set share of it (80 %; 40 %
    here) has been in use: with a pool of one page per tree and a poll after every round, whole games come out the same as with an ample pool and no
    expansion is skipped."""
    import yaml
    import os
    from alphazero_amd.evaluator.nn_eval import NNEvaluator
    from alphazero_amd.game.gameplay import BatchedSelfplay
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "alphazero_amd", "config", "go.yaml")) as f:
        cfg = yaml.safe_load(f)
    H, P, G, mt = 7, 40, 4, 30
    O = H * H + 1
    cfg.update(board_height=H, board_width=H, flat_move_output=O, transform_types=8)
    ev = NNEvaluator(cfg, num_blocks=1, channels=128, dtype="fp32", max_batch_size=1)
    ext = {"player": {"max_playout": P}, "dirichlet_before": 6, "max_turn": mt}
    streams = _streams(np.random.RandomState(9), G, mt, P, O)
    out = []
    for pool in (2 * G * 64, 0):
        sp = BatchedSelfplay(ev, cfg, ext, n_games=G, rng_streams=streams, tree_pool_edges=pool)
        pages0 = sp.eng.counters()["tree_pool_pages"]
        sp.POOL_GROW_AT = 0.4    # (four trees search in step and may each want a page in one round: keep more than that free)
        hist = sp.run(G, rounds_per_poll=1)
        c = sp.eng.counters()
        assert c["expansions_skipped"] == 0
        if pool:
            assert pages0 == 2 * G and c["tree_pool_pages"] > pages0
        out.append([hist[sp.last_slots.index(s)] for s in range(G)])
        sp.eng.close()
    for a, b in zip(*out):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
