"""GPU parity: PUCT select / expand / backup / re-root / move choice of the HIP engine against
fixtures recorded from the reference's MCTSearch with injected float32 priors and values and
replayed RNG draws.  Visit counts, W and P are compared bit for bit, through the C ABI."""
import numpy as np
import pytest

from oracle import oracle as oz
from oracle.fake_eval import fake_eval_planes
from tests.helpers import golden_names, load_golden

pytestmark = pytest.mark.gpu


def _engine_for(g, n_slots=1, record=False, **kw):
    from alphazero_amd.engine import Engine
    return Engine(str(g["game"]), int(g["H"]), int(g["W"]), int(g["k"]), n_slots=n_slots,
                  max_playout=int(g["max_playout"]), transform_types=int(g["transform_types"]),
                  dirichlet_before=int(g["dirichlet_before"]), max_turn=int(g["max_turn"]),
                  d_alpha=float(g["d_alpha"]), d_epsilon=float(g["d_epsilon"]), rng_replay=True,
                  record_trajectories=record, c_punt=float(g["c_punt"]) if "c_punt" in g else 5.0, **kw)


def _evaluate(planes, O, evaluator):
    """The fixture's evaluator: the float32 hash, the same hash widened to float64, or DummyEvaluator's
    float64 uniform priors with value 0 (evaluator/dummy_eval.py:26-28)."""
    if evaluator == "uniform":
        return np.full((planes.shape[0], O), 1 / O), np.zeros(planes.shape[0])
    pri, v = fake_eval_planes(planes, O)
    if evaluator == "fake":
        return pri.astype(np.float64), v.astype(np.float64)
    return pri, v


def _run_search(eng, evaluator="fake32"):
    O = eng.O
    rounds = 0
    while True:
        n = eng.step_select()
        if n == 0:
            break
        planes, valid = eng.leaf_batch()
        pri, v = _evaluate(planes, O, evaluator)
        eng.step_finish(pri, v)
        rounds += 1
    return rounds


@pytest.mark.parametrize("name", golden_names("search_") + golden_names("search64_"))
def test_search_trace_manual_api(name):
    g = load_golden(name)
    evaluator = str(g["evaluator"]) if "evaluator" in g else "fake32"
    f64 = evaluator != "fake32"   # float64 evaluator: the whole search runs in float64, W and P compared as 64-bit patterns
    bits = np.uint64 if f64 else np.uint32
    eng = _engine_for(g, search_f64=f64)
    assert eng.real is (np.float64 if f64 else np.float32)
    eng.rng_inject(0, g["tids"], g["etas"], g["unis"])
    side, evals, ui = 0, 0, 0
    for ply in range(len(g["moves"])):
        assert np.array_equal(eng.board(0), g["boards"][ply]), ply
        st = eng.state_get(0)
        dirichlet = st.turns <= int(g["dirichlet_before"])
        eng.search_begin([0], [side], [int(dirichlet)])
        evals += _run_search(eng, evaluator)
        a, n, w, p, rootN = eng.root_stats(0, side)
        nc = int(g["n_children"][ply])
        assert len(a) == nc, (ply, len(a), nc)
        assert np.array_equal(a, g["acts"][ply][:nc]), ply
        assert np.array_equal(n, g["N"][ply][:nc]), ply
        assert np.array_equal(w.view(bits), g["Wv"][ply][:nc].view(bits)), ply
        assert np.array_equal(p.view(bits), g["P"][ply][:nc].view(bits)), ply
        assert rootN == g["rootN"][ply]
        assert evals == g["cum_evals"][ply]
        it, ie, iu = eng.rng_cursor(0)
        assert it == g["cum_tids"][ply] and ie == g["cum_etas"][ply]
        mv = eng.choose(0, side, float(g["unis"][ui])); ui += 1
        assert mv == g["moves"][ply], ply
        status, _ = eng.do_move([0], [mv])
        assert status[0] == 0
        eng.advance([0], [mv])
        side ^= 1
    assert np.array_equal(eng.board(0), g["final_board"])
    eng.close()


def _selfplay_injected(eng):
    while True:
        n = eng.selfplay_select()
        if n == 0:
            break
        planes, _ = eng.leaf_batch()
        pri, v = fake_eval_planes(planes, eng.O)
        eng.selfplay_finish(pri, v)


@pytest.mark.parametrize("name", ["search_ttt_game", "search_reversi_game", "search_go5_game", "search_ttt_draw", "search_reversi6_pass", "search_go5_maxturn"])
def test_selfplay_machine_full_game(name):
    """Game.start + get_history raw material from the device-side self-play machine, 3 slots
    fed the same draws must all reproduce the reference's game."""
    g = load_golden(name)
    if int(g["ply_limit"]) <= len(g["moves"]):
        pytest.skip("fixture is not a complete game")
    G = 3
    eng = _engine_for(g, n_slots=G, record=True)
    for s in range(G):
        eng.rng_inject(s, g["tids"], g["etas"], g["unis"])
    eng.selfplay_start()
    _selfplay_injected(eng)
    plies, done = eng.selfplay_status()
    assert np.all(done == 1) and np.all(plies == len(g["moves"]))
    O = eng.O
    for s in range(G):
        tr = eng.trajectory_pop(s)
        assert np.array_equal(tr["moves"], g["moves"])
        assert tr["winner"] == g["winner"]
        for ply in range(len(g["moves"])):
            nc = int(g["n_children"][ply])
            exp = np.zeros(O, np.int32)
            exp[g["acts"][ply][:nc]] = g["N"][ply][:nc]
            assert np.array_equal(tr["visits"][ply], exp), ply
        if "hist_probs" in g:
            probs = tr["visits"].astype(np.float64) / tr["visits"].sum(axis=1, keepdims=True)
            assert np.array_equal(probs, g["hist_probs"])
    assert np.array_equal(eng.board(0), g["final_board"])
    c = eng.counters()
    assert c["evals"] == G * int(g["total_evals"]) and c["moves"] == G * len(g["moves"])
    eng.close()


@pytest.mark.parametrize("game,H,W,k,tt,P,db,mt,G", [
    ("mnk", 3, 3, 3, 0, 100, 30, 400, 16),
    ("go", 5, 5, 0, 8, 40, 6, 30, 16),
    ("reversi", 6, 6, 0, 8, 24, 30, 400, 8),
    ("go", 9, 9, 0, 8, 48, 30, 12, 8),
    ("mnk", 7, 7, 4, 8, 30, 30, 400, 8),
    # the named configurations at their full simulation counts, a few plies of a few games each
    ("reversi", 8, 8, 0, 8, 400, 30, 9, 4),
    ("go", 9, 9, 0, 8, 400, 30, 7, 4),
    ("mnk", 15, 15, 5, 8, 800, 30, 3, 2),
    ("go", 19, 19, 0, 8, 800, 30, 5, 4),
    # whole games to their natural end (two passes / full board), noise running out after ply 30
    ("go", 9, 9, 0, 8, 60, 30, 400, 3),
    ("reversi", 8, 8, 0, 8, 60, 30, 400, 3),
])
def test_selfplay_machine_vs_oracle(game, H, W, k, tt, P, db, mt, G):
    """G different seeded games in one engine vs the C oracle's oz_play_game, full trajectories."""
    from alphazero_amd.engine import Engine
    rng = np.random.RandomState(1234 + H * 7 + P)
    eng = Engine(game, H, W, k, n_slots=G, max_playout=P, transform_types=tt, dirichlet_before=db, max_turn=mt,
                 rng_replay=True, record_trajectories=True)
    O = eng.O
    streams = []
    nmax = (mt + 2) * (P + 1)
    for s in range(G):
        tids = rng.randint(0, 8, size=nmax).astype(np.int32)
        etas = rng.gamma(0.03, 1.0, size=(mt + 2) * O)
        unis = rng.rand(mt + 2)
        streams.append((tids, etas, unis))
        eng.rng_inject(s, tids, etas, unis)
    eng.selfplay_start()
    _selfplay_injected(eng)
    plies, done = eng.selfplay_status()
    assert np.all(done == 1)
    for s in range(G):
        tids, etas, unis = streams[s]
        ref = oz.play_game(game, H, W, k, tt, P, db, mt, oz.fn_ptr("oz_fake_eval"), oz.Rng(tids, etas, unis))
        tr = eng.trajectory_pop(s)
        assert np.array_equal(tr["moves"], ref["moves"]), s
        assert np.array_equal(tr["visits"], ref["visits"]), s
        assert tr["winner"] == ref["winner"], s
    eng.close()


def test_tree_arena_overflow_is_reported():
    from alphazero_amd.engine import Engine, EngineError
    eng = Engine("go", 5, 5, n_slots=1, max_playout=50, transform_types=0, tree_edge_cap=60, tree_block_cap=4)
    eng.search_begin([0], [0], [0])
    with pytest.raises(EngineError) as ei:
        _run_search(eng)
    assert ei.value.code == -4
    eng.close()


def test_tree_arena_overflow_policy_skip_keeps_searching():
    """tree_full_policy 1: the playout whose expansion does not fit backs its value up without expanding; the search
    finishes with one visit per playout and the skipped expansions are counted."""
    from alphazero_amd.engine import Engine
    eng = Engine("go", 5, 5, n_slots=1, max_playout=50, transform_types=0, tree_edge_cap=60, tree_block_cap=4,
                 tree_full_policy=1)
    eng.search_begin([0], [0], [0])
    assert _run_search(eng) == 51
    a, n, w, p, rootN = eng.root_stats(0, 0)
    assert rootN == 51 and n.sum() == 50 and len(a) == 26
    c = eng.counters()
    assert c["expansions_skipped"] > 0 and c["edges_created"] <= 60
    edges, blocks = eng.tree_size(0, 0)
    assert edges <= 60 and blocks <= 4
    eng.close()


def _peaked_eval(planes, O):
    """0.99 on the first empty point of the (transformed) board, the rest spread evenly; value 0.1.  With such priors
    almost every visit goes to one child, whose subtree survives the re-root: the trees keep growing ply after ply."""
    single = planes.ndim == 3
    if single:
        planes = planes[None]
    B = planes.shape[0]
    occ = (planes[:, 14] + planes[:, 15]).reshape(B, -1)
    first = np.argmax(occ == 0, axis=1)
    pri = np.full((B, O), np.float32(0.01) / np.float32(O - 1), np.float32)
    pri[np.arange(B), first] = np.float32(0.99)
    val = np.full(B, np.float32(0.1), np.float32)
    return (pri[0], val[0]) if single else (pri, val)


def test_tree_arenas_grow_and_stay_exact():
    """tree_full_policy 2: arenas far too small for the retained subtrees double as often as needed; the stopped
    playout is completed, so whole games still equal the oracle's (whose trees have no limit), visit for visit."""
    from alphazero_amd.engine import Engine
    game, H, P, db, mt, G = "go", 7, 40, 6, 30, 3
    rng = np.random.RandomState(21)
    eng = Engine(game, H, H, n_slots=G, max_playout=P, transform_types=8, dirichlet_before=db, max_turn=mt, rng_replay=True,
                 record_trajectories=True, tree_edge_cap=150, tree_block_cap=6, tree_full_policy=2)
    O = eng.O
    streams = []
    for s in range(G):
        st = (rng.randint(0, 8, size=(mt + 2) * (P + 1)).astype(np.int32), rng.gamma(0.03, 1.0, size=(mt + 2) * O), rng.rand(mt + 2))
        streams.append(st)
        eng.rng_inject(s, *st)
    eng.selfplay_start()
    while eng.selfplay_select() > 0:
        planes, _ = eng.leaf_batch()
        eng.selfplay_finish(*_peaked_eval(planes, O))
    c = eng.counters()
    assert c["tree_edge_cap"] > 150 and c["tree_block_cap"] > 6 and c["expansions_skipped"] == 0
    assert c["peak_tree_blocks"] > 6
    ev = oz.make_eval(_peaked_eval)
    for s in range(G):
        ref = oz.play_game(game, H, H, 0, 8, P, db, mt, ev, oz.Rng(*streams[s]))
        tr = eng.trajectory_pop(s)
        assert np.array_equal(tr["moves"], ref["moves"]) and np.array_equal(tr["visits"], ref["visits"]) and tr["winner"] == ref["winner"], s
    eng.close()


def test_reroot_keeps_subtree_and_compacts():
    """update_with_move: the kept child's statistics survive, the arena shrinks, a move that is
    not a child gives a fresh root (search/mcts.py:336-346)."""
    from alphazero_amd.engine import Engine
    eng = Engine("go", 5, 5, n_slots=1, max_playout=120, transform_types=0)
    eng.search_begin([0], [0], [0])
    _run_search(eng)
    a, n, w, p, rootN = eng.root_stats(0, 0)
    assert rootN == 121 and n.sum() == 120
    e0, b0 = eng.tree_size(0, 0)
    best = int(np.argmax(n))
    eng.do_move([0], [int(a[best])])
    eng.advance_tree(0, 0, int(a[best]))
    a2, n2, w2, p2, rootN2 = eng.root_stats(0, 0)
    assert rootN2 == n[best] and n2.sum() == n[best] - 1
    e1, b1 = eng.tree_size(0, 0)
    assert e1 < e0 and b1 < b0 and b1 >= 1
    eng.advance_tree(0, 0, 25 if 25 not in a2 else int(a2[0]))
    eng.advance_tree(0, 0, 0)  # whatever it was, walk on: unknown children reset the tree
    eng.close()


@pytest.mark.parametrize("K,vloss", [(4, 1.0), (8, 0.5), (16, 1.0)])
def test_several_leaves_per_round_keep_the_tree_consistent(K, vloss):
    """leaves_per_round > 1 (virtual loss; not the reference's schedule): every playout is still counted once, the
    budget is met exactly, no virtual loss is left behind, and fewer rounds are needed."""
    from alphazero_amd.engine import Engine
    P, H = 203, 9
    eng = Engine("go", H, H, n_slots=2, max_playout=P, transform_types=8, leaves_per_round=K, virtual_loss=vloss, seed=5)
    assert eng.K == K
    eng.search_begin([0, 1], [0, 0], [1, 0])
    rounds = 0
    while eng.step_select() > 0:
        planes, valid = eng.leaf_batch()
        assert planes.shape[0] == K * 2
        pri, v = fake_eval_planes(planes, eng.O)          # values in [-0.5, 0.5)
        eng.step_finish(pri, v)
        rounds += 1
    assert rounds <= 1 + (P + K - 1) // K + 1 and rounds < P
    for slot in (0, 1):
        a, n, w, p, rootN = eng.root_stats(slot, 0)
        assert rootN == P + 1 and n.sum() == P and len(a) == H * H + 1
        assert np.all(np.abs(w) <= 0.5 * n + 1e-4)      # a forgotten virtual loss would leave |W| > N / 2
        best = int(np.argmax(n))
        eng.do_move([slot], [int(a[best])])
        eng.advance_tree(slot, 0, int(a[best]))
        a2, n2, w2, p2, rootN2 = eng.root_stats(slot, 0)
        assert rootN2 == n[best] and n2.sum() == n[best] - 1
        assert np.all(np.abs(w2) <= 0.5 * n2 + 1e-4)
    c = eng.counters()
    assert c["evals"] == 2 * (P + 1)
    eng.close()


@pytest.mark.parametrize("game,H,k,tt,K,vloss,P", [("go", 7, 0, 8, 4, 1.0, 150), ("go", 9, 0, 8, 16, 0.5, 203), ("reversi", 8, 0, 8, 8, 1.0, 96),
                                                   ("mnk", 5, 4, 0, 3, 0.25, 40), ("go", 19, 0, 8, 32, 1.0, 320)])
def test_several_leaves_per_round_against_scalar_restatement(game, H, k, tt, K, vloss, P):
    """leaves_per_round = K under a virtual loss is not the reference's schedule (it keeps one playout in flight); its
    definition is the scalar oz_tree_search_k of the oracle (which with K = 1 is pinned to the reference's trees).  Two
    consecutive searches of one tree with a re-root in between: child order, N, and the bit patterns of W and P."""
    from alphazero_amd.engine import Engine
    rng = np.random.RandomState(K * 100 + H)
    eng = Engine(game, H, H, k, n_slots=2, max_playout=P, transform_types=tt, leaves_per_round=K, virtual_loss=vloss,
                 rng_replay=True, dirichlet_before=30)
    O = eng.O
    tids = rng.randint(0, 8, size=4 * (P + 2)).astype(np.int32)
    etas = rng.gamma(0.03, 1.0, size=4 * O)
    eng.rng_inject(1, tids, etas, [])
    st = oz.State(game, H, H, k)
    tree = oz.Tree(O, tt)
    orng = oz.Rng(tids, etas, [])
    for search in range(2):
        eng.search_begin([1], [0], [1])
        while eng.step_select() > 0:
            planes, valid = eng.leaf_batch()
            assert planes.shape[0] == 2 * K
            pri, v = fake_eval_planes(planes, O)
            eng.step_finish(pri, v)
        tree.search_k(st, P, K, vloss, True, oz.fn_ptr("oz_fake_eval"), orng)
        a, n, w, p, rootN = eng.root_stats(1, 0)
        ra, rn, rw, rp = tree.root_stats()
        assert rootN == tree.root_visits()
        assert np.array_equal(a, ra) and np.array_equal(n, rn)
        assert np.array_equal(w.view(np.uint32), rw.view(np.uint32)) and np.array_equal(p.view(np.uint32), rp.view(np.uint32))
        assert eng.rng_cursor(1)[:2] == (orng.r.i_tids, orng.r.i_etas)
        mv = int(a[int(np.argmax(n))])
        eng.do_move([1], [mv]); eng.advance_tree(1, 0, mv)
        st.do_move(mv); tree.update_with_move(mv)
    eng.close()


def test_selfplay_several_leaves_per_round_against_scalar_restatement():
    """Whole games of the device-side self-play machine at K = 4 against oz_play_game_ex with the same schedule."""
    from alphazero_amd.engine import Engine
    game, H, P, db, mt, G, K, vl = "go", 5, 30, 6, 24, 6, 4, 1.0
    rng = np.random.RandomState(77)
    eng = Engine(game, H, H, n_slots=G, max_playout=P, transform_types=8, dirichlet_before=db, max_turn=mt, rng_replay=True,
                 record_trajectories=True, leaves_per_round=K, virtual_loss=vl)
    O = eng.O
    streams = []
    for s in range(G):
        st = (rng.randint(0, 8, size=(mt + 2) * (P + 1)).astype(np.int32), rng.gamma(0.03, 1.0, size=(mt + 2) * O), rng.rand(mt + 2))
        streams.append(st)
        eng.rng_inject(s, *st)
    eng.selfplay_start()
    _selfplay_injected(eng)
    plies, done = eng.selfplay_status()
    assert np.all(done == 1)
    for s in range(G):
        ref = oz.play_game(game, H, H, 0, 8, P, db, mt, oz.fn_ptr("oz_fake_eval"), oz.Rng(*streams[s]), K=K, vloss=vl)
        tr = eng.trajectory_pop(s)
        assert np.array_equal(tr["moves"], ref["moves"]) and np.array_equal(tr["visits"], ref["visits"]) and tr["winner"] == ref["winner"], s
    eng.close()


def test_selfplay_with_several_leaves_per_round():
    from alphazero_amd.engine import Engine
    from alphazero_amd.network.model import fold_to_blob, random_net
    G, P, K = 4, 30, 8
    eng = Engine("mnk", 3, 3, 3, n_slots=G, max_playout=P, num_blocks=1, channels=128, tower_dtype="bf16",
                 record_trajectories=True, seed=7, leaves_per_round=K)
    eng.tower_load(fold_to_blob(random_net(3, 3, 9, 1, 128, seed=3)))
    eng.selfplay_start()
    eng.selfplay_run(9 * (2 + (P + K - 1) // K) + 5)
    plies, done = eng.selfplay_status()
    assert np.all(done == 1) and np.all(plies >= 5) and np.all(plies <= 9)
    for s in range(G):
        tr = eng.trajectory_pop(s)
        assert np.all(tr["visits"].sum(axis=1) >= P) and tr["winner"] in (-1, 0, 1)
        assert np.all(tr["visits"][0:2].sum(axis=1) == P)   # fresh roots: exactly the budget
    eng.close()


@pytest.mark.parametrize("K", [1, 4])
def test_queued_search_rounds_equal_the_stepwise_loop(K):
    """az_search_run (rounds queued on the device, one synchronisation) against the select / finish loop with a host
    round trip per playout: same kernels in the same order, so the trees must be identical; extra rounds after the
    budget is spent must change nothing."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.network.model import fold_to_blob, random_net
    H, P = 9, 40
    blob = fold_to_blob(random_net(H, H, 82, 1, 128, seed=2, perturb_bn=True))
    stats = []
    for queued in (True, False):
        eng = Engine("go", H, H, n_slots=3, max_playout=P, transform_types=8, seed=17, num_blocks=1, channels=128,
                     tower_dtype="fp32", leaves_per_round=K)
        eng.tower_load(blob)
        eng.do_move([0, 1, 2], [40, 41, 30])
        eng.search_begin([0, 1, 2], [1, 1, 0], [1, 0, 1])
        if queued:
            eng.search_run((P + K - 1) // K + 1)      # root expansion + the playout budget
            eng.search_run(3)                         # nothing left to do: no-ops
        while eng.step_select() > 0:
            eng.step_finish()
        stats.append([eng.root_stats(s, sd) for s, sd in ((0, 1), (1, 1), (2, 0))])
        eng.close()
    for a, b in zip(*stats):
        assert a[4] == b[4] == P + 1
        for x, y in zip(a[:4], b[:4]):
            assert np.array_equal(x, y)
        assert a[1].sum() == P


def test_batched_trajectory_pop_equals_single_pops():
    from alphazero_amd.engine import Engine, EngineError
    G, P = 12, 10
    engs = [Engine("mnk", 3, 3, 3, n_slots=G, max_playout=P, seed=9, record_trajectories=True) for _ in range(2)]
    rng = np.random.RandomState(0)
    for e in engs:
        e.selfplay_start()
    for _ in range(9 * (P + 1) + 3):
        pri = rng.dirichlet(np.ones(9), size=G).astype(np.float32)
        val = rng.uniform(-1, 1, G).astype(np.float32)
        for e in engs:
            e.selfplay_select()
            e.selfplay_finish(pri, val)
    (pa, da), (pb, db) = engs[0].selfplay_status(), engs[1].selfplay_status()
    assert np.array_equal(pa, pb) and np.all(da == 1)
    order = [7, 0, 11, 3]
    many = engs[0].trajectory_pop_many(order, pa[order], restart=True)
    for s, tr in zip(order, many):
        one = engs[1].trajectory_pop(s, restart=True)
        assert np.array_equal(tr["moves"], one["moves"]) and np.array_equal(tr["visits"], one["visits"]) and tr["winner"] == one["winner"]
        assert len(tr["moves"]) == pa[s]
    for e in engs:                                   # the popped slots were re-armed, the others still wait
        plies, done = e.selfplay_status()
        assert all(done[s] == 0 and plies[s] == 0 for s in order) and done.sum() == G - len(order)
    assert engs[0].trajectory_pop_many([], []) == []
    with pytest.raises(EngineError):
        engs[0].trajectory_pop_many([7], [0])        # re-armed: not finished
    with pytest.raises(EngineError):
        engs[0].trajectory_pop_many([1, 2], [pa[1] + 1, pa[2]])   # ply counts that do not match the device
    for e in engs:
        e.close()


def test_search_under_positional_superko_against_oracle():
    """enforce_superko inside the search (every playout state is a copy() that keeps previous_hashes, env/go.py:250-252, so
    leaf legality respects superko): whole games on a 4x4 board, where captured points are replayed all the time, against
    the oracle playing with the same rule (the oracle's superko is pinned to the reference by the rulesko_* fixtures)."""
    from alphazero_amd.engine import Engine
    H, P, db, mt, G = 4, 40, 4, 120, 6
    rng = np.random.RandomState(9)
    eng = Engine("go", H, H, n_slots=G, max_playout=P, transform_types=8, dirichlet_before=db, max_turn=mt, rng_replay=True,
                 record_trajectories=True, enforce_superko=True)
    O = eng.O
    streams = []
    for s in range(G):
        st = (rng.randint(0, 8, size=(mt + 2) * (P + 1)).astype(np.int32), rng.gamma(0.03, 1.0, size=(mt + 2) * O), rng.rand(mt + 2))
        streams.append(st)
        eng.rng_inject(s, *st)
    eng.selfplay_start()
    _selfplay_injected(eng)
    differs = 0
    for s in range(G):
        tr = eng.trajectory_pop(s)
        for superko in (True, False):
            cfg = oz.OzGameCfg(0, H, H, 0, 8, P, db, mt, 0.03, 0.25)
            ref = _oracle_game_superko(cfg, streams[s], superko, mt + 2, O)
            same = np.array_equal(tr["moves"], ref["moves"]) and np.array_equal(tr["visits"], ref["visits"]) and tr["winner"] == ref["winner"]
            if superko:
                assert same, s
            else:
                differs += not same
    assert differs > 0        # the rule mattered in at least one of these games
    eng.close()


def _oracle_game_superko(cfg, stream, superko, max_plies, O):
    """Game.start on the oracle with GameState(enforce_superko=superko): two trees, oz_tree_search per move."""
    st = oz.State("go", cfg.H, cfg.W, enforce_superko=superko)
    trees = [oz.Tree(O, cfg.transform_types, cfg.d_alpha, cfg.d_epsilon) for _ in range(2)]
    rng = oz.Rng(*stream)
    moves, visits, cur = [], [], 0
    while not (st.s.is_end or st.s.turns > cfg.max_turn) and len(moves) < max_plies:
        trees[cur].search(st, cfg.max_playout, st.s.turns <= cfg.dirichlet_before, oz.fn_ptr("oz_fake_eval"), rng)
        a, n, w, p = trees[cur].root_stats()
        row = np.zeros(O, np.int32); row[a] = n
        mv = trees[cur].choose(float(stream[2][rng.r.i_unis])); rng.r.i_unis += 1
        moves.append(mv); visits.append(row)
        assert st.do_move(mv) == 0
        trees[0].update_with_move(mv); trees[1].update_with_move(mv)
        cur ^= 1
    return dict(moves=np.array(moves, np.int16), visits=np.array(visits, np.int32).reshape(-1, O), winner=st.winner())
