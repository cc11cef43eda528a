"""CPU tests: the C oracle against fixtures produced by the reference's own Python
(oracle/gen_golden.py) and against the reference's known-answer unit tests."""
import numpy as np
import pytest

from oracle import oracle as oz
from oracle.fake_eval import fake_eval_planes
from tests.helpers import golden_names, load_golden


@pytest.mark.parametrize("name", golden_names("rules_") + golden_names("rulesko_"))
def test_rules_trace(name):
    """rulesko_*: games the reference played with enforce_superko=True (positional superko, env/go.py:303-330); their
    masks forbid points that the other rules allow (superko_bites of them)."""
    g = load_golden(name)
    superko = bool(g.get("superko", 0))
    st = oz.State(str(g["game"]), int(g["H"]), int(g["W"]), int(g["k"]), enforce_superko=superko)
    if superko and not name.endswith("go7"):
        assert g["superko_bites"] > 0
    fi = 0
    for ply in range(len(g["boards"])):
        assert np.array_equal(st.board(), g["boards"][ply]), (name, ply)
        assert np.array_equal(st.legal_mask(), g["masks"][ply]), (name, ply)
        assert st.s.ko == g["ko"][ply], (name, ply)
        assert st.s.cur == g["cur"][ply]
        assert st.s.turns == g["turns"][ply]
        assert (st.s.pris_b, st.s.pris_w) == tuple(g["pris"][ply])
        assert (st.s.passes_b, st.s.passes_w) == tuple(g["passes"][ply])
        while fi < len(g["feat_ply"]) and g["feat_ply"][fi] == ply:
            assert np.array_equal(st.features(int(g["feat_tid"][fi])), g["feats"][fi]), (name, ply, fi)
            fi += 1
        if ply < len(g["moves"]):
            assert st.do_move(int(g["moves"][ply])) == 0
            assert st.s.is_end == g["is_end"][ply]
    assert fi == len(g["feat_ply"])
    assert np.array_equal(st.board(), g["final_board"])
    assert st.s.ko == g["final_ko"]
    assert st.winner() == g["final_winner"]
    assert st.s.is_end == g["final_is_end"]


def test_illegal_moves_rejected():
    st = oz.State("go", 5, 5)
    assert st.do_move(7) == 0
    before = st.board()
    assert st.do_move(7) == -1  # occupied -> IllegalMove, state untouched
    assert np.array_equal(st.board(), before) and st.s.cur == -1
    m = oz.State("mnk", 3, 3, 3)
    assert m.do_move(9) == -1  # no PASS in mnk
    r = oz.State("reversi", 8, 8)
    assert r.do_move(64) == -1  # PASS illegal while a placement exists
    assert r.do_move(0) == -1


def test_transform_tables():
    g = load_golden("transforms")
    for n in (3, 5, 7, 8, 9, 19):
        lab, fwd = g["labels_%d" % n], g["forward_%d" % n]
        src = np.arange(n * n).astype(np.int8 if n * n < 128 else np.int16)
        for tid in range(8):
            got = [oz.lib().oz_reverse_label(i, n, n, tid) for i in range(n * n + 1)]
            assert got == list(lab[tid])
            # forward board transform through int8 chunks (values < 128 per pass)
            b = (np.arange(n * n) % 120).astype(np.int8)
            out = np.zeros(n * n, np.int8)
            oz.lib().oz_transform_board(b.ctypes.data, out.ctypes.data, n, n, tid)
            assert np.array_equal(out, (fwd[tid] % 120).astype(np.int8))
            # label table == source coordinate of each transformed cell
            assert np.array_equal(fwd[tid], lab[tid][:-1])
        del src


def test_fake_eval_c_matches_numpy():
    rng = np.random.RandomState(0)
    for (H, W, O) in ((3, 3, 9), (8, 8, 65), (19, 19, 362), (6, 5, 30)):
        planes = rng.randint(0, 2, size=(17, H, W)).astype(np.int8)
        pr = np.zeros(O, np.float32)
        v = np.zeros(1, np.float32)
        oz.lib().oz_fake_eval(None, planes.ctypes.data_as(oz.C.POINTER(oz.C.c_int8)), H, W, O,
                              pr.ctypes.data_as(oz.C.POINTER(oz.C.c_float)),
                              v.ctypes.data_as(oz.C.POINTER(oz.C.c_float)))
        p2, v2 = fake_eval_planes(planes, O)
        assert np.array_equal(pr, p2) and v[0] == v2


def _replay_search(g):
    game, H, W, k, O = str(g["game"]), int(g["H"]), int(g["W"]), int(g["k"]), int(g["O"])
    rng = oz.Rng(tids=g["tids"], etas=g["etas"], unis=g["unis"])
    evaluator = str(g["evaluator"]) if "evaluator" in g else "fake32"
    real64 = evaluator != "fake32"            # float64 evaluator -> the float64 build of the oracle
    bits = np.uint64 if real64 else np.uint32
    ev = oz.fn_ptr("oz_uniform_eval" if evaluator == "uniform" else "oz_fake_eval", real64)
    st = oz.State(game, H, W, k)
    trees = [oz.Tree(O, int(g["transform_types"]), float(g["d_alpha"]), float(g["d_epsilon"]), real64) for _ in range(2)]
    cur = 0
    evals = 0
    oz.set_c_punt(float(g["c_punt"]) if "c_punt" in g else 5.0, real64)   # (older fixtures: the reference's default)
    try:
        _replay_plies(g, st, trees, rng, ev, bits)
    finally:
        oz.set_c_punt(5.0, real64)
    assert not rng.r.underflow
    assert np.array_equal(st.board(), g["final_board"])
    return st


def _replay_plies(g, st, trees, rng, ev, bits):
    cur = 0
    evals = 0
    for ply in range(len(g["moves"])):
        assert not (st.s.is_end or st.s.turns > int(g["max_turn"]))
        assert np.array_equal(st.board(), g["boards"][ply])
        dirichlet = st.s.turns <= int(g["dirichlet_before"])
        assert int(dirichlet) == g["dirichlet"][ply]
        evals += trees[cur].search(st, int(g["max_playout"]), dirichlet, ev, rng)
        a, n, w, p = trees[cur].root_stats()
        nc = int(g["n_children"][ply])
        assert len(a) == nc, (ply, len(a), nc)
        assert np.array_equal(a, g["acts"][ply][:nc]), ply
        assert np.array_equal(n, g["N"][ply][:nc]), ply
        assert np.array_equal(w.view(bits), g["Wv"][ply][:nc].view(bits)), ply
        assert np.array_equal(p.view(bits), g["P"][ply][:nc].view(bits)), ply
        assert trees[cur].root_visits() == g["rootN"][ply]
        assert evals == g["cum_evals"][ply]
        assert rng.r.i_tids == g["cum_tids"][ply] and rng.r.i_etas == g["cum_etas"][ply]
        u = float(g["unis"][rng.r.i_unis]); rng.r.i_unis += 1
        mv = trees[cur].choose(u)
        assert mv == g["moves"][ply], ply
        assert st.do_move(mv) == 0
        trees[0].update_with_move(mv)
        trees[1].update_with_move(mv)
        cur ^= 1


@pytest.mark.parametrize("name", golden_names("search_") + golden_names("search64_"))
def test_search_trace(name):
    g = load_golden(name)
    st = _replay_search(g)
    if int(g["ply_limit"]) > len(g["moves"]):  # the game ran to its end
        assert st.winner() == g["winner"]


@pytest.mark.parametrize("name", ["search_ttt_game", "search_reversi_game", "search_ttt_draw"])
def test_play_game_and_history(name):
    """oz_play_game == Game.start; trajectory -> Game.get_history arrays (gameplay.py:73-94)."""
    g = load_golden(name)
    game, H, W, k, O = str(g["game"]), int(g["H"]), int(g["W"]), int(g["k"]), int(g["O"])
    rng = oz.Rng(tids=g["tids"], etas=g["etas"], unis=g["unis"])
    tr = oz.play_game(game, H, W, k, int(g["transform_types"]), int(g["max_playout"]),
                      int(g["dirichlet_before"]), int(g["max_turn"]), oz.fn_ptr("oz_fake_eval"), rng,
                      float(g["d_alpha"]), float(g["d_epsilon"]))
    assert np.array_equal(tr["moves"], g["moves"])
    assert tr["winner"] == g["winner"]
    assert tr["n_evals"] == g["total_evals"]
    T = len(tr["moves"])
    probs = tr["visits"].astype(np.float64) / tr["visits"].sum(axis=1, keepdims=True)
    assert np.array_equal(probs, g["hist_probs"])
    result = np.array([1.0 if (i % 2 == (tr["winner"] != 1)) else -1.0 for i in range(T)])
    assert np.array_equal(result, g["hist_result"])
    st = oz.State(game, H, W, k)
    for i in range(T):
        assert np.array_equal(st.features(0), g["hist_state"][i])
        st.do_move(int(tr["moves"][i]))


@pytest.mark.parametrize("name", golden_names("match_"))
def test_two_evaluator_game(name):
    """Game(nn_eval_1, nn_eval_2, ...) with two different evaluators (train/parallel/evaluator.py:81-85): the oracle's
    oz_play_game_ex against games the reference played, per ply visit counts included."""
    g = load_golden(name)
    game, H, W, k, O = str(g["game"]), int(g["H"]), int(g["W"]), int(g["k"]), int(g["O"])
    assert g["salts"][0] != g["salts"][1]
    rng = oz.Rng(tids=g["tids"], etas=g["etas"], unis=g["unis"])
    tr = oz.play_game(game, H, W, k, int(g["transform_types"]), int(g["max_playout"]), int(g["dirichlet_before"]),
                      int(g["max_turn"]), oz.fn_ptr("oz_fake_eval"), rng, float(g["d_alpha"]), float(g["d_epsilon"]),
                      salts=g["salts"])
    assert np.array_equal(tr["moves"], g["moves"]) and tr["winner"] == g["winner"] and tr["n_evals"] == g["total_evals"]
    for ply in range(len(g["moves"])):
        nc = int(g["n_children"][ply])
        exp = np.zeros(O, np.int32)
        exp[g["acts"][ply][:nc]] = g["N"][ply][:nc]
        assert np.array_equal(tr["visits"][ply], exp), ply
    # the salt matters: the same draws with one evaluator on both sides play a different game
    same = oz.play_game(game, H, W, k, int(g["transform_types"]), int(g["max_playout"]), int(g["dirichlet_before"]),
                        int(g["max_turn"]), oz.fn_ptr("oz_fake_eval"), oz.Rng(seed=3), float(g["d_alpha"]), float(g["d_epsilon"]),
                        salts=(int(g["salts"][0]), int(g["salts"][0])))
    other = oz.play_game(game, H, W, k, int(g["transform_types"]), int(g["max_playout"]), int(g["dirichlet_before"]),
                         int(g["max_turn"]), oz.fn_ptr("oz_fake_eval"), oz.Rng(seed=3), float(g["d_alpha"]), float(g["d_epsilon"]),
                         salts=g["salts"])
    assert not (np.array_equal(same["moves"], other["moves"]) and np.array_equal(same["visits"], other["visits"]))


def test_salted_fake_eval_c_matches_numpy():
    rng = np.random.RandomState(1)
    planes = rng.randint(0, 2, size=(17, 5, 5)).astype(np.int8)
    for salt in (0, 77, 2 ** 63 + 5):
        pr = np.zeros(26, np.float32); v = np.zeros(1, np.float32)
        ctx = oz.C.c_uint64(salt)
        oz.lib().oz_fake_eval(oz.C.byref(ctx), planes.ctypes.data_as(oz.C.POINTER(oz.C.c_int8)), 5, 5, 26,
                              pr.ctypes.data_as(oz.C.POINTER(oz.C.c_float)), v.ctypes.data_as(oz.C.POINTER(oz.C.c_float)))
        p2, v2 = fake_eval_planes(planes, 26, salt)
        assert np.array_equal(pr, p2) and v[0] == v2
    assert not np.array_equal(fake_eval_planes(planes, 26, 0)[0], fake_eval_planes(planes, 26, 77)[0])


@pytest.mark.parametrize("name", ["search_go5_game", "search_reversi_game", "search_ttt_game"])
def test_k_leaf_schedule_with_one_leaf_is_the_reference(name):
    """oz_tree_search_k (the several-leaves-per-round schedule the engine offers) with K = 1 is mcts.py's schedule:
    same trees as the fixtures recorded from the reference, W and P bit for bit."""
    g = load_golden(name)
    game, H, W, k, O = str(g["game"]), int(g["H"]), int(g["W"]), int(g["k"]), int(g["O"])
    rng = oz.Rng(tids=g["tids"], etas=g["etas"], unis=g["unis"])
    st = oz.State(game, H, W, k)
    trees = [oz.Tree(O, int(g["transform_types"]), float(g["d_alpha"]), float(g["d_epsilon"])) for _ in range(2)]
    cur = 0
    for ply in range(len(g["moves"])):
        trees[cur].search_k(st, int(g["max_playout"]), 1, 1.0, st.s.turns <= int(g["dirichlet_before"]), oz.fn_ptr("oz_fake_eval"), rng)
        a, n, w, p = trees[cur].root_stats()
        nc = int(g["n_children"][ply])
        assert np.array_equal(a, g["acts"][ply][:nc]) and np.array_equal(n, g["N"][ply][:nc]), ply
        assert np.array_equal(w.view(np.uint32), g["Wv"][ply][:nc].view(np.uint32)), ply
        assert np.array_equal(p.view(np.uint32), g["P"][ply][:nc].view(np.uint32)), ply
        mv = trees[cur].choose(float(g["unis"][rng.r.i_unis])); rng.r.i_unis += 1
        assert mv == g["moves"][ply]
        st.do_move(mv)
        trees[0].update_with_move(mv); trees[1].update_with_move(mv)
        cur ^= 1


def test_k_leaf_schedule_properties():
    """K > 1: the budget is met exactly, every playout is one visit, no virtual loss is left on any root edge."""
    st = oz.State("go", 7, 7)
    for K, vl in ((4, 1.0), (16, 0.5)):
        t = oz.Tree(50, 8)
        n_ev = t.search_k(st, 203, K, vl, True, oz.fn_ptr("oz_fake_eval"), oz.Rng(seed=9))
        a, n, w, p = t.root_stats()
        assert n_ev == 204 and t.root_visits() == 204 and n.sum() == 203
        assert np.all(np.abs(w) <= 0.5 * n + 1e-4)


# ---- the reference's own known-answer tests, restated on the oracle -------------

def _dummy_policy_eval(planes, O):
    # tests/test_mcts.py:102-112: priors proportional to the flat index, PASS prior 0
    d = np.arange(361, dtype=np.float64)
    d = d / d.sum()
    return np.concatenate([d, [0.0]]).astype(np.float32), 0.0


def test_reference_mcts_known_answers():
    ev = oz.make_eval(_dummy_policy_eval)
    rng = oz.Rng()
    gs = oz.State("go", 19, 19)
    # test_playout (tests/test_mcts.py:53-64)
    t = oz.Tree(362, 0)
    for _ in range(9):
        t.playout(gs, ev, rng)
    a, n, w, p = t.root_stats()
    assert len(a) == 362
    assert n[list(a).index(18 * 19 + 18)] == 1
    assert t.count_expanded() - 1 == 8
    # test_update_with_move (tests/test_mcts.py:88-99)
    t = oz.Tree(362, 0)
    t.search(gs, 1, False, ev, rng)
    mv = t.argmax()
    assert mv == 18 * 19 + 18
    gs.do_move(mv)
    t.update_with_move(mv)
    a, n, w, p = t.root_stats()
    assert len(a) > 0
    t.search(gs, 0, False, ev, rng)  # root.visit() only, no playouts
    # next select is (18,17): emulate select via a 1-playout search and look at the visited child
    t.search(gs, 1, False, ev, rng)
    a, n, w, p = t.root_stats()
    assert a[int(np.argmax(n))] == 18 * 19 + 17


def test_reference_feature_known_answers():
    # tests/test_go_state_converter.py:82-151 hand-coded planes for simple_board()
    gs = oz.State("go", 7, 7)
    for x, y in [(0, 0), (1, 0), (0, 1), (1, 1), (0, 2), (3, 4), (3, 3), (4, 5), (4, 2), (5, 4), (5, 3), (4, 3), (4, 4)]:
        assert gs.do_move(x * 7 + y) == 0
    f = gs.features(0)
    own = [f[2 * i] for i in range(8)]
    opp = [f[2 * i + 1] for i in range(8)]
    white_last = np.zeros((7, 7), np.int8)
    for x, y in [(1, 0), (1, 1), (3, 4), (4, 5), (5, 4)]:
        white_last[x, y] = 1
    black_last = np.zeros((7, 7), np.int8)
    for x, y in [(0, 0), (0, 1), (0, 2), (3, 3), (4, 2), (4, 4), (5, 3)]:
        black_last[x, y] = 1
    assert np.array_equal(own[-1], white_last) and np.array_equal(opp[-1], black_last)
    white_first = np.zeros((7, 7), np.int8)
    for x, y in [(1, 0), (1, 1), (3, 4)]:
        white_first[x, y] = 1
    black_first = np.zeros((7, 7), np.int8)
    for x, y in [(0, 0), (0, 1), (0, 2)]:
        black_first[x, y] = 1
    assert np.array_equal(own[0], white_first) and np.array_equal(opp[0], black_first)
    assert np.all(f[16] == 0)  # white to move
    assert gs.s.ko == 4 * 7 + 3
    # tests/test_transformer.py:90-101 forward o reverse == identity for all ids
    b0 = gs.board()
    for tid in range(8):
        tb = np.zeros(49, np.int8)
        oz.lib().oz_transform_board(b0.ctypes.data, tb.ctypes.data, 7, 7, tid)
        for i in range(49):
            lab = oz.lib().oz_reverse_label(i, 7, 7, tid)
            assert tb[i] == b0[lab // 7, lab % 7]
