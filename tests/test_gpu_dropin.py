"""GPU tests of the drop-in Python modules (env_path / state_converter_path / gameplay_path,
MCTSearch, Player): written like the reference's own unit tests and compared with fixtures the
reference produced under the same seeded random draws."""
import random

import numpy as np
import pytest
import yaml

from oracle.fake_eval import fake_eval_planes
from tests.helpers import load_golden

pytestmark = pytest.mark.gpu


def _cfg(game, H, W, O, tt, k=None):
    import os
    import alphazero_amd
    with open(os.path.join(os.path.dirname(alphazero_amd.__file__), "config", game + ".yaml")) as f:
        c = yaml.safe_load(f)
    c.update(board_height=H, board_width=W, flat_move_output=O, transform_types=tt)
    if k is not None:
        c["k"] = k
    return c


class FakeEvaluator:
    def __init__(self, cfg):
        import alphazero_amd.processing.state_converter as sc
        self.cfg = cfg
        self.conv = sc.StateTensorConverter(cfg)
        self.t2a = sc.TensorActionConverter(cfg)
        self.calls = 0

    def eval(self, state):
        self.calls += 1
        planes = self.conv.state_to_tensor(state)[0]
        pri, v = fake_eval_planes(planes, self.cfg["flat_move_output"])
        return self.t2a.tensor_to_action(pri), v

    def load(self, p): pass
    def save(self, p): pass


def _install_seeded_draws(seed):
    """Same generators the fixture generator fed the reference with."""
    import alphazero_amd.search.mcts as mcts
    nprng = np.random.RandomState(seed)
    pyrng = random.Random(seed)
    saved = (mcts.randint, mcts.random_variate_dirichlet, mcts.weighted_random_choice)
    mcts.randint = lambda n: int(nprng.randint(n))
    mcts.random_variate_dirichlet = lambda alpha, n: [pyrng.gammavariate(alpha, 1) for _ in range(n)]

    def wrc(li):
        p, w = zip(*li)
        return pyrng.choices(p, weights=w)[0]
    mcts.weighted_random_choice = wrc
    return saved


def _restore(saved):
    import alphazero_amd.search.mcts as mcts
    mcts.randint, mcts.random_variate_dirichlet, mcts.weighted_random_choice = saved


@pytest.mark.parametrize("name,seed", [("search_ttt_game", 21), ("search_reversi_game", 24), ("search_ttt_draw", 61)])
def test_game_class_reproduces_reference_game(name, seed):
    g = load_golden(name)
    game, H, W, k, O = str(g["game"]), int(g["H"]), int(g["W"]), int(g["k"]), int(g["O"])
    cfg = _cfg(game, H, W, O, int(g["transform_types"]), k or None)
    if game == "mnk":
        import alphazero_amd.env.mnk as mnk
        mnk.game_config.update(board_height=H, board_width=W, k=k)
    import alphazero_amd.game.gameplay as gameplay
    ext = {"player": {"max_playout": int(g["max_playout"])}, "dirichlet_before": int(g["dirichlet_before"]),
           "log_iter": 10 ** 9, "max_turn": int(g["max_turn"])}
    saved = _install_seeded_draws(seed)
    try:
        ev = FakeEvaluator(cfg)
        gm = gameplay.Game(ev, ev, cfg, ext)
        winner = gm.start()
        s, p, r = gm.get_history()
    finally:
        _restore(saved)
    assert winner == g["winner"]
    assert [gm.state._a2i(m) for m in gm.acts_history] == list(g["moves"])
    assert ev.calls == g["total_evals"]
    assert np.array_equal(s, g["hist_state"].astype(np.float64))
    assert np.array_equal(p, g["hist_probs"])
    assert np.array_equal(r, g["hist_result"])
    assert np.array_equal(gm.state.board, g["final_board"])


def test_assigning_the_module_global_c_punt_works_as_in_the_reference():
    """search/mcts.py:9 is a module global that :88 reads on every call, so a user may assign it.  The fixture was
    recorded from the reference after `mcts.c_punt = 1.25`; the drop-in module must pick the assignment up (made after
    the players exist) and reproduce the trees bit for bit, and go back to 5.0 when it is restored."""
    g = load_golden("search_go9_cpunt125")
    H, W, O = int(g["H"]), int(g["W"]), int(g["O"])
    cfg = _cfg("go", H, W, O, int(g["transform_types"]))
    import alphazero_amd.env.go as go
    import alphazero_amd.search.mcts as mcts
    from alphazero_amd.player.mcts_player import Player
    ev = FakeEvaluator(cfg)
    saved = _install_seeded_draws(91)
    old = mcts.c_punt
    try:
        p1 = Player(ev, cfg, {"max_playout": int(g["max_playout"])})
        p2 = Player(ev, cfg, {"max_playout": int(g["max_playout"])})
        mcts.c_punt = float(g["c_punt"])
        st = go.GameState(size=H)
        cur = p1
        for ply in range(len(g["moves"])):
            move, probs = cur.think(st, st.turns <= int(g["dirichlet_before"]))
            a, n, w, p, rootN = cur.mcts._eng.root_stats(0, 0)
            nc = int(g["n_children"][ply])
            assert np.array_equal(a, g["acts"][ply][:nc]) and np.array_equal(n, g["N"][ply][:nc]), ply
            assert np.array_equal(w.view(np.uint32), g["Wv"][ply][:nc].view(np.uint32)), ply
            assert st._a2i(move) == g["moves"][ply], ply
            st.do_move(move)
            p1.ack(move)
            p2.ack(move)
            cur = p2 if cur is p1 else p1
    finally:
        mcts.c_punt = old
        _restore(saved)
    assert np.array_equal(st.board, g["final_board"])


def test_mcts_known_answers_of_reference_unit_tests():
    """tests/test_mcts.py of the reference: playout/expansion counts, re-root, next selection."""
    import alphazero_amd.env.go as go
    from alphazero_amd.search.mcts import MCTSearch
    cfg = _cfg("go", 19, 19, 362, 0)
    d = np.arange(361, dtype=np.float64)
    d = d / d.sum()

    def policy_value(state):
        moves = [(x, y) for x in range(19) for y in range(19)]
        pol = list(zip(moves, d.astype(np.float32)))
        pol.append((None, np.float32(0)))
        return pol, np.float32(0.0)

    gs = go.GameState()
    m = MCTSearch(policy_value, cfg, max_playout=1)
    move = m.calc_move(gs, prop_exp=False)
    assert move == (18, 18)
    gs.do_move(move)
    m.update_with_move(move)
    root = m._root
    assert len(root._children) > 0 and root._parent is None
    # after one more playout the visited child is the next best prior: (18, 17)
    m.calc_move(gs, prop_exp=False)
    ch = m._root._children
    assert max(ch.items(), key=lambda kv: kv[1].visit_count)[0] == (18, 17)
    # 8 playouts from a fresh tree: 362 children, 8 expansions below the root
    m2 = MCTSearch(policy_value, cfg, max_playout=8)
    m2._calc_move(go.GameState())
    ch = m2._root._children
    assert len(ch) == 362 and ch[(18, 18)].visit_count == 1
    edges, blocks = m2._eng.tree_size(0, 0)
    assert blocks - 1 == 8


def test_env_go_known_answers():
    """simple_board() of tests/test_go_state_converter.py / test_transformer.py on the env shim."""
    import alphazero_amd.env.go as go
    import alphazero_amd.processing.state_converter as sc
    gs = go.GameState(size=7)
    for mv in [(0, 0), (1, 0), (0, 1), (1, 1), (0, 2), (3, 4), (3, 3), (4, 5), (4, 2), (5, 4), (5, 3), (4, 3), (4, 4)]:
        gs.do_move(mv)
    target = np.asarray([
        [1, 1, 1, 0, 0, 0, 0], [-1, -1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 1, -1, 0, 0],
        [0, 0, 1, 0, 1, -1, 0], [0, 0, 0, 1, -1, 0, 0], [0, 0, 0, 0, 0, 0, 0]])
    assert np.array_equal(gs.board, target)
    assert gs.ko == (4, 3) and not gs.is_legal((4, 3)) and gs.is_legal(None)
    assert gs.current_player == go.WHITE and gs.num_white_prisoners == 1
    with pytest.raises(go.IllegalMove):
        gs.do_move((0, 0))
    c = gs.copy()
    c.transform(1)
    assert np.array_equal(c.board, np.rot90(target, 1))
    assert np.array_equal(gs.board, target)  # the copy is independent
    cfg = _cfg("go", 7, 7, 50, 8)
    pp = sc.StateTensorConverter(cfg, ["board_history"])
    pc = sc.StateTensorConverter(cfg, ["color"])
    pall = sc.StateTensorConverter(cfg)
    t = pall.state_to_tensor(gs)
    assert t.shape == (1, 17, 7, 7) and t.dtype == np.int8
    assert np.array_equal(t[0][:-1], pp.state_to_tensor(gs)[0]) and np.array_equal(t[0][-1:], pc.state_to_tensor(gs)[0])
    assert np.all(pc.state_to_tensor(gs) == 0)
    assert np.array_equal(t[0][14], (target == -1)) and np.array_equal(t[0][15], (target == 1))
    # forward o reverse transform == identity for all ids (tests/test_transformer.py:90-101)
    rt = sc.ReverseTransformer(cfg).reverse_transform
    for tid in range(8):
        g2 = gs.copy()
        g2.transform(tid)
        b = g2.board
        ap = [((i, j), b[i][j]) for i in range(7) for j in range(7)] + [(None, 0)]
        rt(ap, tid)
        for (x, y), p in ap[:49]:
            assert p == target[x][y]
    with pytest.raises(ValueError):
        sc.StateTensorConverter(cfg, ["nope"])


def test_env_reversi_and_mnk_basics():
    import alphazero_amd.env.mnk as mnk
    import alphazero_amd.env.reversi as rv
    r = rv.GameState()
    assert sorted(r.get_legal_moves()) == [(2, 3), (3, 2), (4, 5), (5, 4)]
    assert not r.is_legal(None)
    assert r.do_move((2, 3)) is False
    assert r.board[3][3] == rv.BLACK and r.stones_played == 5 and r.current_player == rv.WHITE
    mnk.game_config.update(board_height=3, board_width=3, k=3)
    m = mnk.GameState()
    for mv in [(0, 0), (1, 0), (0, 1), (1, 1)]:
        assert m.do_move(mv) is False
    assert m.get_winner() is None
    assert m.do_move((0, 2)) is True
    assert m.get_winner() == mnk.BLACK and m.is_end_of_game and m.copy().winner == mnk.BLACK
    with pytest.raises(mnk.IllegalMove):
        m.do_move((0, 0))
    mnk.game_config.update(board_height=6, board_width=5, k=4)


def test_nn_evaluator_player_and_batched_selfplay():
    """NNEvaluator contract + device-resident search through Player, and BatchedSelfplay histories."""
    import alphazero_amd.env.mnk as mnk
    from alphazero_amd.evaluator.nn_eval import NNEvaluator
    from alphazero_amd.game.gameplay import BatchedSelfplay
    from alphazero_amd.player.mcts_player import Player
    from oracle import tower_ref
    import torch
    cfg = _cfg("mnk", 3, 3, 9, 0, 3)
    mnk.game_config.update(board_height=3, board_width=3, k=3)
    ev = NNEvaluator(cfg, num_blocks=1, channels=128, dtype="fp32", max_batch_size=4, seed=5)
    st = mnk.GameState()
    st.do_move((1, 1))
    pol, v = ev.eval(st)
    assert len(pol) == 9 and pol[0][0] == (0, 0)
    rp, rv = tower_ref.forward(ev.net.state_dict(), torch.from_numpy(st.features(0)).float()[None], 1)
    assert np.abs(np.array([p for _, p in pol]) - rp[0].numpy()).max() < 1e-5 and abs(v - float(rv[0])) < 1e-5
    ev.rwlock.r_acquire(); ev.rwlock.r_release()
    pl = Player(ev, cfg, {"max_playout": 30})
    move, probs = pl.think(st, True)
    assert st.is_legal(move) and abs(sum(p for _, p in probs) - 1.0) < 1e-9 and len(probs) == 8
    pl.ack(move)
    bs = BatchedSelfplay(ev, cfg, {"player": {"max_playout": 20}, "dirichlet_before": 30, "max_turn": 400}, n_games=8, seed=3)
    hist = bs.run(10)
    assert len(hist) == 10
    for s, p, r in hist:
        T = s.shape[0]
        assert 5 <= T <= 9 and s.shape == (T, 17, 3, 3) and p.shape == (T, 9) and r.shape == (T,)
        assert np.allclose(p.sum(axis=1), 1.0) and set(np.unique(r)) <= {-1.0, 1.0}
        assert np.all(s[0][:16] == 0) and np.all(s[0][16] == 1)
    mnk.game_config.update(board_height=6, board_width=5, k=4)


def test_dummy_evaluator_search_runs_in_float64_like_the_reference():
    """DummyEvaluator returns np.float64 priors and the Python int 0 (evaluator/dummy_eval.py:26-28), which makes
    every N/W/P intermediate of the reference's search a double: the drop-in MCTSearch must notice and reproduce the
    reference's trace (fixture search64_go9_uniform, seed 31) bit for bit."""
    g = load_golden("search64_go9_uniform")
    H, W, O = int(g["H"]), int(g["W"]), int(g["O"])
    cfg = _cfg("go", H, W, O, int(g["transform_types"]))
    import alphazero_amd.env.go as go
    from alphazero_amd.evaluator.dummy_eval import DummyEvaluator
    from alphazero_amd.player.mcts_player import Player
    ev = DummyEvaluator(cfg)
    saved = _install_seeded_draws(31)
    try:
        p1 = Player(ev, cfg, {"max_playout": int(g["max_playout"])})
        p2 = Player(ev, cfg, {"max_playout": int(g["max_playout"])})
        st = go.GameState(size=H)
        cur = p1
        for ply in range(len(g["moves"])):
            move, probs = cur.think(st, st.turns <= int(g["dirichlet_before"]))
            eng = cur.mcts._eng
            assert eng.real is np.float64
            a, n, w, p, rootN = eng.root_stats(0, 0)
            nc = int(g["n_children"][ply])
            assert np.array_equal(a, g["acts"][ply][:nc]) and np.array_equal(n, g["N"][ply][:nc]), ply
            assert np.array_equal(w.view(np.uint64), g["Wv"][ply][:nc].view(np.uint64)), ply
            assert np.array_equal(p.view(np.uint64), g["P"][ply][:nc].view(np.uint64)), ply
            assert st._a2i(move) == g["moves"][ply], ply
            st.do_move(move)
            p1.ack(move)
            p2.ack(move)
            cur = p2 if cur is p1 else p1
    finally:
        _restore(saved)
    assert np.array_equal(st.board, g["final_board"])


def test_batched_history_replay_matches_single_game_replay():
    """history_arrays_batch (lockstep replay of many games) == history_arrays (one game at a time), ragged lengths."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.game.gameplay import history_arrays, history_arrays_batch
    from oracle import oracle as oz
    rng = np.random.RandomState(4)
    games = []
    for g in range(5):
        st = oz.State("go", 7, 7)
        moves, visits = [], []
        for _ in range(rng.randint(0, 25)):
            legal = np.flatnonzero(st.legal_mask())
            a = int(rng.choice(legal))
            v = np.zeros(50, np.int32); v[legal] = rng.randint(1, 9, len(legal))
            moves.append(a); visits.append(v)
            if st.do_move(a) != 0 or st.s.is_end:
                break
        games.append((np.array(moves, np.int16), np.array(visits, np.int32).reshape(len(moves), 50), int(rng.choice([-1, 0, 1]))))
    eb = Engine("go", 7, 7, n_slots=8)
    e1 = Engine("go", 7, 7, n_slots=1)
    batch = history_arrays_batch(eb, games)
    for (m, v, w), (s, p, r) in zip(games, batch):
        s1, p1, r1 = history_arrays(e1, m, v, w)
        assert np.array_equal(s, s1) and np.array_equal(p, p1) and np.array_equal(r, r1)
    eb.close(); e1.close()


class _PeakedEvaluator:
    """0.99 on the first empty point: the chosen child keeps nearly all visits, so a re-rooted tree keeps nearly
    everything and grows by about max_playout nodes per ply."""

    def __init__(self, cfg):
        import alphazero_amd.processing.state_converter as sc
        self.O = cfg["flat_move_output"]
        self.t2a = sc.TensorActionConverter(cfg)

    def eval(self, state):
        b = np.asarray(state.board).ravel()
        pri = np.full(self.O, np.float32(0.01) / np.float32(self.O - 1), np.float32)
        pri[int(np.argmax(b == 0))] = np.float32(0.99)
        return self.t2a.tensor_to_action(pri), np.float32(0.1)


def test_search_tree_outgrows_the_default_arena_without_error():
    """The reference's tree has no size limit.  A sharply peaked policy played for 35 plies with tree reuse fills the
    drop-in MCTSearch's default arenas several times over; they grow and the game goes on."""
    import alphazero_amd.env.go as go
    from alphazero_amd.player.mcts_player import Player
    H, P = 9, 60
    cfg = _cfg("go", H, H, H * H + 1, 0)       # no transforms: both players' trees peak on the same points
    ev = _PeakedEvaluator(cfg)
    p1, p2 = Player(ev, cfg, {"max_playout": P}), Player(ev, cfg, {"max_playout": P})
    eng = p1.mcts._eng
    cap0 = eng.counters()["tree_block_cap"]
    st = go.GameState(size=H)
    cur = p1
    for ply in range(35):
        move, probs = cur.think(st, False)
        assert abs(sum(p for _, p in probs) - 1.0) < 1e-9
        st.do_move(move)
        p1.ack(move); p2.ack(move)
        cur = p2 if cur is p1 else p1
    c = eng.counters()
    assert c["tree_block_cap"] > cap0 and c["peak_tree_blocks"] > cap0 and c["expansions_skipped"] == 0


def test_optimizer_checkpoint_loads_into_the_evaluator(tmp_path):
    """Weights reach the evaluators through a file: the optimizer's Network.save('<dir>/ckpt') writes 'ckpt-<step>' and
    the evaluator loads that path (train/parallel/optimization.py:99-101, evaluator.py:109).  Both sides must then give
    the same response."""
    import alphazero_amd.env.mnk as mnk
    from alphazero_amd.evaluator.nn_eval import NNEvaluator
    from alphazero_amd.network.train import Network
    cfg = _cfg("mnk", 3, 3, 9, 0, 3)
    mnk.game_config.update(board_height=3, board_width=3, k=3)
    train = {"num_blocks": 1, "batch_decay": 0.9, "learning_rate": {0: 1e-2, 20: 1e-2, 40: 1e-3, 60: 1e-4, 70: 1e-4}, "momentum": 0.9,
             "l2": 1e-4, "MSE_scaling": 1.0, "save_dir": str(tmp_path)}
    nw = Network(cfg, train_config=train, device="cpu", channels=128)
    rng = np.random.RandomState(0)
    data = ((rng.rand(6, 17, 3, 3) < 0.3).astype(np.float64), np.eye(9)[rng.randint(0, 9, 6)], rng.choice([-1.0, 1.0], 6))
    for _ in range(3):
        nw.update(data)
    path = nw.save(str(tmp_path / "ckpt"))
    ev = NNEvaluator(cfg, num_blocks=1, channels=128, dtype="fp32", max_batch_size=8, seed=9)
    before = ev.eval_batch(data[0].astype(np.int8))[0]
    v0 = ev.version
    ev.load(path)
    assert ev.version == v0 + 1
    p, v = ev.eval_batch(data[0].astype(np.int8))
    rp, rv = nw.response(data)
    assert np.abs(p - rp).max() < 1e-5 and np.abs(v - rv).max() < 1e-5 and np.abs(p - before).max() > 1e-4
    # and back: the evaluator's own file loads into the training side (weights only)
    ev.save(str(tmp_path / "eval.pt"))
    nw2 = Network(cfg, train_config=train, device="cpu", channels=128)
    nw2.load(str(tmp_path / "eval.pt"))
    assert np.array_equal(nw2.response(data)[0], rp)
    mnk.game_config.update(board_height=6, board_width=5, k=4)


def test_batched_selfplay_reproduces_the_oracle_games():
    """BatchedSelfplay.run (the role of Selfplay.run + selfplay_wrapper, train/parallel/selfplay.py:71-108) with replayed
    draws against the oracle's oz_play_game fed by the same network: identical get_history() tuples for 8 games.
    The oracle evaluates its leaves one by one on the same HIP tower (float32; every board is computed independently
    of its batch neighbours, so single-leaf and batched evaluations agree bit for bit)."""
    import alphazero_amd.env.reversi  # noqa: F401  (env_path of the config)
    from alphazero_amd.evaluator.nn_eval import NNEvaluator
    from alphazero_amd.game.gameplay import BatchedSelfplay, outcome_labels
    from oracle import oracle as oz
    H, P, db, mt, G = 6, 16, 5, 400, 8
    cfg = _cfg("reversi", H, H, H * H + 1, 8)
    O = cfg["flat_move_output"]
    ev = NNEvaluator(cfg, num_blocks=1, channels=128, dtype="fp32", max_batch_size=4, seed=4)
    rng = np.random.RandomState(5)
    n_plies = 2 * (H * H + 8)          # room for a second game in every slot (slots are re-armed when popped)
    streams = [(rng.randint(0, 8, size=n_plies * (P + 1)).astype(np.int32), rng.gamma(0.03, 1.0, size=n_plies * O), rng.rand(n_plies))
               for _ in range(G)]
    sp = BatchedSelfplay(ev, cfg, {"player": {"max_playout": P}, "dirichlet_before": db, "max_turn": mt}, n_games=G,
                         rng_streams=streams)
    hist = sp.run(G, rounds_per_poll=100000)      # one poll: every slot finishes its first game before any is popped
    assert sorted(sp.last_slots) == list(range(G)) and sp.eng.counters()["expansions_skipped"] == 0

    def one_leaf(planes, O_):
        p, v = ev.eval_batch(np.ascontiguousarray(planes)[None])
        return p[0], v[0]
    cb = oz.make_eval(one_leaf)
    for slot, (state_np, probs_np, result_np) in zip(sp.last_slots, hist):
        ref = oz.play_game("reversi", H, H, 0, 8, P, db, mt, cb, oz.Rng(*streams[slot]))
        T = len(ref["moves"])
        st = oz.State("reversi", H, H)
        ref_state = np.zeros((T, 17, H, H))
        for i in range(T):
            ref_state[i] = st.features(0)
            st.do_move(int(ref["moves"][i]))
        v = ref["visits"].astype(np.float64)
        assert state_np.shape == ref_state.shape and np.array_equal(state_np, ref_state), slot
        assert np.array_equal(probs_np, v / v.sum(axis=1, keepdims=True)), slot
        assert np.array_equal(result_np, outcome_labels(T, ref["winner"])), slot


def test_batched_selfplay_keeps_surplus_games_and_drains_for_new_weights():
    """Games that finish beyond a run()'s request are handed out by the next call, not dropped; update='drain' lets
    every game in flight end under the old network before the new one is loaded (the reference's reload waits for the
    read locks that running games hold, selfplay.py:76,87,117-122)."""
    import alphazero_amd.env.mnk as mnk
    from alphazero_amd.evaluator.nn_eval import NNEvaluator
    from alphazero_amd.game.gameplay import BatchedSelfplay
    from alphazero_amd.network.model import random_net
    cfg = _cfg("mnk", 3, 3, 9, 0, 3)
    mnk.game_config.update(board_height=3, board_width=3, k=3)
    ev = NNEvaluator(cfg, num_blocks=1, channels=128, dtype="fp32", max_batch_size=4, seed=5)
    ext = {"player": {"max_playout": 10}, "dirichlet_before": 30, "max_turn": 400}
    sp = BatchedSelfplay(ev, cfg, ext, n_games=16, seed=3, update="drain")
    first = sp.run(3, rounds_per_poll=11 * 9 + 5)          # one poll finishes all 16 games; 3 are asked for
    assert len(first) == 3 and len(sp._surplus) == 13
    finished_before = sp.eng.counters()["games_finished"]
    second = sp.run(13)                                     # served from the kept games: nothing new is played
    assert len(second) == 13 and sp.eng.counters()["games_finished"] == finished_before and not sp._surplus
    v_old = sp._version
    ev.set_weights(random_net(3, 3, 9, 1, 128, seed=77).state_dict())
    third = sp.run(20, rounds_per_poll=7)                   # short polls: slots end at different times and wait
    assert len(third) == 20 and sp._version == ev.version != v_old and not sp._draining
    for s, p, r in first + second + third:
        assert 5 <= s.shape[0] <= 9 and np.allclose(p.sum(axis=1), 1.0)
    mnk.game_config.update(board_height=6, board_width=5, k=4)


def test_play_cmd_shaped_game_on_the_device_tower():
    """The loop of play_cmd.py:64-108 (raw network player aside): GameState(), two MCTS players on a network evaluator,
    think(state) called with ONE positional argument, the board read as a 2-D array of {1, -1, 0} for printing
    (play_cmd.py:55-61,86), do_move, both players ack, get_winner at the end - with positional superko switched on as the
    GTP front end does (interface/gtp_wrapper.py:88), on a 5x5 board so that the game ends within the test."""
    import alphazero_amd.env.go as go
    from alphazero_amd.evaluator.nn_eval import NNEvaluator
    from alphazero_amd.player.mcts_player import Player
    H = 5
    cfg = _cfg("go", H, H, H * H + 1, 8)
    ev = NNEvaluator(cfg, num_blocks=1, channels=128, dtype="bf16", max_batch_size=4, seed=3)
    players = [Player(ev, cfg, {"max_playout": 24}), Player(ev, cfg, {"max_playout": 24})]
    state = go.GameState(size=H, enforce_superko=True)
    seen = set()
    cur = 0
    for ply in range(60):
        if state.is_end_of_game:
            break
        rows = ["".join({1: "X", -1: "O", 0: "."}[int(v)] for v in row) for row in state.board]   # print_board reads state.board
        assert len(rows) == H and all(len(r) == H for r in rows)
        move, probs = players[cur].think(state)                 # one positional argument, tuple back
        assert state.is_legal(move) and abs(sum(p for _, p in probs) - 1.0) < 1e-9
        state.do_move(move)
        if move is not go.PASS_MOVE:
            key = np.asarray(state.board).tobytes()
            assert key not in seen                              # positional superko: no whole-board repetition
            seen.add(key)
        for p in players:
            p.ack(move)
        cur ^= 1
    assert state.get_winner() in (go.BLACK, go.WHITE, 0)
    assert players[0].mcts._eng.cfg.enforce_superko == 1
    c = state.copy()
    assert c.enforce_superko and np.array_equal(c.legal_mask(), state.legal_mask())


def test_device_trajectories_through_replay_files_and_ring(tmp_path):
    """Games the device played, written the way DataPool stores them (one np.savez_compressed per game with keys
    arr_0..2, datapool.py:65-66), read back the way it reloads them (datapool.py:54-58) and merged into the replay ring:
    the arrays a reference optimizer would sample are the get_history() tuples the engine produced."""
    import alphazero_amd.env.mnk as mnk
    from alphazero_amd.data.replay import ReplayPool, load_games, save_game
    from alphazero_amd.evaluator.nn_eval import NNEvaluator
    from alphazero_amd.game.gameplay import BatchedSelfplay
    cfg = _cfg("mnk", 3, 3, 9, 0, 3)
    mnk.game_config.update(board_height=3, board_width=3, k=3)
    ev = NNEvaluator(cfg, num_blocks=1, channels=128, dtype="bf16", max_batch_size=4, seed=5)
    sp = BatchedSelfplay(ev, cfg, {"player": {"max_playout": 12}, "dirichlet_before": 30, "max_turn": 400}, n_games=12, seed=1)
    hist = sp.run(12)
    store = str(tmp_path / "pool")
    pool = ReplayPool({"pool_size": 40, "start_data_size": 10, "store_path": store, "load_prev": False})
    for h in hist:
        pool.put(h)                                            # ring + '<count>.npz'
    import os
    files = sorted(os.listdir(store), key=lambda f: int(f.split(".")[0]))
    assert files == ["%d.npz" % i for i in range(12)]
    with np.load(os.path.join(store, "3.npz")) as z:
        assert sorted(z.files) == ["arr_0", "arr_1", "arr_2"]
        assert z["arr_0"].dtype == np.float64 and z["arr_0"].shape[1:] == (17, 3, 3) and z["arr_1"].shape[1] == 9
        for k, a in enumerate(hist[3]):
            assert np.array_equal(z["arr_%d" % k], a)
    back = {a.tobytes() + b.tobytes() + c.tobytes() for a, b, c in load_games(store)}
    assert back == {a.tobytes() + b.tobytes() + c.tobytes() for a, b, c in hist}
    # the ring holds the rows in arrival order, wrapped at the capacity (datapool.py:87-112)
    rows = np.concatenate([h[1] for h in hist])
    n = len(rows)
    assert pool.is_full == (n > 40) and pool.training_started
    last = rows[-40:] if n >= 40 else rows
    ring = np.roll(pool.data_pool[1], -pool.pool_end_index, axis=0) if n > 40 else pool.data_pool[1][:n]
    assert np.array_equal(ring[-len(last):], last)
    s_, p_, r_ = pool.get(16)
    assert s_.shape == (16, 17, 3, 3) and np.allclose(p_.sum(axis=1), 1.0) and set(np.unique(r_)) <= {-1.0, 1.0}
    # a fresh pool with load_prev picks the files up again (datapool.py:50-58)
    pool2 = ReplayPool({"pool_size": 400, "start_data_size": 10, "store_path": store, "load_prev": True})
    assert pool2.count == 12 and pool2.pool_end_index == n
    mnk.game_config.update(board_height=6, board_width=5, k=4)
