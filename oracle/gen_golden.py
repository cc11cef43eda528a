"""TEST INFRASTRUCTURE - NOT PRODUCT CODE.

Generates tests/golden/*.npz by running the reference's own Python
(/root/reference, read-only) for rules, features, transforms, search and the
game loop.  Only this script touches the reference; the fixtures it writes
are plain data (inputs + expected outputs) and are what travels to the GPU
box.  Run from anywhere:

    python oracle/gen_golden.py

Harness-side shims (the reference is untouched): np.int / np.float aliases
that NumPy 2 removed, and a default Loader for yaml.load.
"""
import os
import random
import re
import sys

import numpy as np
import yaml

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

np.int = int
np.float = float
_yl = yaml.load
yaml.load = lambda f, Loader=yaml.SafeLoader: _yl(f, Loader=Loader)

import AlphaZero.env.go as go  # noqa: E402
import AlphaZero.env.mnk as mnk  # noqa: E402
import AlphaZero.env.reversi as reversi  # noqa: E402
import AlphaZero.game.gameplay as gameplay  # noqa: E402
import AlphaZero.player.mcts_player as mcts_player  # noqa: E402
import AlphaZero.processing.state_converter as sc  # noqa: E402
import AlphaZero.search.mcts as mcts  # noqa: E402

from fake_eval import fake_eval_planes  # noqa: E402

gameplay.printlog = lambda *a, **k: None


def load_cfg(name):
    with open(os.path.join(REF, "AlphaZero", "config", name + ".yaml")) as f:
        return yaml.load(f)


def cfg_for(game, H, W, transform_types, k=None):
    base = {"go": "go", "reversi": "reversi", "mnk": "mnk"}[game]
    c = dict(load_cfg(base))
    c["board_height"], c["board_width"] = H, W
    c["flat_move_output"] = H * W + (0 if game == "mnk" else 1)
    c["transform_types"] = transform_types
    if k is not None:
        c["k"] = k
    return c


def a2i(action, H, W):
    return H * W if action is None else action[0] * W + action[1]


def i2a(idx, H, W):
    return None if idx == H * W else (idx // W, idx % W)


def make_state(game, H, W, k=None, superko=False):
    if game == "go":
        return go.GameState(size=H, enforce_superko=superko)
    if game == "reversi":
        return reversi.GameState(size=H)
    mnk.game_config.update(board_height=H, board_width=W, k=k)
    return mnk.GameState()


def legal_mask(state, game, H, W):
    O = H * W + (0 if game == "mnk" else 1)
    m = np.zeros(O, np.uint8)
    for i in range(H * W):
        m[i] = state.is_legal((i // W, i % W))
    if game != "mnk":
        m[H * W] = state.is_legal(None)
    return m


def ko_idx(state, W):
    ko = getattr(state, "ko", None)
    return -1 if ko is None else ko[0] * W + ko[1]


# ---------------------------------------------------------------------------
# rules traces
# ---------------------------------------------------------------------------

def rules_trace(game, H, W, k, seed, max_plies, pass_prob=0.05, moves=None, feat_every=7, superko=False):
    """Random legal playout (or a given move list) through the reference rules."""
    rng = random.Random(seed)
    st = make_state(game, H, W, k, superko)
    bites = 0
    conv = sc.StateTensorConverter(cfg_for(game, H, W, 8, k))
    rec = dict(moves=[], boards=[], masks=[], ko=[], cur=[], is_end=[], pris=[], passes=[], turns=[])
    feats, feat_ply, feat_tid = [], [], []
    ply = 0
    while not st.is_end_of_game and ply < max_plies:
        mask = legal_mask(st, game, H, W)
        if superko:   # how often positional superko (go.py:303-330) forbids what the other rules allow
            st.enforce_superko = False
            bites += int((legal_mask(st, game, H, W) != mask).sum())
            st.enforce_superko = True
        rec["masks"].append(mask)
        rec["boards"].append(np.array(st.board, dtype=np.int8))
        rec["ko"].append(ko_idx(st, W))
        rec["cur"].append(st.current_player)
        rec["pris"].append((getattr(st, "num_black_prisoners", 0), getattr(st, "num_white_prisoners", 0)))
        rec["passes"].append((getattr(st, "passes_black", 0), getattr(st, "passes_white", 0)))
        rec["turns"].append(st.turns)
        if ply % feat_every == 0:
            tids = range(8) if H == W else []
            for tid in tids:
                c = st.copy()
                c.transform(tid)
                feats.append(conv.state_to_tensor(c)[0])
                feat_ply.append(ply)
                feat_tid.append(tid)
            if H != W:  # transform() asserts a square board (mnk.py:199)
                feats.append(conv.state_to_tensor(st)[0])
                feat_ply.append(ply)
                feat_tid.append(0)
        if moves is not None:
            if ply >= len(moves):
                break
            mv = moves[ply]
        else:
            legal = [i for i in range(H * W) if mask[i]]
            if game == "go":
                mv = H * W if (not legal or rng.random() < pass_prob) else rng.choice(legal)
            elif game == "reversi":
                mv = rng.choice(legal) if legal else H * W
            else:
                mv = rng.choice(legal)
        end = st.do_move(i2a(mv, H, W))
        rec["moves"].append(mv)
        rec["is_end"].append(int(end))
        ply += 1
    w = st.get_winner()
    out = dict(
        game=game, H=H, W=W, k=k or 0,
        moves=np.array(rec["moves"], np.int16), boards=np.array(rec["boards"], np.int8),
        masks=np.array(rec["masks"], np.uint8), ko=np.array(rec["ko"], np.int16),
        cur=np.array(rec["cur"], np.int8), is_end=np.array(rec["is_end"], np.int8),
        pris=np.array(rec["pris"], np.int32), passes=np.array(rec["passes"], np.int32),
        turns=np.array(rec["turns"], np.int32),
        final_board=np.array(st.board, np.int8), final_ko=ko_idx(st, W),
        final_winner=2 if w is None else int(w), final_is_end=int(st.is_end_of_game),
        feats=np.array(feats, np.int8), feat_ply=np.array(feat_ply, np.int32),
        feat_tid=np.array(feat_tid, np.int32), superko=int(superko), superko_bites=bites,
    )
    return out


def sgf_moves(path):
    """Main line of an SGF as flat indices (x*19+y), colour sequence checked by caller."""
    txt = open(path).read()
    out = []
    for col, xy in re.findall(r";\s*([BW])\[([a-s]{0,2})\]", txt):
        if len(xy) < 2:
            out.append((col, 361))
        else:
            out.append((col, (ord(xy[0]) - 97) * 19 + (ord(xy[1]) - 97)))
    return out


# ---------------------------------------------------------------------------
# search / gameplay traces
# ---------------------------------------------------------------------------

class Streams:
    def __init__(self, seed):
        self.tids, self.etas, self.unis = [], [], []
        self.nprng = np.random.RandomState(seed)
        self.pyrng = random.Random(seed)


class FakeEvaluator:
    """eval(state) -> ([(action, np.float32)], np.float32): the evaluator contract of
    evaluator/nn_eval_parallel.py:75-92 with the network replaced by a hash."""

    def __init__(self, cfg, real64=None, salt=0):
        self.cfg = cfg
        self.salt = salt      # != 0: a second, different "network" (two-evaluator games)
        self.conv = sc.StateTensorConverter(cfg)
        self.t2a = sc.TensorActionConverter(cfg)
        self.calls = 0
        self.real64 = real64  # None: float32 outputs; "fake": the same hash widened to float64; "uniform": DummyEvaluator

    def eval(self, state):
        self.calls += 1
        planes = self.conv.state_to_tensor(state)[0]
        if self.real64 == "uniform":  # evaluator/dummy_eval.py:26-28: np.full float64 priors, Python int 0
            dims = self.cfg["flat_move_output"]
            return self.t2a.tensor_to_action(np.full((dims,), 1 / dims)), 0
        pri, v = fake_eval_planes(planes, self.cfg["flat_move_output"], self.salt)
        assert pri.dtype == np.float32 and isinstance(v, np.float32)
        if self.real64 == "fake":
            return self.t2a.tensor_to_action(pri.astype(np.float64)), float(v)
        return self.t2a.tensor_to_action(pri), v


def install_rng_hooks(streams):
    def _randint(n):
        v = int(streams.nprng.randint(n))
        streams.tids.append(v)
        return v

    orig_dir = sys.modules["AlphaZero.search.math_helper"].random_variate_dirichlet

    def _dirichlet(alpha, n):
        random._inst.random = streams.pyrng.random
        try:
            v = orig_dir(alpha, n)
        finally:
            del random._inst.random
        streams.etas.extend(v)
        return v

    orig_wrc = sys.modules["AlphaZero.search.math_helper"].weighted_random_choice

    def _wrc(li):
        u = streams.pyrng.random()
        streams.unis.append(u)
        random._inst.random = lambda: u
        try:
            return orig_wrc(li)
        finally:
            del random._inst.random

    mcts.randint = _randint
    mcts.random_variate_dirichlet = _dirichlet
    mcts.weighted_random_choice = _wrc


def root_record(search, H, W, O, real64=False):
    ch = search._root.children
    acts = np.full(O, -1, np.int32)
    N = np.zeros(O, np.int32)
    rt = np.float64 if real64 else np.float32
    Wv = np.zeros(O, rt)
    P = np.zeros(O, rt)
    for i, (a, nd) in enumerate(ch.items()):
        acts[i] = a2i(a, H, W)
        N[i] = nd._visit_cnt
        w = nd._total_action_val
        if real64:
            assert isinstance(w, (float, np.float64, int)), type(w)
            assert isinstance(nd._prior_prob, (float, np.float64)), type(nd._prior_prob)
        else:
            assert isinstance(w, (np.float32, int)), type(w)
            assert isinstance(nd._prior_prob, np.float32), type(nd._prior_prob)
        Wv[i] = rt(w)
        P[i] = nd._prior_prob
    return len(ch), acts, N, Wv, P, search._root._visit_cnt


def search_trace(name, game, H, W, k, transform_types, max_playout, dirichlet_before, max_turn,
                 seed, ply_limit, use_game_class=False, real64=None, salts=None):
    cfg = cfg_for(game, H, W, transform_types, k)
    O = cfg["flat_move_output"]
    streams = Streams(seed)
    install_rng_hooks(streams)
    ev = FakeEvaluator(cfg, real64, salts[0] if salts else 0)
    # Game(nn_eval_1, nn_eval_2, ...) with two different evaluators: train/parallel/evaluator.py:81-85
    ev2 = FakeEvaluator(cfg, real64, salts[1]) if salts else ev
    ext = {"player": {"max_playout": max_playout}, "dirichlet_before": dirichlet_before,
           "log_iter": 10 ** 9, "max_turn": max_turn}
    recs = []
    if game == "mnk":
        mnk.game_config.update(board_height=H, board_width=W, k=k)

    def think_hook(orig):
        def _think(self, state, dirichlet=False):
            move, probs = orig(self, state, dirichlet)
            n, acts, N, Wv, P, rootN = root_record(self.mcts, H, W, O, real64 is not None)
            recs.append(dict(n=n, acts=acts, N=N, W=Wv, P=P, rootN=rootN, move=a2i(move, H, W),
                             dirichlet=int(bool(dirichlet)), n_tids=len(streams.tids),
                             n_etas=len(streams.etas), n_unis=len(streams.unis),
                             board=np.array(state.board, np.int8),
                             evals=ev.calls + (ev2.calls if ev2 is not ev else 0)))
            return move, probs
        return _think

    orig_think = mcts_player.Player.think
    mcts_player.Player.think = think_hook(orig_think)
    hist = None
    try:
        if use_game_class:
            g = gameplay.Game(ev, ev2, cfg, ext)
            winner = g.start()
            hist = g.get_history()
            final_state = g.state
        else:
            p1 = mcts_player.Player(ev, cfg, ext["player"])
            p2 = mcts_player.Player(ev2, cfg, ext["player"])
            st = make_state(game, H, W, k)
            cur = p1
            ply = 0
            while not (st.is_end_of_game or st.turns > max_turn) and ply < ply_limit:
                move, _ = cur.think(st, st.turns <= dirichlet_before)
                st.do_move(move)
                p1.ack(move)
                p2.ack(move)
                cur = p2 if cur is p1 else p1
                ply += 1
            winner = st.get_winner()
            final_state = st
    finally:
        mcts_player.Player.think = orig_think
    out = dict(
        name=name, game=game, H=H, W=W, k=k or 0, O=O, transform_types=transform_types,
        max_playout=max_playout, dirichlet_before=dirichlet_before, max_turn=max_turn,
        d_alpha=cfg["d_alpha"], d_epsilon=cfg["d_epsilon"], ply_limit=ply_limit, c_punt=float(mcts.c_punt),
        tids=np.array(streams.tids, np.int32), etas=np.array(streams.etas, np.float64),
        unis=np.array(streams.unis, np.float64),
        n_children=np.array([r["n"] for r in recs], np.int32),
        acts=np.array([r["acts"] for r in recs], np.int32),
        N=np.array([r["N"] for r in recs], np.int32),
        Wv=np.array([r["W"] for r in recs], np.float64 if real64 else np.float32),
        P=np.array([r["P"] for r in recs], np.float64 if real64 else np.float32),
        evaluator=real64 or "fake32",
        rootN=np.array([r["rootN"] for r in recs], np.int32),
        moves=np.array([r["move"] for r in recs], np.int32),
        dirichlet=np.array([r["dirichlet"] for r in recs], np.int8),
        cum_tids=np.array([r["n_tids"] for r in recs], np.int64),
        cum_etas=np.array([r["n_etas"] for r in recs], np.int64),
        cum_unis=np.array([r["n_unis"] for r in recs], np.int64),
        cum_evals=np.array([r["evals"] for r in recs], np.int64),
        boards=np.array([r["board"] for r in recs], np.int8),
        final_board=np.array(final_state.board, np.int8),
        winner=2 if winner is None else int(winner), total_evals=ev.calls + (ev2.calls if ev2 is not ev else 0),
        salts=np.array(salts if salts else (0, 0), np.uint64),
    )
    if hist is not None:
        out["hist_state"] = hist[0].astype(np.int8)
        assert np.array_equal(out["hist_state"], hist[0])
        out["hist_probs"] = hist[1]
        out["hist_result"] = hist[2]
    return out


def save(name, d):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in d.items()})
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024.0))


def main64():
    """Float64 evaluators (DummyEvaluator's uniform priors / the hash widened to double): every search
    intermediate is then a Python float or np.float64 in the reference."""
    save("search64_go9_uniform", search_trace("go9u", "go", 9, 9, None, 8, 120, 30, 400, 31, 8, real64="uniform"))
    save("search64_go5_fake", search_trace("go5f", "go", 5, 5, None, 8, 60, 4, 60, 32, 40, real64="fake"))
    save("search64_reversi_fake", search_trace("rev", "reversi", 8, 8, None, 8, 32, 30, 400, 33, 24, real64="fake"))
    save("search64_ttt_uniform", search_trace("tttu", "mnk", 3, 3, 3, 0, 100, 30, 400, 34, 10 ** 9, real64="uniform"))


def main_shapes():
    """Hand-built Go shapes, each checked here against what the shape is meant to show before it is saved."""
    def seq(moves):
        return [25 if m is None else m[0] * 5 + m[1] for m in moves]
    # snapback: B throws in at (0,1), W takes it with (0,0) and is left with a 4-stone chain on one liberty, so no ko is
    # marked (go.py:547-550) and B retakes at (0,1), capturing four
    snap = [(3, 0), (1, 0), (2, 1), (2, 0), (1, 2), (1, 1), (4, 4), (0, 2), (0, 1), (0, 0), (0, 1), (4, 0), None, None, None]
    d = rules_trace("go", 5, 5, None, 0, 100, moves=seq(snap), feat_every=2)
    assert d["ko"][10] == -1 and d["masks"][10][1] == 1            # after W's capture: no ko, (0,1) playable
    assert tuple(d["pris"][10]) == (1, 0) and tuple(d["pris"][11]) == (1, 4) and d["final_is_end"] == 1
    save("rules_go5_snapback", d)
    # a two-stone capture, a plain suicide point for either side, and a stone played onto a point without liberties
    # that is legal because it captures (and marks a ko)
    sui = [(1, 0), (0, 0), (1, 1), (0, 1), (0, 2), (4, 4), (0, 1), (4, 3), (4, 0), (3, 0), (2, 2), (4, 1), (3, 1), (2, 0),
           (4, 2), (0, 3), (4, 0), (0, 4), None, None, None]
    d = rules_trace("go", 5, 5, None, 0, 100, moves=seq(sui), feat_every=2)
    assert tuple(d["pris"][5]) == (0, 2)                            # B took two stones with (0,2)
    assert d["masks"][7][0] == 0                                    # W may not play the corner (suicide)
    assert d["masks"][12][20] == 0 and d["masks"][12][0] == 1       # B may not play (4,0), may fill its own corner
    assert d["masks"][16][20] == 1 and d["boards"][16][4][0] == 0   # (4,0) has no empty neighbour but captures (4,1)
    assert d["ko"][17] == 21 and d["masks"][17][21] == 0 and tuple(d["pris"][17]) == (1, 3)
    save("rules_go5_suicide", d)
    # a whole Reversi game (6x6, 16 playouts per move) in which the side to move has to pass twice
    d = search_trace("rev6", "reversi", 6, 6, None, 8, 16, 30, 400, 40, 10 ** 9)
    assert int((d["moves"] == 36).sum()) == 2
    save("search_reversi6_pass", d)
    # a drawn tic-tac-toe game through gameplay.Game: get_history labels a draw like a White win (gameplay.py:93)
    d = search_trace("tttd", "mnk", 3, 3, 3, 0, 100, 30, 400, 61, 10 ** 9, use_game_class=True)
    assert d["winner"] == 0 and len(d["moves"]) == 9 and list(d["hist_result"][:2]) == [-1.0, 1.0]
    save("search_ttt_draw", d)
    # a Go game cut off by max_turn (gameplay.py:46-47: the loop stops once turns > max_turn, i.e. after max_turn + 1
    # plies) and scored as it stands
    d = search_trace("go5mt", "go", 5, 5, None, 8, 20, 4, 10, 70, 10 ** 9)
    assert len(d["moves"]) == 11 and int((d["moves"] == 25).sum()) < 2
    save("search_go5_maxturn", d)


def main_match():
    """Games between two different evaluators, as the gating matches play them (train/parallel/evaluator.py:81-85:
    Game(chal, best) when the challenger is Black, Game(best, chal) otherwise)."""
    A, B = 0, 77
    # (gameplay.Game builds GameState() with the module's default size, so the Game-class traces are 8x8 / 3x3)
    save("match_reversi_ab", search_trace("m_rev_ab", "reversi", 8, 8, None, 8, 16, 30, 400, 81, 10 ** 9, use_game_class=True, salts=(A, B)))
    save("match_reversi_ba", search_trace("m_rev_ba", "reversi", 8, 8, None, 8, 16, 30, 400, 81, 10 ** 9, use_game_class=True, salts=(B, A)))
    save("match_go5_ab", search_trace("m_go5_ab", "go", 5, 5, None, 8, 24, 4, 30, 82, 10 ** 9, salts=(A, B)))
    save("match_ttt_ba", search_trace("m_ttt_ba", "mnk", 3, 3, 3, 0, 50, 30, 400, 83, 10 ** 9, use_game_class=True, salts=(B, A)))
    mnk.game_config.update(board_height=6, board_width=5, k=4)


def main_cpunt():
    """The exploration constant is a module global of the reference (search/mcts.py:9) that a user may assign: traces
    recorded with two other values pin that the engine's c_punt means the same thing."""
    old = mcts.c_punt
    try:
        mcts.c_punt = 1.25
        save("search_go9_cpunt125", search_trace("go9c", "go", 9, 9, None, 8, 48, 30, 400, 91, 8))
        mcts.c_punt = 9.5
        save("search_reversi_cpunt95", search_trace("revc", "reversi", 8, 8, None, 8, 40, 30, 400, 92, 12))
    finally:
        mcts.c_punt = old


def main_superko():
    """Positional superko (enforce_superko=True, what the GTP front end plays with: interface/gtp_wrapper.py:88): long random
    games on small boards, where captured stones are replayed often enough for the rule to bite."""
    total = 0
    for H, seeds, plies, pp in ((3, range(100, 140), 300, 0.0), (4, range(200, 240), 600, 0.005), (5, range(300, 330), 800, 0.005)):
        best = sorted(((rules_trace("go", H, H, None, sd, plies, pass_prob=pp, feat_every=10 ** 9, superko=True)) for sd in seeds),
                      key=lambda d: -d["superko_bites"])[:2]
        for i, d in enumerate(best):
            print("rulesko_go%d_%d" % (H, i), "plies", len(d["moves"]), "superko-forbidden points", d["superko_bites"])
            total += d["superko_bites"]
            save("rulesko_go%d_%d" % (H, i), d)
    assert total >= 10
    save("rulesko_go7", rules_trace("go", 7, 7, None, 93, 900, pass_prob=0.01, feat_every=10 ** 9, superko=True))


def main():
    if "--only64" in sys.argv:
        return main64()
    if "--only-superko" in sys.argv:
        return main_superko()
    if "--only-match" in sys.argv:
        return main_match()
    if "--only-shapes" in sys.argv:
        return main_shapes()
    if "--only-cpunt" in sys.argv:
        return main_cpunt()
    # ---- transforms: label tables of ReverseTransformer and forward board maps
    tabs = {}
    for n in (3, 5, 7, 8, 9, 19):
        cfg = cfg_for("go", n, n, 8)
        rt = sc.ReverseTransformer(cfg)
        lab = np.zeros((8, n * n + 1), np.int32)
        fwd = np.zeros((8, n * n), np.int32)
        for tid in range(8):
            ap = [((i // n, i % n), 0.0) for i in range(n * n)] + [(None, 0.0)]
            rt.reverse_transform(ap, tid)
            lab[tid] = [a2i(a, n, n) for a, _ in ap]
            st = go.GameState(size=n)
            st.board = np.arange(n * n).reshape(n, n)
            st.transform(tid)
            fwd[tid] = np.asarray(st.board).ravel()
        tabs["labels_%d" % n] = lab
        tabs["forward_%d" % n] = fwd
    save("transforms", tabs)

    # ---- rules traces
    save("rules_go5_a", rules_trace("go", 5, 5, None, 1, 150, pass_prob=0.03))
    save("rules_go5_b", rules_trace("go", 5, 5, None, 2, 150, pass_prob=0.10))
    save("rules_go7", rules_trace("go", 7, 7, None, 3, 250, pass_prob=0.03))
    save("rules_go9_a", rules_trace("go", 9, 9, None, 4, 400, pass_prob=0.02, feat_every=23))
    save("rules_go9_b", rules_trace("go", 9, 9, None, 5, 400, pass_prob=0.02, feat_every=23))
    save("rules_go19", rules_trace("go", 19, 19, None, 6, 700, pass_prob=0.004, feat_every=97))
    for i, s in enumerate((7, 8, 9)):
        save("rules_reversi_%d" % i, rules_trace("reversi", 8, 8, None, s, 200, feat_every=11))
    save("rules_reversi6", rules_trace("reversi", 6, 6, None, 10, 200, feat_every=11))
    save("rules_mnk333", rules_trace("mnk", 3, 3, 3, 11, 20, feat_every=2))
    save("rules_mnk654", rules_trace("mnk", 6, 5, 4, 12, 40, feat_every=5))
    save("rules_mnk15", rules_trace("mnk", 15, 15, 5, 13, 230, feat_every=31))
    # the reference's own known-answer position (tests/test_go_state_converter.py:27-64)
    simple = [(0, 0), (1, 0), (0, 1), (1, 1), (0, 2), (3, 4), (3, 3), (4, 5), (4, 2), (5, 4), (5, 3), (4, 3), (4, 4)]
    save("rules_go7_simple_board", rules_trace("go", 7, 7, None, 0, 100, moves=[x * 7 + y for x, y in simple] + [49, 49, 49],
                                               feat_every=1))
    # hand-built: double single-stone capture leaves a stale ko; pass/pass end rule
    seq = [
        # W stones on all four sides of (1,2); (0,2) and (1,1) are single stones in atari.
        # B at (1,2) captures (0,2) first (x-1 neighbour) -> ko=(0,2); the later capture of
        # (1,1) gives the stone a second liberty but the ko mark stays (go.py:542-550).
        (0, 1), (0, 2), (0, 3), (1, 1), (2, 1), (1, 3), (1, 0), (2, 2), (1, 2),
        (4, 4), (4, 0),
    ]
    save("rules_go5_double_capture", rules_trace("go", 5, 5, None, 0, 100, moves=[x * 5 + y for x, y in seq] + [25, 25, 25],
                                                 feat_every=3))
    main_shapes()
    passes = [25, 25, 25, 0]
    save("rules_go5_passes", rules_trace("go", 5, 5, None, 0, 100, moves=passes, feat_every=1))
    # SGF main lines shipped with the reference's tests (19x19, real games with captures/ko)
    for fn in ("sgf/G1.sgf", "sgf/G2.sgf", "sgf/G3.sgf"):
        mv = sgf_moves(os.path.join(REF, "tests", "test_data", fn))
        cols = [c for c, _ in mv]
        if all(cols[i] != cols[i + 1] for i in range(len(cols) - 1)) and cols and cols[0] == "B":
            try:
                save("rules_go19_sgf_" + os.path.basename(fn).split(".")[0].replace("-", "_"),
                     rules_trace("go", 19, 19, None, 0, 1000, moves=[m for _, m in mv], feat_every=61))
            except Exception as e:  # an SGF whose line is illegal under these rules is skipped
                print("skip", fn, repr(e))
        else:
            print("skip (colour order)", fn)

    # ---- search / gameplay traces
    save("search_ttt_game", search_trace("ttt", "mnk", 3, 3, 3, 0, 100, 30, 400, 21, 10 ** 9, use_game_class=True))
    save("search_mnk654", search_trace("mnk654", "mnk", 6, 5, 4, 0, 50, 30, 400, 22, 40))
    save("search_gomoku9", search_trace("gomoku9", "mnk", 9, 9, 5, 8, 40, 30, 400, 23, 14))
    save("search_reversi_game", search_trace("reversi", "reversi", 8, 8, None, 8, 32, 30, 400, 24, 10 ** 9, use_game_class=True))
    save("search_go5_game", search_trace("go5", "go", 5, 5, None, 8, 60, 4, 60, 25, 80))
    save("search_go9", search_trace("go9", "go", 9, 9, None, 8, 64, 30, 400, 26, 20))
    save("search_go19", search_trace("go19", "go", 19, 19, None, 8, 30, 30, 400, 27, 6))
    main64()
    main_match()
    main_superko()
    main_cpunt()


if __name__ == "__main__":
    main()
