"""Two-network (gating) matches on the device: the batched path that routes every leaf to the network
owning the searching tree must reproduce, bit for bit, the same engine driven leaf by leaf with priors
computed separately by each network (the reference's Game(chal, best) / Game(best, chal),
train/parallel/evaluator.py:81-85)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _blobs(H, W, O, NB, C):
    from alphazero_amd.network.model import fold_to_blob, random_net
    return (fold_to_blob(random_net(H, W, O, NB, C, seed=1, perturb_bn=True)),
            fold_to_blob(random_net(H, W, O, NB, C, seed=2, perturb_bn=True)))


@pytest.mark.parametrize("H,C,dtype,G,P,max_turn", [(5, 128, "fp32", 6, 10, 12), (9, 256, "bf16", 6, 10, 12),
                                                   (5, 128, "bf16", 1500, 3, 5),    # > 1024 slots: several partition passes
                                                   (9, 256, "bf16", 320, 3, 5)])    # large enough for the queued rounds whose towers
                                                                                    # read their batch sizes from device memory
def test_two_network_routing_is_exact(H, C, dtype, G, P, max_turn):
    from alphazero_amd.engine import Engine
    NB, O = 1, H * H + 1
    kw = dict(n_slots=G, max_playout=P, transform_types=8, dirichlet_before=4, max_turn=max_turn, record_trajectories=True,
              seed=11, num_blocks=NB, channels=C, tower_dtype=dtype)
    blob_a, blob_b = _blobs(H, H, O, NB, C)
    first = np.resize(np.array([0, 1, 0, 1, 1, 0, 1], np.int32), G)
    rounds = (max_turn + 3) * (P + 1)

    fast = Engine("go", H, H, **kw)
    fast.tower_load(blob_a, net=0)
    fast.tower_load(blob_b, net=1)
    fast.selfplay_start()
    fast.match_assign(np.arange(G), first)
    fast.selfplay_run(rounds)

    slow = Engine("go", H, H, **kw)
    ta = Engine("go", H, H, n_slots=G, num_blocks=NB, channels=C, tower_dtype=dtype)
    tb = Engine("go", H, H, n_slots=G, num_blocks=NB, channels=C, tower_dtype=dtype)
    ta.tower_load(blob_a)
    tb.tower_load(blob_b)
    slow.selfplay_start()
    for _ in range(rounds):
        slow.selfplay_select()
        planes, valid = slow.leaf_batch()
        plies, _ = slow.selfplay_status()
        pa, va = ta.tower_forward(planes)
        pb, vb = tb.tower_forward(planes)
        net = first ^ (plies & 1)              # side 0 (Black, player_1) searches on even plies
        pri = np.where(net[:, None] == 0, pa, pb).astype(np.float32)
        val = np.where(net == 0, va, vb).astype(np.float32)
        slow.selfplay_finish(pri, val)

    pf, df = fast.selfplay_status()
    ps, ds = slow.selfplay_status()
    assert np.array_equal(pf, ps) and np.array_equal(df, ds) and np.all(df == 1)
    nets_differ = False
    for s in range(G):
        a, b = fast.trajectory_pop(s), slow.trajectory_pop(s)
        assert np.array_equal(a["moves"], b["moves"]) and np.array_equal(a["visits"], b["visits"]) and a["winner"] == b["winner"]
    # the two networks really are different (otherwise the routing would not be tested)
    x = (np.random.RandomState(0).rand(2, 17, H, H) < 0.3).astype(np.int8)
    nets_differ = np.abs(ta.tower_forward(x)[0] - tb.tower_forward(x)[0]).max() > 1e-4
    assert nets_differ
    for e in (fast, slow, ta, tb):
        e.close()


def test_match_with_towers_in_board_groups_is_bit_identical(monkeypatch):
    """Two networks, queued rounds (batch sizes in device memory), each tower taking its share through the layers in groups
    of 420 boards - the same convolution kernel as the whole batch gets, so whole games must come out bit for bit the same
    as with whole-batch launches.  (Both towers run on the engine's stream, one after the other: the default.)"""
    from alphazero_amd.engine import Engine
    H, G, P, max_turn, NB = 9, 1680, 3, 5, 1
    O = H * H + 1
    blob_a, blob_b = _blobs(H, H, O, NB, 256)
    first = np.resize(np.array([0, 1, 1, 0, 1], np.int32), G)
    out = []
    for group in ("420", "-1"):
        monkeypatch.setenv("AZ_TOWER_GROUP_BOARDS", group)
        eng = Engine("go", H, H, n_slots=G, max_playout=P, transform_types=8, dirichlet_before=4, max_turn=max_turn,
                     record_trajectories=True, seed=21, num_blocks=NB, channels=256, tower_dtype="bf16")
        eng.tower_load(blob_a, net=0)
        eng.tower_load(blob_b, net=1)
        eng.selfplay_start()
        eng.match_assign(np.arange(G), first)
        eng.selfplay_run((max_turn + 3) * (P + 1))
        plies, done = eng.selfplay_status()
        assert np.all(done == 1)
        out.append([eng.trajectory_pop(s) for s in range(G)])
        eng.close()
    for a, b in zip(*out):
        assert np.array_equal(a["moves"], b["moves"]) and np.array_equal(a["visits"], b["visits"]) and a["winner"] == b["winner"]


def test_batched_match_counts_like_the_reference():
    import yaml
    import os
    from alphazero_amd.evaluator.nn_eval import NNEvaluator
    from alphazero_amd.game.gameplay import BatchedMatch, gating_colors, gating_verdict
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alphazero_amd", "config")
    with open(os.path.join(root, "mnk.yaml")) as f:
        cfg = yaml.safe_load(f)
    chal = NNEvaluator(cfg, num_blocks=1, channels=128, dtype="fp32", max_batch_size=2, seed=5)
    best = NNEvaluator(cfg, num_blocks=1, channels=128, dtype="fp32", max_batch_size=2, seed=6)
    ext = {"num_games": 7, "num_worker": 2,
           "gameplay": {"dirichlet_before": 2, "log_iter": 30, "max_turn": 400, "player": {"max_playout": 8}}}
    m = BatchedMatch(chal, best, cfg, ext, n_parallel=3, seed=3)
    out = m.run(rounds_per_poll=16)
    assert out["colors_of_new"] == gating_colors(7)
    assert len(out["winners"]) == 7 and all(w in (-1, 0, 1) for w in out["winners"])
    assert (out["win_counter"], out["num_not_tie"], out["accepted"]) == gating_verdict(out["winners"], out["colors_of_new"])


@pytest.mark.parametrize("game,H,k,G,P,rounds,K", [("reversi", 8, 0, 400, 6, 70 * 7, 1), ("mnk", 8, 4, 400, 5, 66 * 6, 1),
                                                    ("reversi", 8, 0, 400, 6, 70 * 5, 2)])   # two leaf rows per slot
def test_leaf_compaction_changes_nothing_but_the_work(game, H, k, G, P, rounds, K):
    """Whole games with many finished-game leaves near the end: the tower run on the compacted rows only (device-side
    row list and count) gives the same trajectories, bit for bit, as the tower run on every row, and evaluates fewer
    boards."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.network.model import fold_to_blob, random_net
    O = H * H + (0 if game == "mnk" else 1)
    blob = fold_to_blob(random_net(H, H, O, 1, 256, seed=9, perturb_bn=True))
    out = []
    for compact in (True, False):
        eng = Engine(game, H, H, k, n_slots=G, max_playout=P, transform_types=8, dirichlet_before=4, max_turn=400,
                     record_trajectories=True, seed=5, num_blocks=1, channels=256, tower_dtype="bf16", leaves_per_round=K)
        eng.tower_load(blob)
        eng.tower_leaf_compaction(compact)
        eng.selfplay_start()
        eng.selfplay_run(rounds)
        plies, done = eng.selfplay_status()
        assert np.all(done == 1)
        c = eng.counters()
        out.append(([eng.trajectory_pop(s) for s in range(G)], c["tower_evals"], c["evals"], plies))
        eng.close()
    (ta, ea, na, pa), (tb, eb, nb, pb) = out
    assert np.array_equal(pa, pb) and na == nb
    for a, b in zip(ta, tb):
        assert np.array_equal(a["moves"], b["moves"]) and np.array_equal(a["visits"], b["visits"]) and a["winner"] == b["winner"]
    assert eb == rounds * G * K        # every row of every round
    assert ea < na <= eb               # only live, non-terminal leaves: fewer than the playouts counted


def _fixture_engine(g, G):
    from alphazero_amd.engine import Engine
    return Engine(str(g["game"]), int(g["H"]), int(g["W"]), int(g["k"]), n_slots=G, max_playout=int(g["max_playout"]),
                  transform_types=int(g["transform_types"]), dirichlet_before=int(g["dirichlet_before"]),
                  max_turn=int(g["max_turn"]), d_alpha=float(g["d_alpha"]), d_epsilon=float(g["d_epsilon"]), rng_replay=True,
                  record_trajectories=True)


def _check_against_fixture(tr, g, O):
    assert np.array_equal(tr["moves"], g["moves"]) and tr["winner"] == g["winner"]
    for ply in range(len(g["moves"])):
        nc = int(g["n_children"][ply])
        exp = np.zeros(O, np.int32)
        exp[g["acts"][ply][:nc]] = g["N"][ply][:nc]
        assert np.array_equal(tr["visits"][ply], exp), ply


@pytest.mark.parametrize("names", [("match_reversi_ab", "match_reversi_ba"), ("match_go5_ab",), ("match_ttt_ba",)])
def test_match_assignment_against_reference_games(names):
    """az_match_assign and the device-side leaf split held to games the reference played with two different evaluators
    (Game(ev_a, ev_b, ...), fixtures match_*): the engine says which network owns each round's leaf
    (az_match_leaf_nets, the lists its tower path uses), the test feeds that evaluator's priors, and moves, visit counts
    and winners must be the reference's.  Slot i plays fixture i; a fixture recorded with the evaluators swapped runs
    with net_of_first = 1."""
    from oracle.fake_eval import fake_eval_planes
    from tests.helpers import load_golden
    gs = [load_golden(n) for n in names]
    salts = [int(x) for x in gs[0]["salts"]]          # network 0 <-> salts[0], network 1 <-> salts[1]
    G = len(gs)
    eng = _fixture_engine(gs[0], G)
    first = []
    for s, g in enumerate(gs):
        eng.rng_inject(s, g["tids"], g["etas"], g["unis"])
        first.append(salts.index(int(g["salts"][0])))
        assert salts[1 - first[-1]] == int(g["salts"][1])
    eng.selfplay_start()
    eng.match_assign(np.arange(G), first)
    O = eng.O
    while eng.selfplay_select() > 0:
        planes, valid = eng.leaf_batch()
        nets = eng.match_leaf_nets()
        assert np.all((nets >= 0) <= (valid[:G] == 1))
        pri = np.zeros((G, O), np.float32); val = np.zeros(G, np.float32)
        for s in range(G):
            if nets[s] >= 0:
                pri[s], val[s] = fake_eval_planes(planes[s], O, salts[nets[s]])
            else:                                      # a finished game below the root: the answer is never read
                pri[s], val[s] = np.nan, np.nan
        eng.selfplay_finish(pri, val)
    plies, done = eng.selfplay_status()
    assert np.all(done == 1)
    for s, g in enumerate(gs):
        _check_against_fixture(eng.trajectory_pop(s), g, O)
    eng.close()


def test_match_assignment_against_oracle_many_slots():
    """The same through 40 slots with mixed assignments and their own draws, against oz_play_game_ex."""
    from alphazero_amd.engine import Engine
    from oracle import oracle as oz
    from oracle.fake_eval import fake_eval_planes
    H, P, mt, G = 6, 12, 40, 40
    salts = (5, 99)
    rng = np.random.RandomState(3)
    eng = Engine("reversi", H, H, n_slots=G, max_playout=P, transform_types=8, dirichlet_before=6, max_turn=mt, rng_replay=True,
                 record_trajectories=True)
    O = eng.O
    streams = []
    for s in range(G):
        st = (rng.randint(0, 8, size=(mt + 2) * (P + 1)).astype(np.int32), rng.gamma(0.03, 1.0, size=(mt + 2) * O), rng.rand(mt + 2))
        streams.append(st)
        eng.rng_inject(s, *st)
    first = rng.randint(0, 2, size=G).astype(np.int32)
    eng.selfplay_start()
    eng.match_assign(np.arange(G), first)
    while eng.selfplay_select() > 0:
        planes, valid = eng.leaf_batch()
        nets = eng.match_leaf_nets()
        pri = np.full((G, O), np.nan, np.float32); val = np.full(G, np.nan, np.float32)
        for net in (0, 1):
            rows = np.flatnonzero(nets == net)
            if rows.size:
                pri[rows], val[rows] = fake_eval_planes(planes[rows], O, salts[net])
        eng.selfplay_finish(pri, val)
    for s in range(G):
        a, b = salts[first[s]], salts[1 - first[s]]
        ref = oz.play_game("reversi", H, H, 0, 8, P, 6, mt, oz.fn_ptr("oz_fake_eval"), oz.Rng(*streams[s]), salts=(a, b))
        tr = eng.trajectory_pop(s)
        assert np.array_equal(tr["moves"], ref["moves"]) and np.array_equal(tr["visits"], ref["visits"]) and tr["winner"] == ref["winner"], s
    eng.close()
