"""GPU parity on randomly drawn configurations: game, board, simulation count, noise horizon, game length, slot count,
page size and size of the shared edge pool, tree limits with the growing policy, leaves per round (the virtual-loss
schedule, defined by the oracle's oz_tree_search_k), float32 or float64 search statistics, one evaluator or two with a
random assignment of who moves first (the gating matches) - whole games of the HIP engine against
the C oracle's oz_play_game, visit for visit.  The fixed cases of the other test files pin known shapes; this one walks
the space between them (children straddling pages, trees emptied and refilled by re-roots, pools that run dry and grow).

AZ_FUZZ_N sets the number of configurations (default 16, a few seconds); AZ_FUZZ_SEED the first seed."""
import os

import numpy as np
import pytest

from oracle import oracle as oz
from oracle.fake_eval import fake_eval_planes

pytestmark = pytest.mark.gpu

N = int(os.environ.get("AZ_FUZZ_N", "16"))
SEED0 = int(os.environ.get("AZ_FUZZ_SEED", "0"))
SALTS = (5, 99)   # the two hash "networks" of a two-evaluator game


def draw(seed):
    r = np.random.RandomState(1000 + seed)
    game = ["go", "go", "reversi", "mnk"][r.randint(4)]
    if game == "go":
        H = int(r.choice([5, 6, 7, 9, 13])); k = 0
    elif game == "reversi":
        H = int(r.choice([4, 6, 8])); k = 0
    else:
        H = int(r.choice([3, 5, 7, 9])); k = int(r.randint(3, min(H, 5) + 1))
    P = int(r.choice([1, 7, 24, 60, 130]))
    if H >= 13:
        P = min(P, 60)
    tt = 8 if (game != "mnk" or r.rand() < 0.5) else 0
    db = int(r.choice([0, 3, 30]))
    mt = int(r.choice([4, 12, 40, 400])) if H <= 7 else int(r.choice([4, 12, 30]))
    if mt == 400:
        P = min(P, 60)   # whole games to their natural end: keep the number of host round trips in bounds
    G = int(r.randint(1, 7))
    O = H * H + (0 if game == "mnk" else 1)
    o2 = 1
    while o2 < O:
        o2 *= 2
    page = int(r.choice([0, max(64, o2), max(64, 2 * o2)]))
    mode = r.randint(3)
    kw = dict(tree_page_edges=page)
    if mode == 0:      # automatic limits and pool
        pass
    elif mode == 1:    # a scarce pool, grown on demand
        kw.update(tree_pool_edges=int(2 * G * max(64, o2) * (1 + r.randint(3))), tree_full_policy=2)
    else:              # undersized per-tree limits, grown on demand
        kw.update(tree_edge_cap=int(O * (1 + r.randint(4))), tree_block_cap=int(2 + r.randint(6)), tree_full_policy=2)
    K = int(r.choice([1, 1, 1, 2, 3, 5, 8]))
    vloss = float(r.choice([0.25, 1.0]))
    f64 = bool(r.rand() < 0.2)
    match = bool(K == 1 and not f64 and r.rand() < 0.3)   # two evaluators: Game(ev_a, ev_b) or the two swapped, per slot
    first = r.randint(0, 2, size=G).astype(np.int32)
    return dict(game=game, H=H, k=k, P=P, tt=tt, db=db, mt=mt, G=G, O=O, kw=kw, K=K, vloss=vloss, f64=f64, match=match, first=first)


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + N))
def test_random_configuration_against_the_oracle(seed):
    from alphazero_amd.engine import Engine
    c = draw(seed)
    game, H, k, P, tt, db, mt, G, O = (c[x] for x in ("game", "H", "k", "P", "tt", "db", "mt", "G", "O"))
    K, vloss, f64 = c["K"], c["vloss"], c["f64"]
    rng = np.random.RandomState(seed)
    eng = Engine(game, H, H, k, n_slots=G, max_playout=P, transform_types=tt, dirichlet_before=db, max_turn=mt, rng_replay=True,
                 record_trajectories=True, leaves_per_round=K, virtual_loss=vloss, search_f64=f64, **c["kw"])
    streams = []
    for s in range(G):
        st = (rng.randint(0, 8, size=(mt + 2) * (P + 1)).astype(np.int32), rng.gamma(0.03, 1.0, size=(mt + 2) * O), rng.rand(mt + 2))
        streams.append(st)
        eng.rng_inject(s, *st)
    eng.selfplay_start()
    if c["match"]:
        eng.match_assign(np.arange(G), c["first"])
    while eng.selfplay_select() > 0:
        planes, _ = eng.leaf_batch()
        pri, v = fake_eval_planes(planes, O)
        if c["match"]:   # the engine says which network owns each leaf; rows that need no answer get one that would show if read
            nets = eng.match_leaf_nets()
            pri[:], v[:] = np.nan, np.nan
            for net in (0, 1):
                rows = np.flatnonzero(nets == net)
                if rows.size:
                    pri[rows], v[rows] = fake_eval_planes(planes[rows], O, SALTS[net])
        if f64:   # the same hash evaluator handing out float64: every N / W / P intermediate is then a double
            pri, v = pri.astype(np.float64), v.astype(np.float64)
        eng.selfplay_finish(pri, v)
    ctr = eng.counters()
    page = ctr["tree_page_edges"]
    held = sum(max(1, -(-eng.tree_size(s, side)[0] // page)) for s in range(G) for side in (0, 1))
    assert ctr["tree_pool_pages_in_use"] == held, (c, ctr)
    assert ctr["expansions_skipped"] == 0
    for s in range(G):
        salts = (SALTS[c["first"][s]], SALTS[1 - c["first"][s]]) if c["match"] else None
        ref = oz.play_game(game, H, H, k, tt, P, db, mt, oz.fn_ptr("oz_fake_eval", real64=f64), oz.Rng(*streams[s]), real64=f64,
                           K=K, vloss=vloss, salts=salts)
        tr = eng.trajectory_pop(s)
        assert np.array_equal(tr["moves"], ref["moves"]), (c, s)
        assert np.array_equal(tr["visits"], ref["visits"]), (c, s)
        assert tr["winner"] == ref["winner"], (c, s)
    eng.close()
