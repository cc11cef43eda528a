"""Full-size run (Go 19x19, 800 simulations per move, root noise, 8 transforms) checked through properties that do
not need the CPU oracle to finish a search of this size: visit accounting of PUCT, legality of every chosen move
under the rule kernels, the reference's result labelling, and bit-reproducibility from the seed."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

H, P, G, PLIES = 19, 800, 48, 3


def _run(seed):
    from alphazero_amd.engine import Engine
    from alphazero_amd.network.model import fold_to_blob, random_net
    O = H * H + 1
    eng = Engine("go", H, H, n_slots=G, max_playout=P, transform_types=8, dirichlet_before=30, max_turn=PLIES - 1,
                 record_trajectories=True, seed=seed, num_blocks=2, channels=128, tower_dtype="bf16")
    eng.tower_load(fold_to_blob(random_net(H, H, O, 2, 128, seed=1)))
    eng.selfplay_start()
    eng.selfplay_run(PLIES * (P + 1) + 4)
    plies, done = eng.selfplay_status()
    trs = [eng.trajectory_pop(s) for s in range(G)]
    c = eng.counters()
    eng.close()
    return plies, done, trs, c


def test_go19_800_sims_properties():
    from alphazero_amd.engine import Engine
    plies, done, trs, c = _run(5)
    assert np.all(done == 1) and np.all(plies == PLIES)          # max_turn = PLIES-1 stops the game after PLIES moves
    # every search costs 800 evaluations, +1 when its root was still a leaf (plies 0 and 1 of each game)
    assert c["evals"] >= G * (PLIES * P + 2) and c["moves"] == G * PLIES
    rules = Engine("go", H, H, n_slots=G)
    rules.reset(list(range(G)))
    for t in range(PLIES):
        for s in range(G):
            v = trs[s]["visits"][t]
            total = int(v.sum())
            if t < 2:
                assert total == P                                 # fresh root: exactly one visit per playout
            else:
                assert total >= P                                 # re-rooted subtree keeps its earlier visits
            assert v[int(trs[s]["moves"][t])] > 0                 # the sampled move was visited
        status, ended = rules.do_move(list(range(G)), [int(trs[s]["moves"][t]) for s in range(G)])
        assert np.all(status == 0)                                # ... and is legal under the rule kernel
        mask = rules.legal_mask(list(range(G)))
        assert mask.shape == (G, H * H + 1)
    winners = rules.winner(list(range(G)))
    assert [int(w) for w in winners] == [tr["winner"] for tr in trs]
    # visits only on legal points of the position they were searched from
    rules.reset(list(range(G)))
    m0 = rules.legal_mask(list(range(G)))
    for s in range(G):
        assert np.all(trs[s]["visits"][0][m0[s] == 0] == 0)
    rules.close()


def test_go19_run_is_reproducible_from_the_seed():
    a = _run(9)
    b = _run(9)
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(x["moves"], y["moves"]) and np.array_equal(x["visits"], y["visits"]) and x["winner"] == y["winner"]
    other = _run(10)
    assert any(not np.array_equal(x["moves"], y["moves"]) for x, y in zip(a[2], other[2]))


def test_edge_pool_accounting_at_the_bench_size():
    """3993 games of Go 19x19 at 800 simulations with the bench's pool (a third of one arena per tree): after a few hundred
    rounds, and again after every game has moved and both of its trees were re-rooted, every page in use belongs to exactly
    one tree and nothing was skipped."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.network.model import fold_to_blob, random_net
    Gb, O = 3993, H * H + 1
    eng = Engine("go", H, H, n_slots=Gb, max_playout=P, transform_types=8, dirichlet_before=30, max_turn=400, seed=3,
                 num_blocks=1, channels=128, tower_dtype="bf16", tree_pool_edges=2 * Gb * (P + 1) * O)
    eng.tower_load(fold_to_blob(random_net(H, H, O, 1, 128, seed=1)))
    eng.selfplay_start()

    def held():
        page = eng.counters()["tree_page_edges"]
        return sum(max(1, -(-eng.tree_size(s, side)[0] // page)) for s in range(Gb) for side in (0, 1))

    eng.selfplay_run(300)
    c = eng.counters()
    assert c["tree_pool_pages_in_use"] == held() and c["expansions_skipped"] == 0
    assert c["tree_pool_pages_in_use"] > 2 * Gb                      # the searching trees have taken pages beyond their own
    eng.selfplay_run(P + 1 - 300 + 2)                                 # every game finishes its search, moves, re-roots
    c = eng.counters()
    assert c["moves"] == Gb and c["expansions_skipped"] == 0
    assert c["tree_pool_pages_in_use"] == held()
    assert c["tree_pool_pages_in_use"] < c["tree_pool_pages_peak"] <= c["tree_pool_pages"]   # the re-roots gave pages back
    eng.close()
