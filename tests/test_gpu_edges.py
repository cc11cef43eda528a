"""Edge cases of the C ABI on the device: empty and ragged batches, out-of-range arguments, the smallest and largest
boards, a batch that ends inside a conv tile.  Error behaviour follows the reference where it has one (IllegalMove on
an occupied point, env/go.py:511-516; a non-square board cannot be transformed, env/mnk.py:199) and is an error code
with a message everywhere else."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_empty_batches_are_no_ops():
    from alphazero_amd.engine import Engine
    eng = Engine("go", 5, 5, n_slots=3, transform_types=8)
    before = eng.board(0).copy()
    none = np.zeros(0, np.int32)
    eng.reset(none)
    assert eng.legal_mask(none).shape == (0, 26)
    st, ended = eng.do_move(none, none)
    assert st.size == 0 and ended.size == 0
    assert eng.winner(none).size == 0
    assert eng.features(none).shape == (0, 17, 5, 5)
    eng.search_begin(none, none, none)
    eng.advance(none, none)
    assert np.array_equal(eng.board(0), before)
    eng.close()


def test_arguments_out_of_range_are_refused_and_leave_the_state_alone():
    from alphazero_amd.engine import Engine, EngineError
    eng = Engine("go", 5, 5, n_slots=3, transform_types=8)
    eng.do_move([0], [7])
    snap = eng.board(0).copy()
    for bad in ([3], [-1], [0, 5]):
        with pytest.raises(EngineError):
            eng.legal_mask(bad)
        with pytest.raises(EngineError):
            eng.do_move(bad, [1] * len(bad))
        with pytest.raises(EngineError):
            eng.features(bad)
        with pytest.raises(EngineError):
            eng.reset(bad)
    with pytest.raises(EngineError):
        eng.do_move([1, 1], [2, 3])            # one slot twice in a batch: order would be undefined
    with pytest.raises(EngineError):
        eng.features([0], [8])                 # transform ids are 0..7
    with pytest.raises(EngineError):
        eng.transform(0, -1)
    with pytest.raises(EngineError):
        eng.search_begin([0], [2], [0])        # side is 0 or 1
    # an action index past PASS, a negative one and an occupied point: per-slot status, nothing moves
    st, _ = eng.do_move([0, 1, 2], [7, 26, -2])
    assert list(st) == [-3, -3, -3]
    assert np.array_equal(eng.board(0), snap) and not eng.board(1).any() and not eng.board(2).any()
    assert str(EngineError(-2, "x")) and eng.legal_mask([0])[0, 7] == 0
    eng.close()


def test_rectangular_boards_only_take_the_identity_transform():
    from alphazero_amd.engine import Engine, EngineError
    eng = Engine("mnk", 6, 5, 4, n_slots=2)
    eng.do_move([0], [7])
    assert eng.features([0], [0]).shape == (1, 17, 6, 5)
    with pytest.raises(EngineError):
        eng.features([0], [1])
    with pytest.raises(EngineError):
        eng.transform(0, 3)
    eng.transform(0, 0)
    with pytest.raises(EngineError):
        Engine("mnk", 6, 5, 4, n_slots=2, transform_types=8)
    eng.close()


@pytest.mark.parametrize("game,H,k", [("mnk", 3, 3), ("go", 19, 0), ("mnk", 19, 5)])
def test_smallest_and_largest_boards_play_to_the_end(game, H, k):
    """Random legal games on the extreme board sizes: every move the mask allows is accepted, every other refused."""
    from alphazero_amd.engine import Engine
    G = 4
    eng = Engine(game, H, H, k, n_slots=G, transform_types=8)
    rng = np.random.RandomState(H)
    live = list(range(G))
    for ply in range(2 * H * H + 40):
        if not live:
            break
        masks = eng.legal_mask(live)
        moves = []
        for m in masks:
            legal = np.flatnonzero(m)
            assert legal.size > 0
            # Go: board moves for H*H plies, then passes only, so that every game ends in bounded time
            if game == "go":
                legal = legal[legal != H * H] if (ply < H * H and legal.size > 1) else np.array([H * H])
            moves.append(int(rng.choice(legal)))
        illegal = [int(np.flatnonzero(m == 0)[0]) if (m == 0).any() else -1 for m in masks]
        st, _ = eng.do_move(live, illegal)
        assert all(s == -3 for s in st)
        st, ended = eng.do_move(live, moves)
        assert not st.any()
        live = [s for s, e in zip(live, ended) if not e]
    assert not live
    assert all(w in (-1, 0, 1) for w in eng.winner(list(range(G))))
    eng.close()


@pytest.mark.parametrize("B", [1, 15, 16, 17, 46, 47])
def test_tower_batches_that_end_inside_a_tile_agree_with_single_boards(B):
    """19x19: 361 points per board against 256-point conv tiles, so most batch sizes end inside a tile and several
    kernels (two small-batch shapes, the slab kernel) serve the range; a board's outputs must not depend on its
    neighbours in the batch beyond bf16 accumulation order (the kernels differ in it: 2e-3 on priors)."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.network.model import fold_to_blob, random_net
    H, NB, C = 19, 1, 256
    blob = fold_to_blob(random_net(H, H, 362, NB, C, seed=4, perturb_bn=True))
    big = Engine("go", H, H, n_slots=64, num_blocks=NB, channels=C, tower_dtype="bf16")
    big.tower_load(blob)
    x = (np.random.RandomState(1).rand(64, 17, H, H) < 0.3).astype(np.int8)
    p_all, v_all = big.tower_forward(x)
    p, v = big.tower_forward(x[:B])
    assert np.abs(p - p_all[:B]).max() <= 2e-3 and np.abs(v - v_all[:B]).max() <= 2e-2
    # same kernel, same batch: bit-identical from call to call
    p2, v2 = big.tower_forward(x[:B])
    assert np.array_equal(p, p2) and np.array_equal(v, v2)
    big.close()


def test_float32_and_float64_searches_refuse_each_others_inputs():
    from alphazero_amd.engine import Engine, EngineError
    e32 = Engine("mnk", 3, 3, 3, n_slots=1, max_playout=4)
    e64 = Engine("mnk", 3, 3, 3, n_slots=1, max_playout=4, search_f64=True)
    for eng, other in ((e32, np.float64), (e64, np.float32)):
        eng.search_begin([0], [0], [0])
        eng.step_select()
        lib_fn = eng._L.az_search_step_finish if other is np.float32 else eng._L.az_search_step_finish_f64
        p = np.full((1, 9), 1 / 9, other)
        v = np.zeros(1, other)
        assert lib_fn(eng._h, p.ctypes.data, v.ctypes.data) < 0
        eng.step_finish(np.full((1, 9), 1 / 9), np.zeros(1))   # the matching entry point still works
    with pytest.raises(EngineError):
        e64.step_finish()                                        # no tower outputs for a float64 search
    e32.close()
    e64.close()


def test_graph_replay_equals_plain_launches(monkeypatch):
    """Forwards of small batches are captured once and replayed as a HIP graph: outputs must be bit-identical to plain
    launches, on the capturing call, on replays, and when batch sizes alternate."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.network.model import fold_to_blob, random_net
    H, NB, C = 19, 2, 256
    blob = fold_to_blob(random_net(H, H, 362, NB, C, seed=8, perturb_bn=True))
    x = (np.random.RandomState(2).rand(40, 17, H, H) < 0.3).astype(np.int8)
    monkeypatch.delenv("AZ_NO_GRAPHS", raising=False)
    graphs = Engine("go", H, H, n_slots=40, num_blocks=NB, channels=C, tower_dtype="bf16")
    monkeypatch.setenv("AZ_NO_GRAPHS", "1")          # read when the tower is created
    plain = Engine("go", H, H, n_slots=40, num_blocks=NB, channels=C, tower_dtype="bf16")
    for e in (graphs, plain):
        e.tower_load(blob)
    for B in (1, 5, 1, 32, 5, 40, 32, 1):
        pg, vg = graphs.tower_forward(x[:B])
        pp, vp = plain.tower_forward(x[:B])
        assert np.array_equal(pg, pp) and np.array_equal(vg, vp), B
    graphs.close()
    plain.close()
