"""az_trajectory_pop_many_static (the entry point that allocates nothing: its device staging is a fixed array inside the
library) against az_trajectory_pop_many, through the C ABI, bit for bit, on two calls in a row - the second with a
smaller input - plus its size limits."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _finished_engine(game, H, O_extra, G, max_turn, seed):
    from alphazero_amd.engine import Engine
    from alphazero_amd.network.model import fold_to_blob, random_net
    O = H * H + O_extra
    P = 12
    eng = Engine(game, H, H, 3, n_slots=G, max_playout=P, transform_types=8 if game == "go" else 0, max_turn=max_turn,
                 num_blocks=1, channels=128, tower_dtype="bf16", record_trajectories=True, seed=seed)
    eng.tower_load(fold_to_blob(random_net(H, H, O, 1, 128, seed=seed)))
    eng.selfplay_start()
    for _ in range(40):
        eng.selfplay_run((P + 1) * 4)
        plies, done = eng.selfplay_status()
        if done.all():
            break
    assert done.all()
    return eng, plies


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert np.array_equal(x["moves"], y["moves"]) and np.array_equal(x["visits"], y["visits"]) and x["winner"] == y["winner"]
        assert x["visits"].sum() > 0


@pytest.mark.parametrize("game,H,extra,G,max_turn", [("mnk", 3, 0, 96, 400), ("go", 9, 1, 48, 11)])
def test_static_pop_equals_pop_many_twice(game, H, extra, G, max_turn):
    eng, plies = _finished_engine(game, H, extra, G, max_turn, seed=5)
    first = np.arange(0, G, 2)                 # every second slot: a ragged packed copy
    want = eng.trajectory_pop_many(first, plies[first], restart=False)
    got = eng.trajectory_pop_many(first, plies[first], restart=False, static=True)
    _same(got, want)
    second = np.array([G - 1, 3, 10, 7, 1])    # smaller, other slots, not sorted: the array still holds the first call's rows
    want2 = eng.trajectory_pop_many(second, plies[second], restart=False)
    import torch
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info(0)[0]
    got2 = eng.trajectory_pop_many(second, plies[second], restart=False, static=True)
    # "allocates nothing": the GPU's free memory is what it was before the call (the allocating call above grew its staging
    # buffer on its first use; this one has none to grow)
    assert torch.cuda.mem_get_info(0)[0] == free_before
    _same(got2, want2)
    # restart through the static call re-arms the slots like the allocating call does
    eng.trajectory_pop_many(second, plies[second], restart=True, static=True)
    plies3, done3 = eng.selfplay_status()
    rest = np.setdiff1d(np.arange(G), second)
    assert not done3[second].any() and (plies3[second] == 0).all() and done3[rest].all()
    eng.close()


def test_static_pop_rejects_what_does_not_fit():
    from alphazero_amd.engine import Engine, EngineError
    eng = Engine("go", 19, 19, n_slots=300, record_trajectories=True)
    L = eng._L
    slots = np.arange(300, dtype=np.int32)
    starts = np.zeros(300, np.int32)
    moves = np.zeros(8, np.int16); visits = np.zeros((8, eng.O), np.int32); win = np.zeros(300, np.int32)

    def call(n, total):
        return L.az_trajectory_pop_many_static(eng._h, slots.ctypes.data, n, starts.ctypes.data, total, moves.ctypes.data,
                                               visits.ctypes.data, win.ctypes.data, 0)
    assert call(257, 0) == -1 and b"256" in L.az_last_error(eng._h)
    rows_max = ((8 << 20) - 16384) // (2 + 4 * eng.O)
    assert call(2, rows_max + 1) == -1 and b"exceed" in L.az_last_error(eng._h)
    assert call(2, 0) == -7          # within the limits, but no game here is finished
    eng.close()
