"""Replay ring + trajectory files against the reference's DataPool (train/parallel/datapool.py), pinned by
tests/golden/datapool_ring.npz (written by oracle/gen_golden_datapool.py from the reference itself)."""
import os

import numpy as np

from alphazero_amd.data.replay import ReplayPool, load_games, save_game

from tests.helpers import load_golden


def test_ring_overwrite_and_sampling_match_reference():
    g = load_golden("datapool_ring")
    pool = ReplayPool({'pool_size': int(g["capacity"]), 'start_data_size': int(g["start_data_size"])})
    for k in range(len(g["sizes"])):
        pool.merge_data(tuple(g["put%d_arr%d" % (k, j)] for j in range(3)))
        for j in range(3):
            assert np.array_equal(pool.data_pool[j], g["after%d_pool%d" % (k, j)])
        assert pool.pool_end_index == int(g["after%d_end" % k])
        assert int(pool.is_full) == int(g["after%d_full" % k])
        assert int(pool.training_started) == int(g["after%d_started" % k])
    np.random.seed(123)
    got = pool.get(6)
    for j in range(3):
        assert np.array_equal(got[j], g["get_arr%d" % j])


def test_game_files_round_trip(tmp_path):
    g = load_golden("datapool_ring")
    games = [tuple(g["put%d_arr%d" % (k, j)] for j in range(3)) for k in range(3)]
    pool = ReplayPool({'pool_size': 64, 'start_data_size': 4, 'store_path': str(tmp_path)})
    for data in games:
        pool.put(data)
    assert sorted(os.listdir(tmp_path)) == ["0.npz", "1.npz", "2.npz"]
    with np.load(os.path.join(tmp_path, "0.npz")) as z:
        assert sorted(z.files) == list(g["file_keys"])
        for j in range(3):
            assert np.array_equal(z["arr_%d" % j], games[0][j]) and z["arr_%d" % j].dtype == np.float64
    # load_prev: a new pool rebuilt from the directory holds the same rows (order = directory order)
    again = ReplayPool({'pool_size': 64, 'start_data_size': 4, 'store_path': str(tmp_path), 'load_prev': True})
    assert again.count == 3 and again.pool_end_index == pool.pool_end_index
    rows = lambda p: sorted(map(bytes, p.data_pool[1][:p.pool_end_index]))  # noqa: E731
    assert rows(again) == rows(pool)
    assert len(list(load_games(str(tmp_path)))) == 3
    save_game(str(tmp_path), 7, games[1])
    assert os.path.exists(os.path.join(tmp_path, "7.npz"))
