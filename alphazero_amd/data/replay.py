"""Trajectory files and the replay ring in the reference's formats, so that games produced by the HIP
engine can feed the reference's optimizer (train/parallel/datapool.py) unchanged.

* one finished game = one `<count>.npz` written with `np.savez_compressed(path, state_np, probs_np, result_np)`
  -> keys arr_0 (T,17,H,W), arr_1 (T,O), arr_2 (T,)  (datapool.py:65-66; read back at :54-58)
* ReplayPool = the pre-allocated ring of DataPool.merge_data (datapool.py:79-112) and the uniform
  with-replacement sampling of the 'get' request (datapool.py:70-77), in-process (no server, no pipes).
"""
import os

import numpy as np


def save_game(store_path, count, data):
    """data = Game.get_history() tuple; file name and keys as datapool.py:66."""
    os.makedirs(store_path, exist_ok=True)
    np.savez_compressed(os.path.join(store_path, str(count)), *data)


def load_games(store_path):
    """Yields (state_np, probs_np, result_np) of every .npz in the directory (datapool.py:54-57)."""
    for file in os.listdir(store_path):
        if file.endswith('.npz'):
            with np.load(os.path.join(store_path, file)) as loaded:
                yield loaded['arr_0'], loaded['arr_1'], loaded['arr_2']


class ReplayPool:
    def __init__(self, ext_config):
        self.data_pool = None
        self.training_started = False
        self.pool_capacity = ext_config['pool_size']
        self.pool_end_index = 0
        self.is_full = False
        self.start_data_size = ext_config['start_data_size']
        self.store_path = ext_config.get('store_path')
        self.load_prev = ext_config.get('load_prev')
        self.count = 0
        if self.load_prev and self.store_path is not None:
            if not os.path.isdir(self.store_path):
                os.mkdir(self.store_path)
            for data in load_games(self.store_path):
                self.merge_data(data)
                self.count += 1

    def merge_data(self, data):
        """Appends the rows of one game at the ring's write position, wrapping over the oldest rows; the position and
        the `is_full` / `training_started` flags follow datapool.py:87-112 (a batch ending exactly at the capacity
        leaves the position at the capacity and only the next batch marks the ring full)."""
        n, cap, end = len(data[0]), self.pool_capacity, self.pool_end_index
        if n > cap:
            raise ValueError("a game of %d rows does not fit a pool of %d" % (n, cap))
        if self.data_pool is None:
            self.data_pool = [np.zeros((cap,) + np.shape(item)[1:]) for item in data]
        where = (end + np.arange(n)) % cap
        for ring, item in zip(self.data_pool, data):
            ring[where] = item
        wrapped = end + n > cap
        self.is_full = self.is_full or wrapped
        self.pool_end_index = end + n - cap if wrapped else end + n
        if self.is_full or self.pool_end_index > self.start_data_size:
            self.training_started = True

    def put(self, data):
        self.merge_data(data)
        if self.store_path is not None:
            save_game(self.store_path, self.count, data)
        self.count += 1

    def get(self, batch_size):
        hi = self.pool_capacity if self.is_full else self.pool_end_index
        idxs = np.random.choice(range(hi), batch_size)
        return self.data_pool[0][idxs], self.data_pool[1][idxs], self.data_pool[2][idxs]
