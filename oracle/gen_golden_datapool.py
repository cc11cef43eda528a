"""TEST INFRASTRUCTURE - NOT PRODUCT CODE.

Runs the reference's DataPool.merge_data / file writer (/root/reference, read-only, this container only)
on a seeded sequence of small games and records the ring contents after every put plus one sampled
batch: tests/golden/datapool_ring.npz.  No server process is started.
"""
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import AlphaZero.train.parallel.datapool as datapool  # noqa: E402

datapool.printlog = lambda *a, **k: None


def main():
    rng = np.random.RandomState(5)
    cap, start = 23, 9
    pool = datapool.DataPool({'pool_size': cap, 'start_data_size': start, 'conn_num': 1})
    sizes = [4, 7, 5, 9, 3, 11, 2]
    out = {"capacity": cap, "start_data_size": start, "sizes": np.array(sizes)}
    for k, T in enumerate(sizes):
        data = (rng.randint(0, 2, (T, 17, 3, 3)).astype(np.float64), rng.rand(T, 9), rng.choice([-1.0, 1.0], T))
        for j, a in enumerate(data):
            out["put%d_arr%d" % (k, j)] = a
        pool.merge_data(data)
        for j in range(3):
            out["after%d_pool%d" % (k, j)] = pool.data_pool[j].copy()
        out["after%d_end" % k] = pool.pool_end_index
        out["after%d_full" % k] = int(pool.is_full)
        out["after%d_started" % k] = int(pool.training_started)
    # the 'get' request body (datapool.py:72-76) with a seeded global RNG
    np.random.seed(123)
    idxs = np.random.choice(range(pool.pool_capacity if pool.is_full else pool.pool_end_index), 6)
    out["get_idxs"] = idxs
    for j in range(3):
        out["get_arr%d" % j] = pool.data_pool[j][idxs]
    # file format: what np.load sees for one stored game
    with tempfile.TemporaryDirectory() as d:
        data = tuple(out["put0_arr%d" % j] for j in range(3))
        np.savez_compressed(os.path.join(d, str(0)), *data)  # datapool.py:66
        with np.load(os.path.join(d, "0.npz")) as z:
            out["file_keys"] = np.array(sorted(z.files))
    np.savez_compressed(os.path.join(OUT, "datapool_ring.npz"), **out)
    print("wrote datapool_ring.npz")


if __name__ == "__main__":
    main()
