"""CPU tests of the multi-process exchanges with the gloo backend, world_size 2: weights from rank 0 to every rank, finished
games from every rank to rank 0 in the packed (sparse visit rows) form, and the model-update sequence."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _games(rank, O, shapes, max_count):
    """Seeded ragged games: visit rows with a few dozen non-zero entries, as searches leave them."""
    rng = np.random.RandomState(100 + rank)
    out = []
    for T in shapes:
        visits = np.zeros((T, O), np.int32)
        for t in range(T):
            k = rng.randint(1, min(O, 60) + 1)
            visits[t, rng.choice(O, k, replace=False)] = rng.randint(1, max_count, k)
        out.append(dict(moves=rng.randint(0, O, T).astype(np.int16), visits=visits, winner=int(rng.randint(-1, 2))))
    return out


SHAPES = {0: [401, 3, 0, 250], 1: [77, 401, 401], 2: [], 3: [12, 401]}      # ragged, up to max_turn + 1 = 401 plies of Go 19x19


class _StubEngine:
    """What update_weights needs of an engine."""

    def __init__(self, n):
        self.n, self.loaded = n, []

    def tower_blob_floats(self):
        return self.n

    def tower_load(self, blob, net=0):
        self.loaded.append((net, np.array(blob, copy=True)))


def _same(a, b):
    return len(a) == len(b) and all(np.array_equal(x["moves"], y["moves"]) and np.array_equal(x["visits"], y["visits"]) and
                                    x["winner"] == y["winner"] for x, y in zip(a, b))


def _worker(rank, world, port, q):
    try:
        _work(rank, world, port, q)
    except Exception:          # report instead of leaving the parent to time out
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))


def _work(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from alphazero_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    # 1. weights: rank 0 -> all
    n = 1000
    blob = np.arange(n, dtype=np.float32) * 0.5 if rank == 0 else None
    res["weights"] = bool(np.array_equal(parallel.broadcast_weights(blob, n, dist), np.arange(n, dtype=np.float32) * 0.5))
    # 2. finished games: all -> rank 0, ragged 401-ply x 362 games, small chunks so that several transfers are needed
    O = 362
    mine = _games(rank, O, SHAPES[rank], 900)
    packed = parallel.pack_trajectories(mine, O)
    dense = sum(t["visits"].nbytes + t["moves"].nbytes for t in mine)
    res["ratio"] = parallel.packed_nbytes(packed) / max(dense, 1)
    blobs = parallel.gather_packed(packed, dist, dst=0, chunk_bytes=100001)   # odd on purpose: chunks stay whole int16s
    got = [t for b in blobs for t in parallel.unpack_trajectories(b, O)]
    exp = [t for r in range(world) for t in _games(r, O, SHAPES[r], 900)] if rank == 0 else []
    res["gather"] = _same(got, exp)
    res["n_got"] = len(got)
    # counts beyond int16 switch to 32-bit counts
    big = _games(rank, 30, [5, 2], 100000)
    allb = parallel.gather_trajectories(big, 30, dist)
    res["wide"] = _same(allb, [t for r in range(world) for t in _games(r, 30, [5, 2], 100000)] if rank == 0 else [])
    res["empty"] = len(parallel.gather_trajectories([], O, dist))
    # 3. model update: rank 0 has new weights, every rank's engine loads them
    eng = _StubEngine(64)
    new = np.linspace(-1, 1, 64).astype(np.float32) if rank == 0 else None
    parallel.update_weights(eng, new, dist)
    res["wire_dtype"] = str(parallel.wire_bytes(packed).dtype)     # what send / recv see: RCCL has no int16
    res["update"] = len(eng.loaded) == 1 and eng.loaded[0][0] == 0 and bool(np.array_equal(eng.loaded[0][1], np.linspace(-1, 1, 64).astype(np.float32)))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 4])
def test_broadcast_and_gather(world):
    """world 2, and world 4 with a rank that has nothing to send: rank 0 takes the ranks' blobs one after the other while
    the others wait their turn."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
    total = sum(len(SHAPES[r]) for r in range(world))
    for rank in range(world):
        r = res[rank]
        assert "error" not in r, r["error"]
        assert r["weights"] and r["gather"] and r["wide"] and r["update"] and r["empty"] == 0, (rank, r)
        assert r["wire_dtype"] == "torch.uint8"
        assert r["n_got"] == (total if rank == 0 else 0)      # only rank 0, the replay buffer's home, receives
        if SHAPES[rank]:
            assert r["ratio"] < 0.25                          # packed rows against dense int32 rows


def test_pack_roundtrip_and_sharding():
    from alphazero_amd import parallel
    O = 7
    tr = [dict(moves=np.array([1, 2, 6], np.int16), visits=np.arange(21, dtype=np.int32).reshape(3, 7), winner=-1),
          dict(moves=np.zeros(0, np.int16), visits=np.zeros((0, 7), np.int32), winner=0),
          dict(moves=np.array([0], np.int16), visits=np.array([[0, 0, 70000, 0, 1, 0, 0]], np.int32), winner=1)]
    back = parallel.unpack_trajectories(parallel.pack_trajectories(tr, O), O)
    assert _same(back, tr)
    assert parallel.unpack_trajectories(parallel.pack_trajectories([], O), O) == []
    shards = [parallel.shard_slots(10, 4, r) for r in range(4)]
    assert sorted(sum(shards, [])) == list(range(10)) and [len(s) for s in shards] == [3, 3, 2, 2]
    assert parallel.gather_trajectories(tr, O) == tr
    import pytest
    with pytest.raises(ValueError):
        parallel.unpack_trajectories(torch.zeros(40, dtype=torch.int16), O)
