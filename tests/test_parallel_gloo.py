"""CPU tests of the multi-process exchanges with the gloo backend, world_size 2."""
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


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from alphazero_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1000
    blob = np.arange(n, dtype=np.float32) * 0.5 if rank == 0 else None
    got = parallel.broadcast_weights(blob, n, dist)
    ok_w = bool(np.array_equal(got, np.arange(n, dtype=np.float32) * 0.5))
    O = 10
    rng = np.random.RandomState(rank)
    mine = []
    for g in range(rank + 1):  # ragged: rank 0 has one game, rank 1 two
        T = 3 + g + rank
        mine.append(dict(moves=rng.randint(0, O, T).astype(np.int16), visits=rng.randint(0, 50, (T, O)).astype(np.int32),
                         winner=int(rng.randint(-1, 2))))
    allg = parallel.gather_trajectories(mine, O, dist)
    exp = []
    for r in range(world):
        rr = np.random.RandomState(r)
        for g in range(r + 1):
            T = 3 + g + r
            exp.append(dict(moves=rr.randint(0, O, T).astype(np.int16), visits=rr.randint(0, 50, (T, O)).astype(np.int32),
                            winner=int(rr.randint(-1, 2))))
    ok_t = len(allg) == len(exp) and all(
        np.array_equal(a["moves"], b["moves"]) and np.array_equal(a["visits"], b["visits"]) and a["winner"] == b["winner"]
        for a, b in zip(allg, exp))
    empty = parallel.gather_trajectories([], O, dist)
    q.put((rank, ok_w, ok_t, len(empty)))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in ps:
        p.join(timeout=60)
    for rank, ok_w, ok_t, n_empty in res:
        assert ok_w and ok_t and n_empty == 0, (rank, ok_w, ok_t, n_empty)


def test_pack_roundtrip_and_sharding():
    from alphazero_amd import parallel
    O = 7
    tr = [dict(moves=np.array([1, 2, 6], np.int16), visits=np.arange(21, dtype=np.int32).reshape(3, 7), winner=-1),
          dict(moves=np.zeros(0, np.int16), visits=np.zeros((0, 7), np.int32), winner=0)]
    back = parallel.unpack_trajectories(parallel.pack_trajectories(tr, O), O)
    assert len(back) == 2 and np.array_equal(back[0]["visits"], tr[0]["visits"]) and back[1]["winner"] == 0
    shards = [parallel.shard_slots(10, 4, r) for r in range(4)]
    assert sorted(sum(shards, [])) == list(range(10)) and [len(s) for s in shards] == [3, 3, 2, 2]
    assert parallel.gather_trajectories(tr, O) == tr
