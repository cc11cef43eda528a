"""Worker of tests/test_gpu_multiproc.py: one rank of a two-rank self-play job on one GPU (gloo for the exchanges, since
RCCL refuses two ranks on one card).  Sequence: broadcast weights A -> games start -> mid-game, rank 0 broadcasts weights B
and every rank loads them -> the next round's leaves are evaluated by B on every rank -> games finish -> finished games go
to rank 0 in the packed form, straight from device tensors.  Prints one JSON line per rank."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from alphazero_amd import parallel
    from alphazero_amd.engine import Engine
    from alphazero_amd.network.model import fold_to_blob, random_net
    H, G, P, NB, C = 5, 24, 12, 1, 128
    O = H * H + 1
    kw = dict(num_blocks=NB, channels=C, tower_dtype="fp32")
    eng = Engine("go", H, H, n_slots=G, max_playout=P, transform_types=8, max_turn=20, seed=100 + rank, record_trajectories=True, **kw)
    blob_a = fold_to_blob(random_net(H, H, O, NB, C, seed=1, perturb_bn=True)) if rank == 0 else None
    blob_b = fold_to_blob(random_net(H, H, O, NB, C, seed=2, perturb_bn=True)) if rank == 0 else None
    dev = torch.device("cuda", 0)
    # device path end to end: the broadcast's result is loaded into the tower from GPU memory (az_tower_load_weights_dev)
    blob_a = parallel.update_weights(eng, blob_a, dist, dev).cpu().numpy()
    eng.selfplay_start()
    eng.selfplay_run(3 * (P + 1) + 4)                      # three plies in, mid-search
    plies0, done0 = eng.selfplay_status()
    blob_b = parallel.update_weights(eng, blob_b, dist, dev).cpu().numpy()   # the model update arrives while every slot is mid-game
    eng.selfplay_run(1)
    planes, _ = eng.leaf_batch()
    p, v = eng.leaf_outputs()
    probe = Engine("go", H, H, n_slots=G, **kw)
    used = {}
    for name, blob in (("a", blob_a), ("b", blob_b)):
        probe.tower_load(blob)
        rp, rv = probe.tower_forward(planes)
        used[name] = bool(np.array_equal(p, rp) and np.array_equal(v, rv))
    probe.close()
    for _ in range(200):
        eng.selfplay_run(64)
        plies, done = eng.selfplay_status()
        if done.all():
            break
    fin = np.flatnonzero(done)
    moves, visits, n_rows, win = eng.trajectory_pop_many_torch(fin, plies[fin])
    packed = parallel.pack_sparse(moves, visits, n_rows, win)             # on the GPU
    local = parallel.unpack_trajectories(packed, O)
    host = eng.trajectory_pop_many(fin, plies[fin])
    same_local = len(local) == len(host) and all(np.array_equal(a["moves"], b["moves"]) and np.array_equal(a["visits"], b["visits"]) and
                                                 a["winner"] == b["winner"] for a, b in zip(local, host))
    blobs = parallel.gather_packed(packed, dist, dst=0)   # device tensor in, device tensors out (gloo stages through the host, RCCL does not)
    got = [t for b in blobs for t in parallel.unpack_trajectories(b, O)]
    digest = [int(t["visits"].sum()) * 31 + int(t["moves"].astype(np.int64).sum()) for t in local]
    all_digests = [None] * world
    dist.all_gather_object(all_digests, digest)
    ok_gather = True
    if rank == 0:
        flat = [d for ds in all_digests for d in ds]
        ok_gather = flat == [int(t["visits"].sum()) * 31 + int(t["moves"].astype(np.int64).sum()) for t in got]
    print(json.dumps({"rank": rank, "mid_game": bool((plies0 >= 1).all() and not done0.any()), "used": used,
                      "finished": int(done.sum()), "same_local": bool(same_local), "n_received": len(got), "gather_ok": bool(ok_gather),
                      "packed_bytes": parallel.packed_nbytes(packed), "dense_bytes": int(visits.numel() * 4 + moves.numel() * 2),
                      "weights_equal": bool(np.array_equal(blob_b, fold_to_blob(random_net(H, H, O, NB, C, seed=2, perturb_bn=True))))}),
          flush=True)
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
