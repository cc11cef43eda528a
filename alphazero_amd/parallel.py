"""Multi-GPU glue: one process per GPU, games sharded by slot, no collective on the search path.

Two exchanges exist (they replace the reference's checkpoint-path notification and its
data_queue / TCP trajectory hand-off, AlphaZero/train/parallel/selfplay.py:82-85,110-122):
  * broadcast_weights  - rank 0's folded weight blob to every rank (RCCL broadcast on GPU tensors;
                          gloo on CPU tensors in tests)
  * gather_trajectories - finished games of every rank to every rank (all_gather of packed arrays)
"""
import numpy as np
import torch


def broadcast_weights(blob, n_floats, dist=None, device=None):
    """blob: float32 numpy on rank 0 (ignored elsewhere).  Returns the blob on every rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        if blob is None:
            raise ValueError("rank 0 must supply the weight blob")
        return np.ascontiguousarray(blob, np.float32)
    dev = device if device is not None else torch.device("cpu")
    if dist.get_rank() == 0:
        if blob is None or blob.size != n_floats:
            raise ValueError("rank 0 blob has the wrong size")
        t = torch.from_numpy(np.ascontiguousarray(blob, np.float32)).to(dev)
    else:
        t = torch.empty(n_floats, dtype=torch.float32, device=dev)
    dist.broadcast(t, src=0)
    return t.cpu().numpy()


def pack_trajectories(trajs, O):
    """[{moves int16[T], visits int32[T,O], winner}] -> flat int32 array:
    n_games, then per game: T, winner, moves[T], visits[T*O]."""
    parts = [np.array([len(trajs)], np.int32)]
    for tr in trajs:
        T = len(tr["moves"])
        assert tr["visits"].shape == (T, O)
        parts.append(np.array([T, tr["winner"]], np.int32))
        parts.append(np.asarray(tr["moves"], np.int32))
        parts.append(np.asarray(tr["visits"], np.int32).ravel())
    return np.concatenate(parts)


def unpack_trajectories(flat, O):
    flat = np.asarray(flat, np.int32)
    n, pos, out = int(flat[0]), 1, []
    for _ in range(n):
        T, w = int(flat[pos]), int(flat[pos + 1]); pos += 2
        moves = flat[pos:pos + T].astype(np.int16); pos += T
        visits = flat[pos:pos + T * O].reshape(T, O).copy(); pos += T * O
        out.append(dict(moves=moves, visits=visits, winner=w))
    return out


def gather_trajectories(trajs, O, dist=None, device=None):
    """All ranks contribute their finished games; every rank gets the concatenation (rank order)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(trajs)
    dev = device if device is not None else torch.device("cpu")
    flat = torch.from_numpy(pack_trajectories(trajs, O)).to(dev)
    n = torch.tensor([flat.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(dist.get_world_size())]
    dist.all_gather(sizes, n)
    mx = int(max(s.item() for s in sizes))
    pad = torch.zeros(mx, dtype=torch.int32, device=dev)
    pad[:flat.numel()] = flat
    bufs = [torch.zeros(mx, dtype=torch.int32, device=dev) for _ in range(dist.get_world_size())]
    dist.all_gather(bufs, pad)
    out = []
    for s, b in zip(sizes, bufs):
        out.extend(unpack_trajectories(b[:int(s.item())].cpu().numpy(), O))
    return out


def shard_slots(total_games, world, rank):
    """Games are independent units: game id -> rank by id % world (remainder to the low ranks)."""
    return [g for g in range(total_games) if g % world == rank]
