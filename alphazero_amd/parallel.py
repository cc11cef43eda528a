"""Multi-GPU glue: one process per GPU, games sharded by slot, no collective on the search path.

Two exchanges exist (they replace the reference's checkpoint-path notification and its
data_queue / TCP trajectory hand-off, AlphaZero/train/parallel/selfplay.py:82-85,110-122):
  * broadcast_weights   - rank 0's folded weight blob to every rank (RCCL broadcast on GPU tensors, loaded into the tower
                          from GPU memory; gloo on CPU tensors in tests)
  * gather_trajectories - finished games of every rank to the rank that owns the replay buffer (rank 0), in a
                          compact form: moves int16, visit counts as sparse (action, count) pairs.  A dense
                          int32 row of a 19x19 game is 1448 bytes per ply and at most max_playout + 1 of its 362
                          entries are non-zero (typically a few dozen); 4000 finished games of 400 plies are
                          2.3 GB dense and 0.1 - 0.4 GB packed.  Everything stays a torch tensor on the caller's
                          device until rank 0 unpacks it; transfers are point-to-point, chunked, as bytes.
"""
import numpy as np
import torch


def _is_gloo(dist):
    return str(dist.get_backend()).lower() == "gloo"


def _wire_device(dist, device):
    """Where a tensor has to live for this backend to move it: RCCL ("nccl") takes tensors in GPU memory as they are; gloo
    (CPU tests, and the two-ranks-on-one-card rehearsal, where RCCL refuses a second rank per device) takes host tensors."""
    if _is_gloo(dist):
        return torch.device("cpu")
    if device is None or torch.device(device).type != "cuda":
        raise ValueError("the nccl backend moves GPU tensors: pass the rank's cuda device")
    return torch.device(device)


def broadcast_weights(blob, n_floats, dist=None, device=None, as_tensor=False):
    """blob: float32 numpy (or tensor) on rank 0, ignored elsewhere.  Every rank gets the blob: as a float32 numpy array, or,
    with as_tensor=True, as a tensor on `device` - with RCCL that is the broadcast's own receive buffer, so on a GPU rank the
    weights never pass through host memory (Engine.tower_load_torch takes it from there)."""
    single = dist is None or not dist.is_initialized() or dist.get_world_size() == 1
    dev = torch.device(device) if device is not None else torch.device("cpu")
    if single or dist.get_rank() == 0:
        if blob is None:
            raise ValueError("rank 0 must supply the weight blob")
        src = blob if torch.is_tensor(blob) else torch.from_numpy(np.ascontiguousarray(blob, np.float32))
        if src.numel() != n_floats or src.dtype != torch.float32:
            raise ValueError("rank 0 blob has the wrong size or type")
    if single:
        t = src.to(dev).contiguous() if as_tensor else src
        return t if as_tensor else t.cpu().numpy()
    wire = _wire_device(dist, dev)
    t = src.to(wire).contiguous() if dist.get_rank() == 0 else torch.empty(n_floats, dtype=torch.float32, device=wire)
    dist.broadcast(t, src=0)
    if as_tensor:
        return t.to(dev)
    return t.cpu().numpy()


def update_weights(engine, blob, dist=None, device=None, net=0):
    """The model update of selfplay.py:110-122 for every rank's engine: rank 0 passes the new folded blob (others None),
    every rank receives it and loads it into its tower; games in flight go on with the new network from the next round.
    With a cuda `device` the blob stays in GPU memory from the broadcast into the tower's weight arrays
    (az_tower_load_weights_dev).  Returns the blob as this rank holds it (tensor on `device`, or numpy without one)."""
    on_gpu = device is not None and torch.device(device).type == "cuda" and hasattr(engine, "tower_load_torch")
    got = broadcast_weights(blob, engine.tower_blob_floats(), dist, device, as_tensor=on_gpu)
    if on_gpu:
        engine.tower_load_torch(got, net=net)
    else:
        engine.tower_load(got, net=net)
    return got


# ---- packed trajectories --------------------------------------------------------------------------------------------
# One int16 tensor per rank:
#   header int32[8]: magic, n_games, total_rows, total_nnz, O, count_words (1: int16 counts, 2: int32 counts), 0, 0
#   plies int32[n_games] | winners int32[n_games] | moves int16[total_rows] | row_nnz int16[total_rows] |
#   cols int16[total_nnz] | counts int16[total_nnz] or int32[total_nnz]
_MAGIC = 0x415A5452  # "AZTR"


def _as_i16(t):
    if t.numel() == 0:
        return torch.empty(0, dtype=torch.int16, device=t.device)
    return t.contiguous().view(torch.int16)


def pack_sparse(moves, visits, plies, winners):
    """moves int16 (rows,), visits int32 (rows, O): the games' plies back to back (az_trajectory_pop_many layout);
    plies / winners: per game (any int array-like).  Tensors may live on any device; the result lives there too."""
    dev = visits.device
    rows, O = visits.shape
    plies_t = torch.as_tensor(np.asarray(plies, np.int32)).to(dev)
    win_t = torch.as_tensor(np.asarray(winners, np.int32)).to(dev)
    if int(plies_t.sum()) != rows or moves.numel() != rows:
        raise ValueError("plies do not add up to the number of rows")
    nz = visits != 0
    row_nnz = nz.sum(dim=1).to(torch.int16)
    idx = nz.nonzero(as_tuple=False)              # row-major: rows ascending, columns ascending within a row
    cols = idx[:, 1].to(torch.int16)
    counts = visits[nz]
    wide = bool(counts.numel()) and int(counts.max()) > 32767
    head = torch.tensor([_MAGIC, plies_t.numel(), rows, counts.numel(), O, 2 if wide else 1, 0, 0], dtype=torch.int32, device=dev)
    parts = [_as_i16(head), _as_i16(plies_t), _as_i16(win_t), moves.to(torch.int16).contiguous(), row_nnz, cols,
             _as_i16(counts.to(torch.int32)) if wide else counts.to(torch.int16)]
    return torch.cat(parts)


def pack_trajectories(trajs, O, device=None):
    """[{moves int16[T], visits int32[T,O], winner}] (host arrays, as Engine.trajectory_pop_many returns) -> packed tensor."""
    dev = device if device is not None else torch.device("cpu")
    if trajs:
        moves = np.concatenate([np.asarray(t["moves"], np.int16) for t in trajs])
        visits = np.concatenate([np.asarray(t["visits"], np.int32).reshape(-1, O) for t in trajs])
    else:
        moves, visits = np.zeros(0, np.int16), np.zeros((0, O), np.int32)
    return pack_sparse(torch.from_numpy(moves).to(dev), torch.from_numpy(visits).to(dev),
                       [len(t["moves"]) for t in trajs], [t["winner"] for t in trajs])


def unpack_trajectories(blob, O=None):
    """Packed tensor -> [{moves int16[T], visits int32[T,O], winner}] with host arrays, one game at a time (the dense
    rows of all games together never exist)."""
    b = blob.detach().cpu().contiguous()
    head = b[:16].view(torch.int32).tolist()
    if head[0] != _MAGIC:
        raise ValueError("not a packed trajectory blob")
    n, rows, nnz, o, words = head[1], head[2], head[3], head[4], head[5]
    if O is not None and O != o:
        raise ValueError("blob was packed for %d actions, not %d" % (o, O))
    pos = 16

    def take(count, w=1, dtype=torch.int16):
        nonlocal pos
        t = b[pos:pos + count * w]
        pos += count * w
        return (t.clone().view(dtype) if w == 2 else t).numpy()   # (clone: a view needs an aligned storage offset)

    plies = take(n, 2, torch.int32); winners = take(n, 2, torch.int32)
    moves = take(rows); row_nnz = take(rows).astype(np.int64); cols = take(nnz).astype(np.int64)
    counts = take(nnz, 2, torch.int32) if words == 2 else take(nnz).astype(np.int32)
    if pos != b.numel():
        raise ValueError("packed trajectory blob has the wrong length")
    nnz_start = np.concatenate([[0], np.cumsum(row_nnz)])
    out, r0 = [], 0
    for g in range(n):
        T = int(plies[g])
        visits = np.zeros((T, o), np.int32)
        lo, hi = int(nnz_start[r0]), int(nnz_start[r0 + T])
        rr = np.repeat(np.arange(T), row_nnz[r0:r0 + T])
        visits[rr, cols[lo:hi]] = counts[lo:hi]
        out.append(dict(moves=moves[r0:r0 + T].copy(), visits=visits, winner=int(winners[g])))
        r0 += T
    return out


def gather_packed(blob, dist=None, dst=0, chunk_bytes=64 << 20):
    """Every rank's packed tensor to rank `dst`: a tiny all_gather of the lengths, then point-to-point transfers in chunks of
    chunk_bytes (64 MiB).  Returns the list of int16 tensors in rank order on dst, [] elsewhere.
    On the wire the blob is plain bytes (a uint8 view): RCCL has no 16-bit integer type.  With RCCL the tensors stay in GPU
    memory end to end; with gloo a GPU tensor is staged through host memory."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [blob]
    if blob.dtype != torch.int16:
        raise ValueError("a packed trajectory blob is an int16 tensor")
    world, rank, dev = dist.get_world_size(), dist.get_rank(), blob.device
    wire = _wire_device(dist, dev)
    chunk = max(2, int(chunk_bytes) & ~1)                       # whole int16 elements per chunk
    n = torch.tensor([blob.numel() * 2], dtype=torch.int64, device=wire)
    sizes = [torch.zeros(1, dtype=torch.int64, device=wire) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    if rank != dst:
        raw = wire_bytes(blob).to(wire)
        for a in range(0, raw.numel(), chunk):
            dist.send(raw[a:a + chunk], dst=dst)
        return []
    out = []
    for r in range(world):
        if r == dst:
            out.append(blob)
            continue
        buf = torch.empty(sizes[r], dtype=torch.uint8, device=wire)
        for a in range(0, sizes[r], chunk):
            dist.recv(buf[a:a + chunk], src=r)
        out.append(buf.view(torch.int16).to(dev))
    return out


def wire_bytes(blob):
    """The uint8 view of a packed blob that goes over the wire (same memory, no copy)."""
    return blob.contiguous().view(torch.uint8)


def gather_trajectories(trajs, O, dist=None, device=None, dst=0):
    """All ranks contribute their finished games; rank `dst` (the replay buffer's home, datapool.py:114-121) gets the
    concatenation in rank order, the others an empty list."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(trajs)
    blobs = gather_packed(pack_trajectories(trajs, O, device), dist, dst)
    out = []
    for b in blobs:
        out.extend(unpack_trajectories(b, O))
    return out


def packed_nbytes(blob):
    return blob.numel() * 2


def shard_slots(total_games, world, rank):
    """Games are independent units: game id -> rank by id % world (remainder to the low ranks)."""
    return [g for g in range(total_games) if g % world == rank]
