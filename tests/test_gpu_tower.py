"""GPU numerics: HIP tower vs the fp32 CPU restatement of network/model.py (oracle/tower_ref.py).
Tolerances: 1e-5 abs for fp32, 1e-2 abs for the 16-bit towers (bf16 and half precision), on softmax priors and the tanh
value.  Where a case cannot hold the literal 1e-2 (the bf16 value at 10 and 20 blocks) its test id says so."""
import numpy as np
import pytest
import torch

from oracle import tower_ref

pytestmark = pytest.mark.gpu

TOL = {"fp32": 1e-5, "bf16": 1e-2, "f16": 1e-2}


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "f16"])
@pytest.mark.parametrize("game,H,W,NB,C,B", [
    ("mnk", 3, 3, 2, 128, 5),
    ("reversi", 8, 8, 3, 128, 9),
    ("go", 9, 9, 2, 256, 7),
    ("go", 19, 19, 2, 256, 3),
    ("mnk", 6, 5, 1, 128, 4),
    # batches past the small-batch kernels: the slab kernel (256 channels) and the generic tile kernel
    ("go", 19, 19, 1, 256, 20),
    ("go", 9, 9, 1, 256, 90),
    ("reversi", 8, 8, 1, 256, 100),
    ("mnk", 15, 15, 1, 256, 30),
    ("mnk", 3, 3, 1, 256, 700),
    ("go", 9, 9, 1, 128, 90),
])
def test_tower_matches_fp32_reference(dtype, game, H, W, NB, C, B):
    from alphazero_amd.engine import Engine
    from alphazero_amd.network.model import fold_to_blob, random_net
    O = H * W + (0 if game == "mnk" else 1)
    net = random_net(H, W, O, NB, C, seed=H * 100 + NB, perturb_bn=True)
    eng = Engine(game, H, W, 3, n_slots=4 if B < 16 else B, num_blocks=NB, channels=C, tower_dtype=dtype)
    blob = fold_to_blob(net)
    assert blob.size == eng.tower_blob_floats()
    eng.tower_load(blob)
    rng = np.random.RandomState(0)
    x = (rng.rand(B, 17, H, W) < 0.3).astype(np.int8)
    p, v = eng.tower_forward(x)  # B > n_slots exercises the chunked path
    rp, rv = tower_ref.forward(net.state_dict(), torch.from_numpy(x).float(), NB)
    rp, rv = rp.numpy(), rv.numpy()
    assert np.all(np.isfinite(p)) and np.all(np.isfinite(v))
    assert np.abs(p - rp).max() <= TOL[dtype], np.abs(p - rp).max()
    assert np.abs(v - rv).max() <= TOL[dtype], np.abs(v - rv).max()
    assert np.allclose(p.sum(axis=1), 1.0, atol=1e-4)
    eng.close()


def _boards(B, H, W, seed=0):
    return (np.random.RandomState(seed).rand(B, 17, H, W) < 0.3).astype(np.int8)


def _assert_close(p, v, net, NB, x, tol):
    rp, rv = tower_ref.forward(net.state_dict(), torch.from_numpy(x).float(), NB)
    assert np.all(np.isfinite(p)) and np.all(np.isfinite(v))
    assert np.abs(p - rp.numpy()).max() <= tol, np.abs(p - rp.numpy()).max()
    assert np.abs(v - rv.numpy()).max() <= tol, np.abs(v - rv.numpy()).max()


# Batches of more than 128 conv tiles (32768 board points) run the instantiations self-play and the benchmark use: two
# workgroups per tile (128 output channels each) with a slab of 24 pieces of 16 positions (19x19) or 27 pieces (the other
# boards), together with the one-chunk stem form of the one-workgroup kernel; 64 boards of 19x19 run that kernel's
# two-way channel split.  The network is evaluated on the whole batch, the fp32
# restatement on boards from its start, middle and (partly filled) last tile.
@pytest.mark.parametrize("game,H,B", [
    ("go", 19, 96),        # 136 tiles, slab of 24 pieces
    ("go", 9, 410),        # 130 tiles, slab of 27 pieces (a tile spans up to 426 positions)
    ("reversi", 8, 520),   # 130 tiles, 27 pieces (422 positions)
    ("mnk", 15, 150),      # 132 tiles, 27 pieces (394 positions)
    ("go", 19, 64),        # 91 tiles: two workgroups per tile (32 channels per wave)
])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_full_tile_conv_kernels_match_fp32_reference(dtype, game, H, B):
    from alphazero_amd.engine import Engine
    from alphazero_amd.network.model import fold_to_blob, random_net
    O = H * H + (0 if game == "mnk" else 1)
    NB = 1
    net = random_net(H, H, O, NB, 256, seed=H + B, perturb_bn=True)
    eng = Engine(game, H, H, 5, n_slots=B, num_blocks=NB, channels=256, tower_dtype=dtype)
    eng.tower_load(fold_to_blob(net))
    x = _boards(B, H, H, seed=B)
    p, v = eng.tower_forward(x)
    pick = np.array([0, 1, 2, B // 2, B // 2 + 1, B - 3, B - 2, B - 1])
    _assert_close(p[pick], v[pick], net, NB, x[pick], TOL[dtype])
    assert np.allclose(p.sum(axis=1), 1.0, atol=1e-4)
    eng.close()


def test_device_count_and_row_list_path_matches_fp32_reference():
    """A search round whose tower pass takes its rows and their number from device memory (leaf compaction): 300 of
    512 slots search, the tower runs on the 300 listed rows in place, and what it wrote for those leaves is compared
    with the fp32 restatement evaluated on the planes the engine reported for them."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.network.model import fold_to_blob, random_net
    H, G, NB = 9, 512, 1
    rng = np.random.RandomState(5)
    net = random_net(H, H, H * H + 1, NB, 256, seed=11, perturb_bn=True)
    eng = Engine("go", H, H, n_slots=G, max_playout=4, transform_types=8, num_blocks=NB, channels=256, tower_dtype="bf16",
                 seed=3)
    eng.tower_load(fold_to_blob(net))
    slots = np.sort(rng.choice(G, 300, replace=False)).astype(np.int32)
    for _ in range(8):   # a few random legal stones so that the planes differ from slot to slot
        mask = eng.legal_mask(slots)[:, :H * H]
        moves = [int(rng.choice(np.flatnonzero(m))) for m in mask]
        status, _ = eng.do_move(slots, moves)
        assert np.all(status == 0)
    eng.search_begin(slots, np.zeros(300, np.int32), np.zeros(300, np.int32))
    assert eng.step_select() == 300
    planes, valid = eng.leaf_batch()
    assert np.array_equal(np.flatnonzero(valid), slots)
    eng.step_finish()                      # tower on the listed rows, then expand / backup
    p, v = eng.leaf_outputs()
    pick = slots[[0, 1, 2, 150, 151, 297, 298, 299]]
    _assert_close(p[pick], v[pick], net, NB, planes[pick], TOL["bf16"])
    assert np.allclose(p[slots].sum(axis=1), 1.0, atol=1e-4)
    # the tree got exactly these priors: root children of a searched slot, in the evaluator's index order
    # (transform id of the leaf applied), carry p of the legal actions
    a, n, w, pr, rn = eng.root_stats(int(slots[0]), 0)
    assert len(a) > 0 and np.isin(pr, p[slots[0]]).all()
    eng.close()


# Large batches go through the tower in groups of boards, each group through all layers over the same two activation
# buffers (the second convolution of a block writes over its residual input).  AZ_TOWER_GROUP_BOARDS forces a group size
# (read when the engine is created; -1 = the whole batch per layer).
@pytest.mark.parametrize("game,H,B,group", [
    ("go", 19, 150, 40),     # groups of 40, 40, 40 and 30 boards: the tile kernels on every group
    ("go", 19, 800, 0),      # automatic: groups of 363, 363 and 74 boards on 256 compute units
    ("go", 9, 700, 130),     # 5 groups of 130 boards and one of 50, slab of 4 x 128 positions
    ("reversi", 8, 64, 7),   # 10 groups of 7 boards, the last one of 1: the kernels for few boards
])
def test_tower_in_board_groups_matches_the_whole_batch(monkeypatch, game, H, B, group):
    from alphazero_amd.engine import Engine
    from alphazero_amd.network.model import fold_to_blob, random_net
    O = H * H + (0 if game == "mnk" else 1)
    NB = 2
    net = random_net(H, H, O, NB, 256, seed=H + B, perturb_bn=True)
    x = _boards(B, H, H, seed=B + 1)
    out = {}
    for g in (-1, group):
        monkeypatch.setenv("AZ_TOWER_GROUP_BOARDS", str(g))
        eng = Engine(game, H, H, 5, n_slots=B, num_blocks=NB, channels=256, tower_dtype="bf16")
        eng.tower_load(fold_to_blob(net))
        out[g] = eng.tower_forward(x)
        eng.close()
    p, v = out[group]
    # the same arithmetic per board whichever group it is in (groups that run the same kernels as the whole batch are
    # bit-identical); a smaller last group may get kernels that order their fp32 sums differently, which flips bf16
    # roundings of some activations: half the bf16 tolerance is allowed for that
    assert np.abs(p - out[-1][0]).max() <= 5e-3 and np.abs(v - out[-1][1]).max() <= 5e-3
    pick = np.unique(np.array([0, 1, B // 3, B // 3 + 1, B // 2, 2 * B // 3, B - 2, B - 1]))
    _assert_close(p[pick], v[pick], net, NB, x[pick], TOL["bf16"])
    assert np.allclose(p.sum(axis=1), 1.0, atol=1e-4)


def test_board_groups_with_device_count_and_row_list(monkeypatch):
    """Leaf compaction (rows and their number in device memory) with the tower in groups of 100 boards: 330 of 600 slots
    search, so three groups are full, one holds 30 boards and two are empty."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.network.model import fold_to_blob, random_net
    monkeypatch.setenv("AZ_TOWER_GROUP_BOARDS", "100")
    H, G, NB, S = 9, 600, 1, 330
    rng = np.random.RandomState(6)
    net = random_net(H, H, H * H + 1, NB, 256, seed=12, perturb_bn=True)
    eng = Engine("go", H, H, n_slots=G, max_playout=4, transform_types=8, num_blocks=NB, channels=256, tower_dtype="bf16", seed=4)
    eng.tower_load(fold_to_blob(net))
    slots = np.sort(rng.choice(G, S, replace=False)).astype(np.int32)
    for _ in range(6):
        mask = eng.legal_mask(slots)[:, :H * H]
        status, _ = eng.do_move(slots, [int(rng.choice(np.flatnonzero(m))) for m in mask])
        assert np.all(status == 0)
    eng.search_begin(slots, np.zeros(S, np.int32), np.zeros(S, np.int32))
    assert eng.step_select() == S
    planes, valid = eng.leaf_batch()
    eng.step_finish()
    p, v = eng.leaf_outputs()
    pick = slots[[0, 1, 99, 100, 101, 199, 200, 299, 300, 301, S - 2, S - 1]]
    _assert_close(p[pick], v[pick], net, NB, planes[pick], TOL["bf16"])
    assert np.allclose(p[slots].sum(axis=1), 1.0, atol=1e-4)
    eng.close()


def _deep_net(H, O, NB, seed):
    """Random weights with the batch-norm statistics of every layer set from a batch of positions (then an offset
    beta), as a trained network's moving averages are.  With untouched statistics a random 20-block tower has
    activations that grow by about 1.5x in variance per block and policy logits of magnitude 50: its softmax then
    amplifies bf16 rounding (a 0.59 prior moved by 0.02 in such a net), which says nothing about the kernels."""
    from alphazero_amd.network.model import CenterBN, calibrate_bn, random_net
    net = random_net(H, H, O, NB, 256, seed=seed)
    calibrate_bn(net, torch.from_numpy(_boards(6, H, H, seed=seed + 100)).float())
    g = torch.Generator().manual_seed(seed)
    for m in net.modules():
        if isinstance(m, CenterBN):
            m.beta.data.copy_(torch.randn(m.beta.shape, generator=g) * 0.1)
    return net


# The depths the configurations name: 6 blocks (Reversi), 10 (Go 9x9, 15x15), 20 (Go 19x19), 256 channels.
# Anchor: the restatement evaluated in float64 (the exact value of the reference's formula).
#   fp32 tower: priors and value within 1e-5.
#   f16 tower (half precision): priors and value within the literal 1e-2 at every depth (measured: 4e-3 and less).
#   bf16 tower: priors within the literal 1e-2 at every depth (measured: 3e-3 and less).  The bf16 VALUE is at the edge of
#   1e-2 already at 6 blocks (8e-3 on the three boards of the first test below, 1.2e-2 on boards of the full-tile test) and
#   1e-2 ... 4e-2 away at 10 and 20 blocks, for any bf16 implementation: a CPU model of the same
#   arithmetic (oracle/tower_ref.forward_bf16_model) is as far from the float64 value as the HIP tower, and no variant of
#   the arithmetic that keeps bf16 operands on the matrix cores gets closer (tests/study_tower_arith_variants.py).  Those cases
#   carry "value_widened" in their id and are held to 1e-2 + 3 x the CPU model's own distance on the same boards (never
#   more than 1e-1): the kernels may not add to what the number format costs.
LITERAL, WIDENED = "literal_1e-2", "value_widened_to_bf16_model"


def _depth_check(p, v, net, NB, x, dtype, H, O, value_bound=LITERAL):
    rp, rv = tower_ref.forward(net.state_dict(), torch.from_numpy(x).float(), NB, dtype=torch.float64)
    rp, rv = rp.numpy(), rv.numpy()
    assert np.all(np.isfinite(p)) and np.all(np.isfinite(v))
    ep, ev = np.abs(p - rp).max(), np.abs(v - rv).max()
    print("%s: value error %.2e, priors error %.2e" % (dtype, ev, ep))
    if dtype == "fp32":
        assert ep <= 1e-5 and ev <= 1e-5, (ep, ev)
        return
    assert ep <= 1e-2, ep
    if value_bound == LITERAL:
        assert ev <= 1e-2, ev
        return
    assert dtype == "bf16"
    from alphazero_amd.network.model import fold_to_blob
    mp, mv = tower_ref.forward_bf16_model(fold_to_blob(net), torch.from_numpy(x).float(), H, H, O, NB, 256)
    assert np.abs(mp.numpy() - p).max() <= 1e-2     # the two bf16 evaluations agree on the priors as well
    bound = min(1e-2 + 3.0 * np.abs(mv.numpy() - rv).max(), 1e-1)
    print("  widened value bound %.2e" % bound)
    assert ev <= bound, (ev, bound)


DEPTHS = [("reversi", 8, 6, 3), ("go", 9, 10, 3), ("mnk", 15, 10, 2), ("go", 19, 20, 2)]


@pytest.mark.parametrize("dtype,value_bound,game,H,NB,B",
                         [("fp32", LITERAL) + d for d in DEPTHS] + [("f16", LITERAL) + d for d in DEPTHS] +
                         [("bf16", LITERAL if d[2] <= 6 else WIDENED) + d for d in DEPTHS])
def test_tower_at_named_depths(dtype, value_bound, game, H, NB, B):
    from alphazero_amd.engine import Engine
    from alphazero_amd.network.model import fold_to_blob
    O = H * H + (0 if game == "mnk" else 1)
    net = _deep_net(H, O, NB, seed=NB)
    eng = Engine(game, H, H, 5, n_slots=4, num_blocks=NB, channels=256, tower_dtype=dtype)
    eng.tower_load(fold_to_blob(net))
    x = _boards(B, H, H, seed=NB)
    p, v = eng.tower_forward(x)
    _depth_check(p, v, net, NB, x, dtype, H, O, value_bound)
    eng.close()


FULL_TILE = [("reversi", 8, 6, 520), ("go", 9, 10, 410), ("go", 19, 20, 96)]


@pytest.mark.parametrize("dtype,value_bound,game,H,NB,B",
                         [("f16", LITERAL) + d for d in FULL_TILE] + [("bf16", WIDENED) + d for d in FULL_TILE])
def test_full_tile_kernels_at_named_depths(dtype, value_bound, game, H, NB, B):
    """The same depths through the full-tile 16-bit kernels (the batch sizes of self-play); three boards are checked."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.network.model import fold_to_blob
    O = H * H + 1
    net = _deep_net(H, O, NB, seed=NB + 1)
    eng = Engine(game, H, H, n_slots=B, num_blocks=NB, channels=256, tower_dtype=dtype)
    eng.tower_load(fold_to_blob(net))
    x = _boards(B, H, H, seed=NB + 1)
    p, v = eng.tower_forward(x)
    pick = np.array([0, B // 2, B - 1])
    _depth_check(p[pick], v[pick], net, NB, x[pick], dtype, H, O, value_bound)
    eng.close()


def test_tower_requires_weights():
    from alphazero_amd.engine import Engine, EngineError
    eng = Engine("mnk", 3, 3, 3, n_slots=2, num_blocks=1, channels=128)
    with pytest.raises(EngineError):
        eng.tower_forward(np.zeros((1, 17, 3, 3), np.int8))
    eng.selfplay_start()
    with pytest.raises(EngineError):
        eng.selfplay_run(1)
    eng.close()


def test_selfplay_with_tower_runs_and_counts():
    """End to end on the device: select -> tower -> finish -> move for a few hundred rounds."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.network.model import fold_to_blob, random_net
    G, P = 8, 20
    eng = Engine("mnk", 3, 3, 3, n_slots=G, max_playout=P, num_blocks=1, channels=128, tower_dtype="bf16",
                 record_trajectories=True, seed=7)
    eng.tower_load(fold_to_blob(random_net(3, 3, 9, 1, 128, seed=3)))
    eng.selfplay_start()
    eng.selfplay_run(9 * (P + 1) + 5)
    plies, done = eng.selfplay_status()
    assert np.all(done == 1) and np.all(plies >= 5) and np.all(plies <= 9)
    c = eng.counters()
    assert c["games_finished"] == G and c["moves"] == plies.sum()
    for s in range(G):
        tr = eng.trajectory_pop(s, restart=True)
        assert len(tr["moves"]) == plies[s] and tr["winner"] in (-1, 0, 1)
        assert np.all(tr["visits"].sum(axis=1) >= P)
    plies2, done2 = eng.selfplay_status()
    assert np.all(done2 == 0) and np.all(plies2 == 0)
    eng.selfplay_run(30)
    eng.close()


def test_tower_forward_on_device_tensors():
    """az_tower_forward_dev: device pointers in, device pointers out, same numbers as the host-buffer entry point."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.network.model import fold_to_blob, random_net
    H, NB, C = 9, 2, 256
    eng = Engine("go", H, H, n_slots=32, num_blocks=NB, channels=C, tower_dtype="bf16")
    eng.tower_load(fold_to_blob(random_net(H, H, H * H + 1, NB, C, seed=2, perturb_bn=True)))
    x = (np.random.RandomState(1).rand(70, 17, H, H) < 0.3).astype(np.int8)   # 70 > n_slots: chunked
    p, v = eng.tower_forward(x)
    pt, vt = eng.tower_forward_torch(torch.from_numpy(x).cuda())
    assert pt.is_cuda and np.array_equal(pt.cpu().numpy(), p) and np.array_equal(vt.cpu().numpy(), v)
    eng.close()


@pytest.mark.parametrize("dtype", ["bf16", "f16", "fp32"])
def test_weights_loaded_from_device_memory_equal_the_host_load(dtype):
    """az_tower_load_weights_dev (padding, bf16 conversion and copies done by kernels on a blob in GPU memory, what the RCCL
    broadcast of a model update delivers) leaves the tower with the same weights as az_tower_load_weights from a host
    array: same outputs bit for bit, for both networks of an engine; a blob of the wrong size or a host pointer is refused."""
    from alphazero_amd.engine import Engine, EngineError
    from alphazero_amd.network.model import fold_to_blob, random_net
    H, NB, C = 9, 3, 256
    O = H * H + 1
    blobs = [fold_to_blob(random_net(H, H, O, NB, C, seed=s, perturb_bn=True)) for s in (11, 12)]
    blobs[1][7] = np.float32(np.nan)                   # the two conversions must agree on a NaN's bit pattern as well
    x = _boards(40, H, H, seed=3)
    host = Engine("go", H, H, n_slots=40, num_blocks=NB, channels=C, tower_dtype=dtype)
    devi = Engine("go", H, H, n_slots=40, num_blocks=NB, channels=C, tower_dtype=dtype)
    host.tower_load(blobs[0])
    devi.tower_load_torch(torch.from_numpy(blobs[0]).cuda())
    p0, v0 = host.tower_forward(x)
    p1, v1 = devi.tower_forward(x)
    assert np.array_equal(p0, p1) and np.array_equal(v0, v1)
    # a second load replaces the first (a model update), and network 1 of a match takes the same path
    host.tower_load(blobs[1]); devi.tower_load_torch(torch.from_numpy(blobs[1]).cuda())
    p0, v0 = host.tower_forward(x); p1, v1 = devi.tower_forward(x)
    assert np.array_equal(p0.view(np.uint32), p1.view(np.uint32)) and np.array_equal(v0.view(np.uint32), v1.view(np.uint32))
    devi.tower_load_torch(torch.from_numpy(blobs[0]).cuda(), net=1)
    with pytest.raises(EngineError):
        devi._ck(devi._L.az_tower_load_weights_dev(devi._h, 0, torch.zeros(10, device="cuda").data_ptr(), 10))
    with pytest.raises(EngineError):
        devi._ck(devi._L.az_tower_load_weights_dev(devi._h, 0, blobs[0].ctypes.data, blobs[0].size))   # a host pointer
    host.close(); devi.close()
