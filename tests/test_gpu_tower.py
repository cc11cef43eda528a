"""GPU numerics: HIP tower vs the fp32 CPU restatement of network/model.py (oracle/tower_ref.py).
Tolerances are the north star's: 1e-5 abs for fp32, 1e-2 abs for bf16 (on softmax priors and the
tanh value)."""
import numpy as np
import pytest
import torch

from oracle import tower_ref

pytestmark = pytest.mark.gpu

TOL = {"fp32": 1e-5, "bf16": 1e-2}


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
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
