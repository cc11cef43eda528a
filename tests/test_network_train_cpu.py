"""Training step and TF-name weight interchange (network/main.py, network/model.py:90-99).  TensorFlow is
absent, so these are checked against independent NumPy restatements of the formulas (parity unpinned)."""
import numpy as np
import torch

from alphazero_amd.network.model import random_net
from alphazero_amd.network.tf_names import from_tf_named, to_tf_named
from alphazero_amd.network.train import Network, piecewise_lr

CFG = {"board_height": 3, "board_width": 3, "flat_move_output": 9}
TRAIN = {"num_blocks": 1, "batch_decay": 0.9, "learning_rate": {0: 1e-2, 20: 1e-2, 40: 1e-3, 60: 1e-4, 70: 1e-4},
         "momentum": 0.9, "l2": 1e-4, "MSE_scaling": 1.0, "save_dir": "model"}


def test_piecewise_schedule_follows_tf_semantics():
    s = TRAIN["learning_rate"]
    # tf.train.piecewise_constant(step, [20, 40, 60, 70], [1e-2, 1e-2, 1e-3, 1e-4, 1e-4])
    assert [piecewise_lr(t, s) for t in (0, 20, 21, 40, 41, 60, 61, 70, 71, 10 ** 6)] == \
        [1e-2, 1e-2, 1e-2, 1e-2, 1e-3, 1e-3, 1e-4, 1e-4, 1e-4, 1e-4]


def test_tf_named_layouts_and_round_trip():
    net = random_net(3, 3, 9, 2, 128, seed=4, perturb_bn=True)
    d = to_tf_named(net)
    assert d["model/W0"].shape == (3, 3, 17, 128) and d["model/resblock_1/W2"].shape == (3, 3, 128, 128)
    assert d["model/policy_head/W0"].shape == (1, 1, 128, 2) and d["model/policy_head/linear/W"].shape == (18, 9)
    assert d["model/value_head/F1/W"].shape == (9, 256) and d["model/value_head/F2/W"].shape == (256, 1)
    assert d["model/resblock_0/B1/BatchNorm/moving_variance"].shape == (128,)
    assert len(d) == 4 + 2 * (2 + 6) + 6 + 8 and not any("gamma" in k for k in d)
    # HWIO really is the layout a SAME cross-correlation uses: out[o,y,x] = sum_{i,dy,dx} in[i,y+dy-1,x+dx-1] * W[dy,dx,i,o]
    x = torch.randn(1, 17, 3, 3)
    ref = net.stem(x)[0].detach().numpy()
    xp = np.pad(x[0].numpy(), ((0, 0), (1, 1), (1, 1)))
    W = d["model/W0"]
    got = np.zeros_like(ref)
    for dy in range(3):
        for dx in range(3):
            got += np.einsum("iyx,io->oyx", xp[:, dy:dy + 3, dx:dx + 3], W[dy, dx])
    assert np.abs(got - ref).max() < 1e-5
    other = random_net(3, 3, 9, 2, 128, seed=9)
    from_tf_named(other, d)
    for (k, a), (_, b) in zip(net.state_dict().items(), other.state_dict().items()):
        assert torch.equal(a, b), k


def test_update_is_momentum_sgd_on_the_reference_loss():
    torch.manual_seed(0)
    nw = Network(CFG, train_config=TRAIN, device="cpu", channels=128)
    rng = np.random.RandomState(1)
    B = 6
    state = (rng.rand(B, 17, 3, 3) < 0.3).astype(np.float64)
    action = rng.dirichlet(np.ones(9), B)
    result = rng.choice([-1.0, 1.0], B)
    before = {k: v.detach().clone() for k, v in nw.net.named_parameters()}
    # independent evaluation of the loss in training mode (batch statistics)
    nw.net.train()
    _, v_hat, logits = nw.net(torch.tensor(state, dtype=torch.float32))
    lg, vh = logits.detach().numpy().astype(np.float64), v_hat.detach().numpy().astype(np.float64)
    lse = np.log(np.exp(lg - lg.max(1, keepdims=True)).sum(1)) + lg.max(1)
    p_loss = np.mean(-(action * (lg - lse[:, None])).sum(1))
    v_loss = np.mean((vh - result) ** 2 / 4.0)
    r_loss = 1e-4 * sum(float((w.double() ** 2).sum()) / 2.0 for w in before.values())
    # restore BN moving stats touched by the probe forward, then take the real step
    nw2 = Network(CFG, train_config=TRAIN, device="cpu", channels=128)
    nw2.net.load_state_dict({k: v for k, v in nw.net.state_dict().items()})
    for m in nw2.net.modules():
        if hasattr(m, "running_mean"):
            m.running_mean.zero_(); m.running_var.fill_(1.0)
    loss = nw2.update((state, action, result))
    assert abs(loss - (p_loss + v_loss + r_loss)) < 1e-5
    assert nw2.get_global_step() == 1
    # first momentum step: w1 = w0 - lr * g
    nw.net.zero_grad()
    _, v_hat, logits = nw.net(torch.tensor(state, dtype=torch.float32))
    from alphazero_amd.network.train import az_loss
    az_loss(logits, v_hat, torch.tensor(action, dtype=torch.float32), torch.tensor(result, dtype=torch.float32),
            list(nw.net.parameters()), 1e-4, 1.0).backward()
    for (k, p), (_, q) in zip(nw.net.named_parameters(), nw2.net.named_parameters()):
        assert torch.allclose(q.detach(), before[k] - 1e-2 * p.grad, atol=1e-6), k
    # moving averages moved towards the batch statistics with decay 0.9
    assert float(nw2.net.stem_bn.running_var.mean()) != 1.0
    # second step uses the accumulated momentum: accum = 0.9*g1 + g2
    loss2 = nw2.update((state, action, result))
    assert np.isfinite(loss2) and nw2.get_global_step() == 2


def test_response_evaluate_save_load(tmp_path):
    nw = Network(CFG, train_config=TRAIN, device="cpu", channels=128)
    rng = np.random.RandomState(2)
    state = (rng.rand(4, 17, 3, 3) < 0.3).astype(np.float64)
    action = np.eye(9)[rng.randint(0, 9, 4)]
    result = rng.choice([-1.0, 1.0], 4)
    p, v = nw.response((state,))
    assert p.shape == (4, 9) and v.shape == (4,) and np.allclose(p.sum(1), 1, atol=1e-5)
    loss, acc, mse = nw.evaluate((state, action, result))
    assert np.isfinite(loss) and 0 <= acc <= 1 and abs(mse - np.mean((result - v) ** 2) / 4) < 1e-6
    nw.update((state, action, result))
    path = nw.save(str(tmp_path / "ckpt"))
    assert path.endswith("ckpt-1")
    nw2 = Network(CFG, train_config=TRAIN, device="cpu", channels=128)
    nw2.load(path)
    p2, v2 = nw2.response((state,))
    p1, v1 = nw.response((state,))
    assert np.array_equal(p1, p2) and np.array_equal(v1, v2) and nw2.get_global_step() == 1


def test_load_pretrained_takes_the_newest_step(tmp_path):
    """load_pretrained restores the latest '<prefix>-<step>' of save_dir, as tf.train.latest_checkpoint does for the
    reference (network/main.py:92-95)."""
    from alphazero_amd.network.train import latest_checkpoint
    cfg = dict(TRAIN, save_dir=str(tmp_path))
    assert latest_checkpoint(str(tmp_path)) is None
    nw = Network(CFG, train_config=cfg, device="cpu", channels=128)
    rng = np.random.RandomState(3)
    data = ((rng.rand(4, 17, 3, 3) < 0.3).astype(np.float64), np.eye(9)[rng.randint(0, 9, 4)], rng.choice([-1.0, 1.0], 4))
    for _ in range(3):
        nw.update(data)
        nw.save(str(tmp_path / "ckpt"))          # ckpt-1, ckpt-2, ckpt-3
    (tmp_path / "notes.txt").write_text("x")
    assert latest_checkpoint(str(tmp_path)).endswith("ckpt-3")
    nw2 = Network(CFG, train_config=cfg, device="cpu", channels=128, load_pretrained=True)
    assert nw2.get_global_step() == 3
    assert np.array_equal(nw2.response(data)[0], nw.response(data)[0])
    import pytest
    with pytest.raises(FileNotFoundError):
        Network(CFG, train_config=dict(TRAIN, save_dir=str(tmp_path / "none")), device="cpu", channels=128, load_pretrained=True)
