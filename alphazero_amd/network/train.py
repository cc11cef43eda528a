"""Training-side `Network` with the reference's interface (network/main.py:13-178) on PyTorch-ROCm:
`update(data) -> loss`, `response(data) -> (R_p, R_v)`, `evaluate(data) -> (loss, acc, mse)`,
`get_global_step()`, `save(filename)`, `load(filename)`.

Loss (network/model.py:90-99): mean softmax cross-entropy against the search distribution
+ MSE_scaling * mean((v_hat - z)^2 / 4) + l2 * sum over all trainable variables of sum(w^2)/2
(tf.contrib l2_regularizer = scale * tf.nn.l2_loss).  Optimiser: momentum SGD (accum = m*accum + g;
w -= lr*accum, tf.train.MomentumOptimizer) with the piecewise-constant schedule of reinforce.yaml
(tf.train.piecewise_constant(step, boundaries[1:], values): values[i] holds while step <= boundary i).
Batch norm in training mode uses batch statistics and updates the moving averages with `batch_decay`.

Parity note: TensorFlow is absent and the reference has no test for this path, so the arithmetic is
checked against an independent NumPy restatement of the formulas above, not against TF (unpinned).
"""
import os

import numpy as np
import torch
import torch.nn.functional as F
import yaml

from .model import AZNet, CenterBN

_DEFAULT_TRAIN_CONFIG = {
    "num_blocks": 19, "batch_decay": 0.9,
    "learning_rate": {0: 1e-2, 20000: 1e-2, 40000: 1e-3, 60000: 1e-4, 70000: 1e-4},
    "momentum": 0.9, "l2": 1e-4, "MSE_scaling": 1.0, "save_dir": "model",
}


def piecewise_lr(step, scheme):
    boundaries = sorted(int(k) for k in scheme.keys())
    values = [float(scheme[k]) for k in sorted(scheme.keys(), key=int)]
    for b, v in zip(boundaries[1:], values):
        if step <= b:
            return v
    return values[-1]


def az_loss(logits, v_hat, p, z, params, l2, mse_scaling):
    v_loss = torch.mean((v_hat - z) ** 2 / 4.0)
    p_loss = torch.mean(-(p * F.log_softmax(logits, dim=1)).sum(dim=1))
    r_loss = l2 * sum((w ** 2).sum() / 2.0 for w in params)
    return p_loss + v_loss * mse_scaling + r_loss


class Network:
    def __init__(self, game_config, num_gpu=1, train_config=None, load_pretrained=False, device=None, channels=256):
        if train_config is None:
            cfg = dict(_DEFAULT_TRAIN_CONFIG)
        elif isinstance(train_config, dict):
            cfg = dict(train_config)
        else:
            with open(train_config) as fh:
                cfg = yaml.safe_load(fh)
        self._train_config = cfg
        self._game_config = game_config
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        self.net = AZNet(game_config["board_height"], game_config["board_width"], game_config["flat_move_output"],
                         cfg["num_blocks"], channels).to(self.device)
        for m in self.net.modules():
            if isinstance(m, CenterBN):
                m.decay = float(cfg["batch_decay"])
        self.global_step = 0
        self.opt = torch.optim.SGD(self.net.parameters(), lr=piecewise_lr(0, cfg["learning_rate"]),
                                   momentum=float(cfg["momentum"]), dampening=0.0, nesterov=False)
        if load_pretrained:
            # network/main.py:92-95 restores tf.train.latest_checkpoint(save_dir): the newest '<prefix>-<step>' of save()
            path = latest_checkpoint(cfg["save_dir"])
            if path is None:
                raise FileNotFoundError("no '<prefix>-<step>' checkpoint in %r" % cfg["save_dir"])
            self.load(path)

    def _t(self, a):
        return torch.as_tensor(np.asarray(a), dtype=torch.float32, device=self.device)

    def _loss(self, data):
        state, action, result = data
        _, v_hat, logits = self.net(self._t(state))
        loss = az_loss(logits, v_hat, self._t(action), self._t(result), list(self.net.parameters()),
                       float(self._train_config["l2"]), float(self._train_config["MSE_scaling"]))
        return loss, logits, v_hat

    def update(self, data):
        self.net.train()
        for g in self.opt.param_groups:
            g["lr"] = piecewise_lr(self.global_step, self._train_config["learning_rate"])
        self.opt.zero_grad(set_to_none=True)
        loss, _, _ = self._loss(data)
        loss.backward()
        self.opt.step()
        self.global_step += 1
        return float(loss.detach())

    def response(self, data):
        self.net.eval()
        with torch.no_grad():
            p, v, _ = self.net(self._t(data[0]))
        return p.cpu().numpy(), v.cpu().numpy()

    def evaluate(self, data):
        self.net.eval()
        with torch.no_grad():
            loss, logits, v_hat = self._loss(data)
        R_p, R_v = F.softmax(logits, dim=1).cpu().numpy(), v_hat.cpu().numpy()
        acc = np.mean((np.argmax(R_p, axis=1) == np.argmax(data[1], axis=1)).astype(np.float32))
        mse = np.mean(np.square(np.asarray(data[2]) - R_v)) / 4.
        return float(loss), acc, mse

    def get_global_step(self):
        return self.global_step

    def save(self, filename):
        """filename is a prefix; the step is appended as TF's Saver does (main.py:160-169)."""
        path = "%s-%d" % (filename, self.global_step)
        torch.save({"model": self.net.state_dict(), "opt": self.opt.state_dict(), "global_step": self.global_step}, path)
        return path

    def load(self, filename):
        """A file of save(), or a bare state dict as NNEvaluator.save writes it (weights only: step and optimizer
        state are then left as they are)."""
        ck = torch.load(filename, weights_only=True, map_location=self.device)
        if "model" not in ck:
            self.net.load_state_dict(ck)
            return
        self.net.load_state_dict(ck["model"])
        self.opt.load_state_dict(ck["opt"])
        self.global_step = int(ck["global_step"])


def latest_checkpoint(save_dir):
    """Path of the '<prefix>-<step>' file with the largest step in save_dir (what Network.save writes), or None."""
    import re
    best, best_step = None, -1
    if os.path.isdir(save_dir):
        for name in os.listdir(save_dir):
            m = re.fullmatch(r".*-(\d+)", name)
            if m and int(m.group(1)) > best_step and os.path.isfile(os.path.join(save_dir, name)):
                best, best_step = os.path.join(save_dir, name), int(m.group(1))
    return best
