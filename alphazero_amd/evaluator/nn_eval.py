"""Evaluator object with the reference's contract (evaluator/nn_eval_parallel.py:30-130):
`eval(state) -> ([(action, prob)] in index order, value)`, `load(path)`, `save(path)`, `.rwlock`,
context manager.  The network runs in the HIP tower of an engine owned by this object; there is
no listener process and no queue - batching happens inside the engine across game slots."""
import threading

import numpy as np
import torch

from ..engine import Engine, game_config_to_engine_kwargs
from ..network.model import AZNet, fold_to_blob, random_net
from ..processing.state_converter import StateTensorConverter, TensorActionConverter


class RWLock:
    """Readers-writer lock with the reference's method names (train/parallel/util.py:22-46)."""

    def __init__(self):
        self._cond = threading.Condition()
        self._readers = 0
        self._writer = False

    def r_acquire(self):
        with self._cond:
            while self._writer:
                self._cond.wait()
            self._readers += 1

    def r_release(self):
        with self._cond:
            self._readers -= 1
            self._cond.notify_all()

    def w_acquire(self):
        with self._cond:
            while self._writer or self._readers:
                self._cond.wait()
            self._writer = True

    def w_release(self):
        with self._cond:
            self._writer = False
            self._cond.notify_all()


class NNEvaluator:
    def __init__(self, game_config, num_blocks=19, channels=256, dtype="bf16", max_batch_size=32, seed=0, net=None, device=0):
        self.game_config = game_config
        self.num_blocks, self.channels, self.dtype = num_blocks, channels, dtype
        kw = game_config_to_engine_kwargs(game_config)
        self._H, self._W, self._O = kw["board_height"], kw["board_width"], game_config['flat_move_output']
        self.net = net if net is not None else random_net(self._H, self._W, self._O, num_blocks, channels, seed=seed)
        self.rwlock = RWLock()
        self._eng = Engine(n_slots=max_batch_size, num_blocks=num_blocks, channels=channels, tower_dtype=dtype, device=device, **kw)
        self._conv = StateTensorConverter(game_config)
        self._t2a = TensorActionConverter(game_config)
        self.version = 0
        self._refold()

    def _refold(self):
        self._tower_blob = fold_to_blob(self.net)
        self._eng.tower_load(self._tower_blob)
        self.version += 1

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        return False

    def eval(self, state):
        planes = self._conv.state_to_tensor(state)
        p, v = self._eng.tower_forward(planes)
        return self._t2a.tensor_to_action(p[0]), v[0]

    def eval_batch(self, planes):
        """planes int8 (B,17,H,W) -> (priors float32 (B,O), values float32 (B,)): Network.response."""
        return self._eng.tower_forward(planes)

    def load(self, filename):
        self.rwlock.w_acquire()
        try:
            # either a bare state dict (save() below) or the optimizer side's checkpoint (network/train.py Network.save:
            # {'model', 'opt', 'global_step'} in '<prefix>-<step>'), which is how weights reach the evaluators in the
            # reference: net.save('.../ckpt') then nn_eval.load('.../ckpt-<step>') (train/parallel/evaluator.py:109)
            ck = torch.load(filename, weights_only=True, map_location="cpu")
            self.net.load_state_dict(ck["model"] if "model" in ck else ck)
            self._refold()
        finally:
            self.rwlock.w_release()

    def save(self, filename):
        # a reader of the weights: a load() in another thread (the writer) may not swap them half-way through the file
        self.rwlock.r_acquire()
        try:
            torch.save(self.net.state_dict(), filename)
        finally:
            self.rwlock.r_release()

    def set_weights(self, state_dict):
        self.rwlock.w_acquire()
        try:
            self.net.load_state_dict(state_dict)
            self._refold()
        finally:
            self.rwlock.w_release()
