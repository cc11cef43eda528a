"""Drop-in for AlphaZero.processing.state_converter (config key state_converter_path).

The planes come from the engine's feature kernel; index/label conversions are plain Python
because their only job is to build the Python lists the evaluator/search contracts carry.
"""
import importlib

import numpy as np


class StateTensorConverter(object):
    """state -> (1, 17, H, W) int8 (processing/state_converter.py:11-82)."""

    def __init__(self, config, feature_list=None):
        self._config = config
        self._game_env = importlib.import_module(config['env_path'])
        steps = config['history_step'] * config['planes_per_step']
        sizes = {"board_history": steps, "color": 1}
        if feature_list is None:
            feature_list = ["board_history", "color"]
        self.feature_list = feature_list
        self.output_dim = 0
        self._slices = []
        for feat in feature_list:
            feat = feat.lower()
            if feat not in sizes:
                raise ValueError("unknown feature: %s" % feat)
            self._slices.append(slice(0, steps) if feat == "board_history" else slice(steps, steps + 1))
            self.output_dim += sizes[feat]

    def state_to_tensor(self, state):
        planes = state._planes if hasattr(state, "_planes") else state.features(0)
        parts = [planes[s] for s in self._slices]
        return np.concatenate(parts).reshape((1, self.output_dim, state.height, state.width)).astype(np.int8)


class TensorActionConverter(object):
    """probability vector -> [(action, prob)] in index order, PASS (None) last (:85-110)."""

    def __init__(self, config):
        self._wid = config['board_width']
        self._hei = config['board_height']
        self._allow_pass_move = (config['flat_move_output'] - self._wid * self._hei == 1)

    def tensor_to_action(self, tensor):
        w = self._wid
        res = [((i // w, i % w), tensor[i]) for i in range(self._hei * w)]
        if self._allow_pass_move:
            res.append((None, tensor[self._hei * w]))
        return res


def transform_source_index(idx, H, W, transform_id):
    """Cell of the un-transformed board that transformed cell `idx` shows; equally the label the
    reference's ReverseTransformer gives the idx-th policy entry."""
    x, y = idx // W, idx % W
    k = transform_id % 4
    if k == 1:
        x, y = y, W - 1 - x
    elif k == 2:
        x, y = H - 1 - x, W - 1 - y
    elif k == 3:
        x, y = H - 1 - y, x
    if transform_id // 4 == 1:
        y = W - 1 - y
    return x * W + y


class ReverseTransformer(object):
    """Relabels an action list computed on a transformed board, in place (:113-180)."""

    def __init__(self, config):
        self._config = config
        self._wid = config['board_width']
        self._hei = config['board_height']

    def _relabel(self, action_prob, tid):
        H, W = self._hei, self._wid
        for i in range(H * W):
            (x, y), p = action_prob[i]
            s = transform_source_index(x * W + y, H, W, tid)
            action_prob[i] = ((s // W, s % W), p)

    def lr_reflection(self, action_prob):
        self._relabel(action_prob, 4)

    def reverse_nprot90(self, action_prob, transform_id):
        if transform_id not in (0, 1, 2, 3):
            raise NotImplementedError
        if transform_id:
            self._relabel(action_prob, transform_id)

    def reverse_transform(self, action_prob, transform_id):
        self.reverse_nprot90(action_prob, transform_id % 4)
        if transform_id // 4 == 1:
            self.lr_reflection(action_prob)
