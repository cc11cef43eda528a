"""Drop-in for AlphaZero.evaluator.dummy_eval: uniform priors, value 0 (float64, as the reference)."""
import os

import numpy as np
import yaml

from ..processing.state_converter import TensorActionConverter

_cfg_dir = os.path.join(os.path.dirname(__file__), '..', 'config')
with open(os.path.join(_cfg_dir, 'game.yaml')) as f:
    game_selection = yaml.safe_load(f)['game']
with open(os.path.join(_cfg_dir, game_selection + '.yaml')) as c:
    game_config = yaml.safe_load(c)


class DummyEvaluator:
    def __init__(self, config=None):
        self._config = config or game_config
        self._t2a = TensorActionConverter(self._config)

    def load(self, save_dir):
        pass

    def save(self, save_dir):
        pass

    def eval(self, state):
        dims = self._config['flat_move_output']
        return self._t2a.tensor_to_action(np.full((dims,), 1 / dims)), 0
