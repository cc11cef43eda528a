"""Evaluator without a network: every action gets the same prior and the value is 0.

Takes the place of AlphaZero/evaluator/dummy_eval.py:18-28.  The priors are Python/NumPy float64 and the value is the
int 0, exactly as there, so a search fed by it runs its N/W/P arithmetic in double precision (the drop-in MCTSearch
notices and uses the float64 engine).
"""
import functools
import os

import numpy as np
import yaml

from ..processing.state_converter import TensorActionConverter


@functools.lru_cache(maxsize=None)
def default_game_config():
    """The game named in config/game.yaml, read the first time an evaluator is built without a config of its own."""
    cfg_dir = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir, "config"))
    with open(os.path.join(cfg_dir, "game.yaml")) as fh:
        chosen = yaml.safe_load(fh)["game"]
    with open(os.path.join(cfg_dir, "%s.yaml" % chosen)) as fh:
        return yaml.safe_load(fh)


class DummyEvaluator:
    def __init__(self, config=None):
        cfg = config if config is not None else default_game_config()
        self._n_actions = int(cfg["flat_move_output"])
        self._label = TensorActionConverter(cfg).tensor_to_action

    # the reference's evaluator contract: checkpoints are no-ops here
    def load(self, save_dir):
        return None

    def save(self, save_dir):
        return None

    def eval(self, state):
        uniform = np.full((self._n_actions,), 1 / self._n_actions)
        return self._label(uniform), 0


def __getattr__(name):
    """`game_config` and `game_selection`, module attributes in the reference's file, resolved on first use."""
    if name == "game_config":
        return default_game_config()
    if name == "game_selection":
        return default_game_config()["name"]
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
