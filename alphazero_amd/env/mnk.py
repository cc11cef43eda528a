"""Drop-in for AlphaZero.env.mnk (config key env_path).  Board size and k come from the module-level
`game_config` dict, as in the reference (env/mnk.py:12-13,22-24), initialised from config/mnk.yaml."""
import os

import yaml

from ._base import BLACK, EMPTY, PASS_MOVE, WHITE, BaseGameState, IllegalMove  # noqa: F401

with open(os.path.join(os.path.dirname(__file__), '..', 'config', 'mnk.yaml')) as f:
    game_config = yaml.safe_load(f)


class GameState(BaseGameState):
    """m,n,k rules on the engine (env/mnk.py:16-211)."""
    _game = "mnk"

    def __init__(self, history_length=8):
        self.k = game_config['k']
        self._setup(game_config['board_height'], game_config['board_width'], self.k, history_length)

    def copy(self):
        saved = dict(game_config)
        game_config.update(board_height=self.height, board_width=self.width, k=self.k)
        try:
            other = GameState()
        finally:
            game_config.update(saved)
        return self._copy_into(other)

    @property
    def winner(self):
        b = self._b()
        return int(b.winner) if b.winner_set else None

    def is_legal(self, action):
        (x, y) = action  # PASS is not an action of this game: unpacking None raises, as in the reference
        return super().is_legal((x, y))

    def get_legal_moves(self):
        m = self.legal_mask()
        return [(i // self.width, i % self.width) for i in range(self.height * self.width) if m[i]]
