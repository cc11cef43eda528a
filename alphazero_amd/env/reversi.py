"""Drop-in for AlphaZero.env.reversi (config key env_path)."""
from ._base import BLACK, EMPTY, PASS_MOVE, WHITE, BaseGameState, IllegalMove  # noqa: F401


class GameState(BaseGameState):
    """Reversi rules on the engine (env/reversi.py:11-228)."""
    _game = "reversi"

    def __init__(self, size=8, history_length=8):
        self.size = size
        self._setup(size, size, 0, history_length)

    def copy(self):
        return self._copy_into(GameState(self.size))

    @property
    def stones_played(self):
        return int(self._b().stones_played)

    def get_legal_moves(self):
        m = self.legal_mask()
        return [(i // self.width, i % self.width) for i in range(self.height * self.width) if m[i]]
