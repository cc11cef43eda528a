"""Drop-in for AlphaZero.env.go (config key env_path): same constants, GameState and IllegalMove."""
from ._base import BLACK, EMPTY, PASS_MOVE, WHITE, BaseGameState, IllegalMove  # noqa: F401


class GameState(BaseGameState):
    """Go rules on the engine (env/go.py:11-594), with positional superko on request (env/go.py:303-330: what the
    reference's GTP front end plays with; self-play leaves it off)."""
    _game = "go"

    def __init__(self, size=19, komi=7.5, enforce_superko=False, history_length=8):
        if komi != 7.5:
            raise NotImplementedError("the engine scores with komi 7.5")
        self.size = size
        self.komi = komi
        self._superko = bool(enforce_superko)
        self.handicaps = []
        self._setup(size, size, 0, history_length, superko=self._superko)

    @property
    def enforce_superko(self):
        return self._superko

    @enforce_superko.setter
    def enforce_superko(self, on):
        if bool(on) != self._superko:
            raise NotImplementedError("enforce_superko is fixed when the state is created (it selects the engine that keeps "
                                      "the position history)")

    def copy(self):
        return self._copy_into(GameState(self.size, self.komi, self._superko))

    @property
    def ko(self):
        k = int(self._b().ko)
        return None if k < 0 else (k // self.width, k % self.width)

    @property
    def num_black_prisoners(self):
        return int(self._b().num_black_prisoners)

    @property
    def num_white_prisoners(self):
        return int(self._b().num_white_prisoners)

    @property
    def passes_black(self):
        return int(self._b().passes_black)

    @property
    def passes_white(self):
        return int(self._b().passes_white)

    def is_suicide(self, action):
        (x, y) = action
        return self.board[x][y] == EMPTY and action != self.ko and not self.is_legal(action)

    def get_legal_moves(self, include_eyes=True):
        m = self.legal_mask()
        return [(i // self.width, i % self.width) for i in range(self.height * self.width) if m[i]]
