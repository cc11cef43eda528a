"""GameState objects of the env modules: thin views onto slots of a rules-only HIP engine.

Every rule evaluation (legality, captures, flips, k-in-a-row, scoring, transforms) runs in the
engine's kernels; this file only keeps the Python-visible attributes of the reference classes
(AlphaZero/env/go.py, reversi.py, mnk.py) in sync with the device state.
"""
import numpy as np

from ..engine import Engine, IllegalMove  # noqa: F401  (IllegalMove is re-exported by the env modules)

WHITE = -1
BLACK = +1
EMPTY = 0
PASS_MOVE = None

_POOL_SLOTS = 256


class _SlotPool:
    """Rules-only engines (no trees, no tower) whose slots back GameState objects."""

    def __init__(self, game, H, W, k, superko=False):
        self.key = (game, H, W, k)
        self.superko = superko
        self.engines = []
        self.free = []

    def acquire(self):
        if not self.free:
            game, H, W, k = self.key
            eng = Engine(game, H, W, k, n_slots=_POOL_SLOTS, max_playout=0, transform_types=8 if H == W else 0,
                         tree_edge_cap=1, tree_block_cap=2, max_depth=2, enforce_superko=self.superko)
            self.engines.append(eng)
            self.free.extend((eng, s) for s in reversed(range(_POOL_SLOTS)))
        return self.free.pop()

    def release(self, eng, slot):
        self.free.append((eng, slot))


_pools = {}


def _pool(game, H, W, k, superko=False):
    key = (game, H, W, k, bool(superko))
    if key not in _pools:
        _pools[key] = _SlotPool(game, H, W, k, bool(superko))
    return _pools[key]


class BaseGameState(object):
    _game = None

    def _setup(self, H, W, k=0, history_length=8, superko=False):
        if history_length != 8:
            raise ValueError("the engine keeps history_length = 8 boards (config history_step)")
        self.height, self.width = H, W
        self._k = k
        self._pool = _pool(self._game, H, W, k, superko)
        self._eng, self._slot = self._pool.acquire()
        self._eng.reset([self._slot])
        self._blob = None
        self.history = []
        self.history_length = history_length

    def __del__(self):
        try:
            self._pool.release(self._eng, self._slot)
        except Exception:
            pass

    # ---- device state access -------------------------------------------------
    def _b(self):
        if self._blob is None:
            self._blob = self._eng.state_get(self._slot)
        return self._blob

    def _dirty(self):
        self._blob = None

    def _a2i(self, action):
        if action is PASS_MOVE:
            return self.height * self.width
        x, y = action
        if not (0 <= x < self.height and 0 <= y < self.width):
            return -1
        return x * self.width + y

    def _i2a(self, idx):
        return PASS_MOVE if idx == self.height * self.width else (idx // self.width, idx % self.width)

    @property
    def board(self):
        b = self._b()
        return np.frombuffer(b.board, dtype=np.int8, count=self.height * self.width).reshape(self.height, self.width).astype(int)

    @property
    def board_history(self):
        b = self._b()
        n = self.height * self.width
        return [np.frombuffer(b.history[t], dtype=np.int8, count=n).reshape(self.height, self.width).astype(int) for t in range(7)]

    @property
    def current_player(self):
        return int(self._b().current_player)

    @current_player.setter
    def current_player(self, v):
        b = self._b()
        b.current_player = int(v)
        self._eng.state_set(self._slot, b)
        self._dirty()

    @property
    def is_end_of_game(self):
        return bool(self._b().is_end_of_game)

    @property
    def turns(self):
        return int(self._b().turns)

    def get_current_player(self):
        return self.current_player

    def _on_board(self, position):
        (x, y) = position
        return 0 <= x < self.height and 0 <= y < self.width

    # ---- rules -------------------------------------------------------------------
    def _copy_into(self, other):
        other._eng.state_copy(other._slot, self._eng, self._slot, ref_semantics=True)
        other._dirty()
        other.history = list(self.history)
        return other

    def legal_mask(self):
        """uint8[O] in tensor-index order (PASS last where the game has one)."""
        return self._eng.legal_mask([self._slot])[0]

    def is_legal(self, action):
        idx = self._a2i(action)
        if idx < 0:
            return False
        m = self.legal_mask()
        return idx < len(m) and bool(m[idx])

    def do_move(self, action, color=None):
        idx = self._a2i(action)
        if idx < 0:
            raise IllegalMove(str(action))
        status, ended = self._eng.do_move([self._slot], [idx], None if color is None else [int(color)])
        if status[0] == -7:
            raise RuntimeError("the game is longer than the engine's superko record")
        if status[0] != 0:
            raise IllegalMove(str(action))
        self._dirty()
        self.history.append(action)
        return bool(ended[0])

    def get_winner(self):
        w = int(self._eng.winner([self._slot])[0])
        return None if w == 2 else w

    def transform(self, transform_id):
        assert self.height == self.width or transform_id == 0
        self._eng.transform(self._slot, int(transform_id))
        self._dirty()

    def features(self, transform_id=0):
        """int8 (17, H, W) planes of copy().transform(id): StateTensorConverter's output."""
        return self._eng.features([self._slot], [int(transform_id)])[0]
