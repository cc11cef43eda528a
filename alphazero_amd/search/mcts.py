"""Drop-in for AlphaZero.search.mcts: MCTSearch whose tree lives in the HIP engine.

PUCT select / rule replay / expand / backup / re-root run in the engine's kernels, one leaf per
round exactly as the reference's sequential search (mcts.py:173-296).  The evaluator stays the
caller's `eval(state) -> ([(action, prob)], value)` callable; it is shown the transformed leaf
state and its priors are injected into the device tree.  Random draws are taken from the same
module-level names as the reference (`randint`, `random_variate_dirichlet`,
`weighted_random_choice`) in the same order, so seeding/monkeypatching behaves identically.
"""
import numpy as np
from numpy.random import randint

from ..engine import Engine, game_config_to_engine_kwargs
from .math_helper import random_variate_dirichlet, weighted_random_choice

# Exploration constant of the selection score.  As in the reference (search/mcts.py:9, read at :88 on every call), assigning
# this module global changes the searches that follow: MCTSearch hands the current value to the engine before each search.
c_punt = 5.0


class _LeafView(object):
    """What an evaluator may read of the (transformed) leaf: planes, boards, player, shape."""

    def __init__(self, planes, H, W):
        self._planes = planes
        self.height, self.width = H, W
        self.current_player = 1 if planes[16, 0, 0] else -1

    @property
    def board(self):
        p = self._planes.astype(int)
        return (p[14] - p[15]) * self.current_player

    @property
    def board_history(self):
        p = self._planes.astype(int)
        return [(p[2 * t] - p[2 * t + 1]) * self.current_player for t in range(7)]

    def features(self, transform_id=0):
        assert transform_id == 0
        return self._planes


class _RootChild(object):
    def __init__(self, n, w, p):
        self._visit_cnt, self._total_action_val, self._prior_prob = int(n), w, p
        self.visit_count, self.prior_prob = int(n), p


class _RootView(object):
    """Read-only picture of the root (MCTreeNode subset: _children, _visit_cnt, is_leaf)."""

    def __init__(self, search):
        a, n, w, p, rn = search._eng.root_stats(0, 0)
        self._visit_cnt = self.visit_count = rn
        self._parent = None
        self._children = self.children = {search._i2a(int(a[i])): _RootChild(n[i], w[i], p[i]) for i in range(len(a))}

    def is_leaf(self):
        return self._children == {}

    def is_root(self):
        return True


class MCTSearch(object):
    def __init__(self, evaluator, game_config, max_playout=1600, leaves_per_round=1):
        """leaves_per_round > 1 (device evaluators only) searches several leaves of the tree per round under a virtual
        loss: more simulations per second for a single game, but no longer the reference's visit counts."""
        self._evaluator = evaluator
        self._max_playout = max_playout
        self.d_alpha = game_config['d_alpha']
        self.d_epsilon = game_config['d_epsilon']
        self._transform_types = game_config['transform_types']
        self.enable_transform = not (self._transform_types == 0 or self._transform_types == 1)
        kw = game_config_to_engine_kwargs(game_config)
        self._H, self._W = kw["board_height"], kw["board_width"]
        self._O = game_config['flat_move_output']
        # a device evaluator (NNEvaluator of this package) lets the whole search stay on the GPU
        owner = getattr(evaluator, "__self__", None)
        self._tower = owner if (owner is not None and hasattr(owner, "_tower_blob")) else None
        tower_kw = {}
        if self._tower is not None:
            tower_kw = dict(num_blocks=self._tower.num_blocks, channels=self._tower.channels, tower_dtype=self._tower.dtype)
        if leaves_per_round > 1 and self._tower is None:
            raise ValueError("leaves_per_round > 1 needs a device evaluator (NNEvaluator)")
        # replayed draws follow the reference's sequential order, which several leaves in flight do not have
        # the reference's tree has no size limit and a re-rooted tree keeps the chosen child's subtree (a sharply peaked
        # policy keeps most of it, ply after ply): the trees' limits and the edge pool double whenever one runs out (tree_full_policy 2)
        self._eng_kw = dict(n_slots=1, max_playout=max_playout, rng_replay=leaves_per_round <= 1,
                            leaves_per_round=leaves_per_round, tree_full_policy=2, **kw)
        self._multi = leaves_per_round > 1
        self._K = max(1, int(leaves_per_round))
        self._eng = Engine(**self._eng_kw, **tower_kw)
        self._loaded_version = None
        self._searched = False
        # Arithmetic of N/W/P follows the evaluator's output type, as NumPy's promotion makes it in the reference:
        # float32 priors (the network) keep the search in float32, Python floats / float64 (DummyEvaluator) in
        # float64.  Decided by the first evaluation of a host evaluator; the device tower is always float32.
        self._f64 = False if self._tower is not None else None

    @property
    def _root(self):
        return _RootView(self)

    def _i2a(self, idx):
        return None if idx == self._H * self._W else (idx // self._W, idx % self._W)

    def _a2i(self, action):
        return self._H * self._W if action is None else action[0] * self._W + action[1]

    def _calc_move(self, state, dirichlet=False):
        if bool(getattr(state, "enforce_superko", False)) != bool(self._eng_kw.get("enforce_superko", False)):
            # the search plays by the state's rules (copy() hands enforce_superko on to every playout, env/go.py:250):
            # the first search decides which engine holds the tree
            if self._searched:
                raise ValueError("enforce_superko changed between searches of one tree")
            self._eng.close()
            self._eng_kw["enforce_superko"] = bool(state.enforce_superko)
            tower_kw = {}
            if self._tower is not None:
                tower_kw = dict(num_blocks=self._tower.num_blocks, channels=self._tower.channels, tower_dtype=self._tower.dtype)
            self._eng = Engine(**self._eng_kw, **tower_kw)
            self._loaded_version = None
        self._searched = True
        eng = self._eng
        if eng.cfg.c_punt != float(c_punt):
            eng.set_c_punt(c_punt)
        eng.state_copy(0, state._eng, state._slot, ref_semantics=False)
        _, n_children, _, _, _ = eng.root_stats(0, 0)
        root_is_leaf = len(n_children) == 0
        # draws in the reference's order: transform id for the root, noise, one id per playout
        n_tids = (self._max_playout + (1 if root_is_leaf else 0)) if self.enable_transform else 0
        tids, etas = [], []
        if not self._multi:
            tids = [int(randint(self._transform_types)) for _ in range(1 if (root_is_leaf and self.enable_transform) else 0)]
        if dirichlet and not self._multi:
            n = int(state.legal_mask().sum()) if root_is_leaf else len(n_children)
            etas = list(random_variate_dirichlet(self.d_alpha, n))
        if not self._multi:
            tids += [int(randint(self._transform_types)) for _ in range(n_tids - len(tids))]
            eng.rng_inject(0, tids, etas, [])
        eng.search_begin([0], [0], [int(bool(dirichlet))])
        if self._tower is not None:
            if self._loaded_version != self._tower.version:
                eng.tower_load(self._tower._tower_blob)
                self._loaded_version = self._tower.version
            # the rounds the search certainly needs go to the device in one call; the loop picks up what is left
            # (several leaves per round do not always use a whole round's budget)
            eng.search_run((self._max_playout + self._K - 1) // self._K + (1 if root_is_leaf else 0))
            while eng.step_select() > 0:
                eng.step_finish()
            return
        while eng.step_select() > 0:
            planes, _ = eng.leaf_batch()
            cand, value = self._evaluator(_LeafView(planes[0], self._H, self._W))
            f64 = not isinstance(cand[0][1], np.float32)
            if self._f64 is None:
                self._f64 = f64
                if f64:  # nothing is in the trees yet: restart this search on a float64 engine with the same draws
                    eng.close()
                    eng = self._eng = Engine(**self._eng_kw, search_f64=True, c_punt=c_punt)
                    eng.state_copy(0, state._eng, state._slot, ref_semantics=False)
                    eng.rng_inject(0, tids, etas, [])
                    eng.search_begin([0], [0], [int(bool(dirichlet))])
                    assert eng.step_select() > 0
            elif f64 != self._f64:
                raise TypeError("evaluator changed its output dtype between float32 and float64 within one search tree")
            pri = np.array([[p for _, p in cand]], eng.real)
            val = np.array([value], eng.real)
            eng.step_finish(pri, val)

    def _get_search_probs(self):
        a, n, _, _, _ = self._eng.root_stats(0, 0)
        total = int(n.sum())
        return [(self._i2a(int(a[i])), int(n[i]) / total) for i in range(len(a))]

    def calc_move(self, state, dirichlet=False, prop_exp=True):
        self._calc_move(state, dirichlet)
        if prop_exp:
            return weighted_random_choice(self._get_search_probs())
        a, n, _, _, _ = self._eng.root_stats(0, 0)
        return self._i2a(int(a[int(np.argmax(n))]))  # first maximum, as Python's max()

    def calc_move_with_probs(self, state, dirichlet=False):
        self._calc_move(state, dirichlet)
        probs = self._get_search_probs()
        result = weighted_random_choice(self._get_search_probs())
        return result, probs

    def update_with_move(self, last_move):
        self._eng.advance_tree(0, 0, self._a2i(last_move))
