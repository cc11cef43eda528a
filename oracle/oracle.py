"""TEST INFRASTRUCTURE - NOT PRODUCT CODE.

ctypes binding of oracle/liboracle.so (the scalar C restatement of the
reference's rules / features / search / game loop).  Imported only by tests/,
`__graft_entry__.smoke()` and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_SO64 = os.path.join(_HERE, "liboracle64.so")   # same source built with -DOZ_REAL=double

GO, REVERSI, MNK = 0, 1, 2
GAME_IDS = {"go": GO, "reversi": REVERSI, "mnk": MNK}
MAXC, MAXO, NHIST, NPLANES = 361, 362, 7, 17
NOMOVE = -2
SK_MAX = 1024


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


class OzState(C.Structure):
    _fields_ = [
        ("game", C.c_int), ("H", C.c_int), ("W", C.c_int), ("k", C.c_int), ("O", C.c_int),
        ("board", C.c_int8 * MAXC),
        ("hist", (C.c_int8 * MAXC) * NHIST),
        ("cur", C.c_int), ("ko", C.c_int), ("turns", C.c_int), ("nhist", C.c_int),
        ("last1", C.c_int), ("last2", C.c_int), ("is_end", C.c_int),
        ("winner", C.c_int), ("winner_set", C.c_int),
        ("passes_b", C.c_int), ("passes_w", C.c_int),
        ("pris_b", C.c_int), ("pris_w", C.c_int),
        ("stones_played", C.c_int),
        ("komi", C.c_double),
        ("superko", C.c_int), ("nprev", C.c_int), ("cur_hash", C.c_uint64),
        ("sk_moves", C.c_int16 * SK_MAX), ("sk_hash", C.c_uint64 * SK_MAX),
    ]


class OzRng(C.Structure):
    _fields_ = [
        ("tids", C.POINTER(C.c_int32)), ("n_tids", C.c_int64), ("i_tids", C.c_int64),
        ("etas", C.POINTER(C.c_double)), ("n_etas", C.c_int64), ("i_etas", C.c_int64),
        ("unis", C.POINTER(C.c_double)), ("n_unis", C.c_int64), ("i_unis", C.c_int64),
        ("seed", C.c_uint64), ("underflow", C.c_int),
    ]


class OzGameCfg(C.Structure):
    _fields_ = [
        ("game", C.c_int), ("H", C.c_int), ("W", C.c_int), ("k", C.c_int),
        ("transform_types", C.c_int),
        ("max_playout", C.c_int), ("dirichlet_before", C.c_int), ("max_turn", C.c_int),
        ("d_alpha", C.c_double), ("d_epsilon", C.c_double),
    ]


class OzTrajectory(C.Structure):
    _fields_ = [
        ("n_plies", C.c_int), ("winner", C.c_int),
        ("moves", C.POINTER(C.c_int16)), ("visits", C.POINTER(C.c_int32)),
        ("n_evals", C.c_int64),
    ]


EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int8), C.c_int, C.c_int, C.c_int,
                      C.POINTER(C.c_float), C.POINTER(C.c_float))
EVAL_FN64 = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int8), C.c_int, C.c_int, C.c_int,
                        C.POINTER(C.c_double), C.POINTER(C.c_double))

_libs = {}


def lib(real64=False):
    """The oracle library; real64 selects the build whose search statistics are float64."""
    if real64 not in _libs:
        so = _SO64 if real64 else _SO
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        SP = C.POINTER(OzState)
        L.oz_init.argtypes = [SP, C.c_int, C.c_int, C.c_int, C.c_int]
        L.oz_copy.argtypes = [SP, SP]
        L.oz_is_legal.argtypes = [SP, C.c_int]
        L.oz_legal_mask.argtypes = [SP, C.c_void_p]
        L.oz_do_move.argtypes = [SP, C.c_int]
        L.oz_do_move_color.argtypes = [SP, C.c_int, C.c_int]
        L.oz_winner.argtypes = [SP]
        L.oz_transform_board.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.oz_features.argtypes = [SP, C.c_int, C.c_void_p]
        L.oz_reverse_label.argtypes = [C.c_int] * 4
        L.oz_tree_new.restype = C.c_void_p
        L.oz_tree_new.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double]
        L.oz_tree_free.argtypes = [C.c_void_p]
        L.oz_tree_search.argtypes = [C.c_void_p, SP, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(OzRng)]
        L.oz_tree_root_stats.argtypes = [C.c_void_p] * 5
        L.oz_tree_root_visits.argtypes = [C.c_void_p]
        L.oz_tree_choose.argtypes = [C.c_void_p, C.c_double]
        L.oz_tree_argmax.argtypes = [C.c_void_p]
        L.oz_tree_update_with_move.argtypes = [C.c_void_p, C.c_int]
        L.oz_tree_num_nodes.restype = C.c_int64
        L.oz_tree_num_nodes.argtypes = [C.c_void_p]
        L.oz_tree_playout.argtypes = [C.c_void_p, SP, C.c_void_p, C.c_void_p, C.POINTER(OzRng)]
        L.oz_tree_count_expanded.argtypes = [C.c_void_p]
        L.oz_play_game.argtypes = [C.POINTER(OzGameCfg), C.c_void_p, C.c_void_p, C.POINTER(OzRng),
                                   C.POINTER(OzTrajectory), C.c_int]
        L.oz_play_game_ex.argtypes = [C.POINTER(OzGameCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                      C.POINTER(OzRng), C.POINTER(OzTrajectory), C.c_int]
        L.oz_tree_search_k.argtypes = [C.c_void_p, SP, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                       C.POINTER(OzRng)]
        L.oz_set_c_punt.argtypes = [C.c_double]
        L.oz_set_c_punt.restype = None
        L.oz_gamma_small_alpha.restype = C.c_double
        L.oz_gamma_small_alpha.argtypes = [C.c_double, C.POINTER(C.c_uint64)]
        _libs[real64] = L
    return _libs[real64]


def set_c_punt(c_punt, real64=False):
    """Assign the module global c_punt (search/mcts.py:9) of the oracle build in question."""
    lib(real64).oz_set_c_punt(float(c_punt))


def fn_ptr(name, real64=False):
    return C.cast(getattr(lib(real64), name), C.c_void_p)


class State:
    """Thin object wrapper: actions are flat indices (x*W+y, PASS = H*W)."""

    def __init__(self, game, H, W, k=0, enforce_superko=False):
        self.s = OzState()
        lib().oz_init(C.byref(self.s), GAME_IDS.get(game, game), H, W, k)
        if enforce_superko:
            self.s.superko = 1

    @property
    def H(self): return self.s.H
    @property
    def W(self): return self.s.W
    @property
    def O(self): return self.s.O
    @property
    def ncell(self): return self.s.H * self.s.W

    def copy(self):
        o = State.__new__(State)
        o.s = OzState()
        lib().oz_copy(C.byref(o.s), C.byref(self.s))
        return o

    def clone_exact(self):
        o = State.__new__(State)
        o.s = OzState()
        C.memmove(C.byref(o.s), C.byref(self.s), C.sizeof(OzState))
        return o

    def board(self):
        return np.frombuffer(self.s.board, dtype=np.int8, count=self.ncell).reshape(self.H, self.W).copy()

    def history_boards(self):
        a = np.frombuffer(self.s.hist, dtype=np.int8).reshape(NHIST, MAXC)[:, :self.ncell]
        return a.reshape(NHIST, self.H, self.W).copy()

    def is_legal(self, a): return bool(lib().oz_is_legal(C.byref(self.s), int(a)))

    def legal_mask(self):
        m = np.zeros(self.O, dtype=np.uint8)
        lib().oz_legal_mask(C.byref(self.s), m.ctypes.data)
        return m

    def do_move(self, a, color=0):
        return lib().oz_do_move_color(C.byref(self.s), int(a), int(color))

    def winner(self): return lib().oz_winner(C.byref(self.s))

    def features(self, tid=0):
        out = np.zeros((NPLANES, self.H, self.W), dtype=np.int8)
        lib().oz_features(C.byref(self.s), int(tid), out.ctypes.data)
        return out


class Rng:
    """Replay streams (numpy arrays) or an internal generator when None."""

    def __init__(self, tids=None, etas=None, unis=None, seed=0):
        self.r = OzRng()
        self._keep = []
        if tids is not None:
            a = np.ascontiguousarray(tids, dtype=np.int32); self._keep.append(a)
            self.r.tids = a.ctypes.data_as(C.POINTER(C.c_int32)); self.r.n_tids = a.size
        if etas is not None:
            a = np.ascontiguousarray(etas, dtype=np.float64); self._keep.append(a)
            self.r.etas = a.ctypes.data_as(C.POINTER(C.c_double)); self.r.n_etas = a.size
        if unis is not None:
            a = np.ascontiguousarray(unis, dtype=np.float64); self._keep.append(a)
            self.r.unis = a.ctypes.data_as(C.POINTER(C.c_double)); self.r.n_unis = a.size
        self.r.seed = seed


def make_eval(pyfunc, real64=False):
    """pyfunc(planes int8 (17,H,W), O) -> (priors[O], value) -> C callback in the library's real type."""
    rt = np.float64 if real64 else np.float32

    def _cb(ctx, planes, H, W, O, priors, value):
        p = np.ctypeslib.as_array(planes, shape=(NPLANES, H, W))
        pr, v = pyfunc(p, O)
        np.ctypeslib.as_array(priors, shape=(O,))[:] = np.asarray(pr, dtype=rt)
        value[0] = rt(v)
    return (EVAL_FN64 if real64 else EVAL_FN)(_cb)


class Tree:
    def __init__(self, O, transform_types, d_alpha=0.03, d_epsilon=0.25, real64=False):
        self.O = O
        self.real64 = real64
        self.L = lib(real64)
        self.t = self.L.oz_tree_new(O, transform_types, d_alpha, d_epsilon)

    def __del__(self):
        if getattr(self, "t", None):
            self.L.oz_tree_free(self.t)
            self.t = None

    def search(self, state, max_playout, dirichlet, ev, rng):
        evp = ev if isinstance(ev, C.c_void_p) else C.cast(ev, C.c_void_p)
        return self.L.oz_tree_search(self.t, C.byref(state.s), max_playout, int(dirichlet), evp, None, C.byref(rng.r))

    def search_k(self, state, max_playout, K, vloss, dirichlet, ev, rng, ctx=None):
        """The several-leaves-per-round schedule (oz_tree_search_k)."""
        evp = ev if isinstance(ev, C.c_void_p) else C.cast(ev, C.c_void_p)
        rc = self.L.oz_tree_search_k(self.t, C.byref(state.s), max_playout, int(K), float(vloss), int(dirichlet), evp, ctx,
                                     C.byref(rng.r))
        if rc < 0:
            raise RuntimeError("oracle: playout deeper than OZ_KMAXD")
        return rc

    def playout(self, state, ev, rng):
        evp = ev if isinstance(ev, C.c_void_p) else C.cast(ev, C.c_void_p)
        self.L.oz_tree_playout(self.t, C.byref(state.s), evp, None, C.byref(rng.r))

    def root_stats(self):
        a = np.zeros(self.O, np.int32); n = np.zeros(self.O, np.int32)
        rt = np.float64 if self.real64 else np.float32
        w = np.zeros(self.O, rt); p = np.zeros(self.O, rt)
        c = self.L.oz_tree_root_stats(self.t, a.ctypes.data, n.ctypes.data, w.ctypes.data, p.ctypes.data)
        return a[:c], n[:c], w[:c], p[:c]

    def root_visits(self): return self.L.oz_tree_root_visits(self.t)
    def choose(self, u): return self.L.oz_tree_choose(self.t, float(u))
    def argmax(self): return self.L.oz_tree_argmax(self.t)
    def update_with_move(self, a): self.L.oz_tree_update_with_move(self.t, int(a))
    def count_expanded(self): return self.L.oz_tree_count_expanded(self.t)


def play_game(game, H, W, k, transform_types, max_playout, dirichlet_before, max_turn,
              ev, rng, d_alpha=0.03, d_epsilon=0.25, max_plies=None, real64=False, ev2=None, salts=None, K=1, vloss=1.0):
    """One game.  ev2: evaluator of player_2 (default: the same as player_1's).  salts = (a, b): context words handed to
    the two evaluators (oz_fake_eval reads a uint64 salt).  K > 1: several leaves per round (oz_tree_search_k)."""
    cfg = OzGameCfg(GAME_IDS.get(game, game), H, W, k, transform_types, max_playout,
                    dirichlet_before, max_turn, d_alpha, d_epsilon)
    O = H * W + (0 if GAME_IDS.get(game, game) == MNK else 1)
    if max_plies is None:
        max_plies = max_turn + 2
    moves = np.zeros(max_plies, np.int16)
    visits = np.zeros((max_plies, O), np.int32)
    tr = OzTrajectory()
    tr.moves = moves.ctypes.data_as(C.POINTER(C.c_int16))
    tr.visits = visits.ctypes.data_as(C.POINTER(C.c_int32))
    evp = ev if isinstance(ev, C.c_void_p) else C.cast(ev, C.c_void_p)
    ev2 = ev if ev2 is None else ev2
    evp2 = ev2 if isinstance(ev2, C.c_void_p) else C.cast(ev2, C.c_void_p)
    ctx = [None, None]
    if salts is not None:
        keep = [C.c_uint64(int(salts[0])), C.c_uint64(int(salts[1]))]
        ctx = [C.cast(C.pointer(keep[0]), C.c_void_p), C.cast(C.pointer(keep[1]), C.c_void_p)]
    rc = lib(real64).oz_play_game_ex(C.byref(cfg), evp, ctx[0], evp2, ctx[1], int(K), float(vloss), C.byref(rng.r),
                                     C.byref(tr), max_plies)
    if rc != 0:
        raise RuntimeError("oracle: RNG replay stream underflow")
    n = tr.n_plies
    return dict(moves=moves[:n].copy(), visits=visits[:n].copy(), winner=tr.winner, n_evals=tr.n_evals)
