"""Drop-in for AlphaZero.game.gameplay (config key gameplay_path): one two-player game.

`Game` keeps the reference's constructor, `start()`, `get_history()`, `.winner`, `.state`
(game/gameplay.py:17-94).  For thousands of concurrent games use `BatchedSelfplay`, which drives
the engine's device-side self-play machine and returns the same `get_history()` tuples.
"""
import importlib

import numpy as np

from ..engine import Engine, game_config_to_engine_kwargs
from ..player import mcts_player as player


def printlog(*args):
    pass


def history_arrays(eng, moves, visits, winner, slot=0):
    """(state_np (T,17,H,W) f64, probs_np (T,O) f64, result_np (T,) f64) of game/gameplay.py:73-94,
    the states re-derived by replaying `moves` through the engine's rule and feature kernels."""
    T = len(moves)
    H, W, O = eng.H, eng.W, eng.O
    state_np = np.zeros((T, 17, H, W))
    eng.reset([slot])
    for i in range(T):
        state_np[i] = eng.features([slot], [0])[0]
        status, _ = eng.do_move([slot], [int(moves[i])])
        assert status[0] == 0
    probs_np = np.asarray(visits, np.float64) / np.asarray(visits, np.float64).sum(axis=1, keepdims=True) if T else np.zeros((0, O))
    return state_np, probs_np, outcome_labels(T, winner)


def history_arrays_batch(eng, games):
    """history_arrays for many finished games at once: `games` = [(moves, visits, winner)], at most eng.G of them.
    All games are replayed in lockstep, one batched feature call and one batched move call per ply."""
    n = len(games)
    assert n <= eng.G
    H, W, O = eng.H, eng.W, eng.O
    lens = [len(g[0]) for g in games]
    states = [np.zeros((T, 17, H, W)) for T in lens]
    eng.reset(list(range(n)))
    for i in range(max(lens, default=0)):
        act = [k for k in range(n) if lens[k] > i]
        planes = eng.features(act, [0] * len(act))
        for q, k in enumerate(act):
            states[k][i] = planes[q]
        status, _ = eng.do_move(act, [int(games[k][0][i]) for k in act])
        assert np.all(status == 0)
    out = []
    for k, (moves, visits, winner) in enumerate(games):
        T = lens[k]
        v = np.asarray(visits, np.float64)
        probs_np = v / v.sum(axis=1, keepdims=True) if T else np.zeros((0, O))
        out.append((states[k], probs_np, outcome_labels(T, winner)))
    return out


def outcome_labels(n_plies, winner, black=1):
    """result_np of game/gameplay.py:93: ply i is labelled +1 when the side to move at ply i (Black on even plies)
    is the winner; a draw (winner 0) is labelled as a White win, as the reference does."""
    black_to_move = (np.arange(n_plies) % 2) == 0
    return np.where(black_to_move == (winner == black), 1.0, -1.0)


class Game:
    """One game between two evaluators (game/gameplay.py:17-94): Game(nn_eval_1, nn_eval_2, game_config, ext_config),
    start() -> winner, get_history() -> (state_np, probs_np, result_np); player_1 moves first."""

    def __init__(self, nn_eval_1, nn_eval_2, game_config, ext_config):
        env = importlib.import_module(game_config['env_path'])
        self._pass, self._black = env.PASS_MOVE, env.BLACK
        self.state = env.GameState()
        self.player_1, self.player_2 = (player.Player(ev, game_config, ext_config['player']) for ev in (nn_eval_1, nn_eval_2))
        self.winner = None
        self._shape = (game_config['history_step'] * game_config['planes_per_step'] + game_config['additional_planes'],
                       game_config['board_height'], game_config['board_width'])
        self._n_actions = game_config['flat_move_output']
        self.dirichlet_before, self.log_iter, self.max_turn = (ext_config[k] for k in ('dirichlet_before', 'log_iter', 'max_turn'))
        self._plies = []   # per ply: (planes of the pre-move state, [(action, pi)], move)

    # the reference keeps three parallel lists; they are views of the ply records here
    state_history = property(lambda self: [p[0] for p in self._plies])
    probs_history = property(lambda self: [p[1] for p in self._plies])
    acts_history = property(lambda self: [p[2] for p in self._plies])

    def _ply(self):
        st = self.state
        mover = self.player_2 if len(self._plies) & 1 else self.player_1
        move, probs = mover.think(st, st.turns <= self.dirichlet_before)
        self._plies.append((st.features(0), probs, move))
        st.do_move(move)
        for p in (self.player_1, self.player_2):   # both trees follow every move
            p.ack(move)

    def start(self):
        st = self.state
        while not st.is_end_of_game and st.turns <= self.max_turn:
            if st.turns % self.log_iter == 0:
                printlog(str(st.turns), 'moves')
            self._ply()
        self.winner = st.get_winner()
        printlog('end', self.winner)
        return self.winner

    def _flat(self, action):
        _, h, w = self._shape
        return h * w if action == self._pass else action[0] * w + action[1]

    def get_history(self):
        T = len(self._plies)
        state_np = np.zeros((T,) + self._shape)
        probs_np = np.zeros((T, self._n_actions))
        for i, (planes, probs, _) in enumerate(self._plies):
            state_np[i] = planes
            if probs:
                probs_np[i, [self._flat(a) for a, _ in probs]] = [p for _, p in probs]
        return state_np, probs_np, outcome_labels(T, self.winner, self._black)


def grow_pool_when_used(eng, share=0.8):
    """Doubles the engine's shared pool of tree-edge pages once more than `share` of it has been in use at some moment
    (and every tree at its limit would need more than it holds): called between polls by the batched drivers, so that an
    expansion never lacks a page although the pool is not sized for the worst case up front."""
    c = eng.counters()
    if c["tree_pool_pages_peak"] > share * c["tree_pool_pages"]:
        most = 2 * eng.G * -(-c["tree_edge_cap"] // c["tree_page_edges"])
        if c["tree_pool_pages"] < most:
            eng.grow_pool()
            return True
    return False


class BatchedSelfplay:
    """`n_games` concurrent self-play games on one GPU; the role of Selfplay.run +
    selfplay_wrapper (train/parallel/selfplay.py:71-108) without a process per game.

    The engine runs with tree_full_policy 1: should a search tree ever outgrow its limit (long forced lines with
    almost all visits kept across re-roots), that playout's node stays a leaf instead of stopping every game;
    `eng.counters()["expansions_skipped"]` tells how often that happened (0 = the run is exact).  The pool of edge
    pages all trees share starts at half of what every tree at its limit would need and is doubled between polls
    whenever more than POOL_GROW_AT of it has been in use, so a page is never what an expansion lacks.

    Finished games are turned into get_history() tuples on a second engine (its own HIP stream) by a
    background thread, so the self-play engine keeps running rounds meanwhile.

    New weights (nn_eval.load / set_weights bump nn_eval.version; selfplay.py:110-122) are picked up between polls.
    update="immediate" (default): the games in flight continue with the new network from the next round on.
    update="drain": as in the reference, where every running game holds the evaluator's read lock and the reload waits
    for all of them - finished slots are not re-armed until every game in flight has ended, then the weights are
    loaded and all slots start again, so no game ever sees two networks.

    rng_streams: optional list of (transform ids, gamma draws, uniforms), one per slot - the engine then replays these
    draws instead of using its own generator (parity tests)."""

    REPLAY_SLOTS = 64
    POOL_GROW_AT = 0.8

    def __init__(self, nn_eval, game_config, ext_config, n_games, seed=0, device=0, update="immediate", rng_streams=None,
                 tree_pool_edges=0):
        if update not in ("immediate", "drain"):
            raise ValueError("update must be 'immediate' or 'drain'")
        kw = game_config_to_engine_kwargs(game_config)
        self.eng = Engine(n_slots=n_games, max_playout=ext_config['player']['max_playout'],
                          dirichlet_before=ext_config['dirichlet_before'], max_turn=ext_config['max_turn'],
                          record_trajectories=True, seed=seed, num_blocks=nn_eval.num_blocks, channels=nn_eval.channels,
                          tower_dtype=nn_eval.dtype, device=device, tree_full_policy=1, rng_replay=rng_streams is not None,
                          tree_pool_edges=tree_pool_edges, **kw)
        if rng_streams is not None:
            if len(rng_streams) != n_games:
                raise ValueError("one (tids, etas, unis) triple per slot")
            for s, st in enumerate(rng_streams):
                self.eng.rng_inject(s, *st)
        self._nn_eval = nn_eval
        self._version = None
        self._replay = Engine(n_slots=self.REPLAY_SLOTS, device=device, **kw)
        self.n_games = n_games
        self._started = False
        self._update = update
        self._draining = False
        self._empty = np.zeros(n_games, bool)   # slots popped without a restart while draining
        self._surplus = []                      # (slot, history) of games that finished beyond an earlier run()'s request
        self.last_slots = []                    # slot of every game the last run() returned

    def _sync_weights(self):
        if self._version != self._nn_eval.version:
            self.eng.tower_load(self._nn_eval._tower_blob)
            self._version = self._nn_eval.version

    def _convert(self, popped):
        out = []
        for i in range(0, len(popped), self.REPLAY_SLOTS):
            out += history_arrays_batch(self._replay, popped[i:i + self.REPLAY_SLOTS])
        return out

    def _watch_pool(self):
        grow_pool_when_used(self.eng, self.POOL_GROW_AT)

    def _weights_between_polls(self):
        if self._version == self._nn_eval.version:
            return
        if self._update == "immediate" or self._version is None:
            self._sync_weights()
        else:
            self._draining = True

    def run(self, num_games, rounds_per_poll=64):
        """Play until `num_games` games have finished; returns their get_history() tuples in finishing order.  Games
        that finish in the same poll beyond the request are kept and returned first by the next call."""
        from concurrent.futures import ThreadPoolExecutor
        self._weights_between_polls()
        if not self._started:
            self.eng.selfplay_start()
            self._started = True
        slots = [s for s, _ in self._surplus]
        ready = [h for _, h in self._surplus]
        pending = []
        with ThreadPoolExecutor(max_workers=1) as pool:   # one worker: conversions stay in finishing order
            while len(slots) < num_games:
                self.eng.selfplay_run(rounds_per_poll)    # the C call releases the GIL: the worker converts meanwhile
                plies, done = self.eng.selfplay_status()
                self._watch_pool()
                fin = np.flatnonzero((done != 0) & ~self._empty)
                popped = [(tr["moves"], tr["visits"], tr["winner"])
                          for tr in self.eng.trajectory_pop_many(fin, plies[fin], restart=not self._draining)]
                if popped:
                    slots += [int(s) for s in fin]
                    pending.append(pool.submit(self._convert, popped))
                if self._draining:
                    self._empty[fin] = True
                    if self._empty.all():                 # every game in flight has ended: swap the network, start over
                        self._sync_weights()
                        self.eng.selfplay_start()
                        self._empty[:] = False
                        self._draining = False
                else:
                    self._weights_between_polls()
            out = ready + [h for f in pending for h in f.result()]
        self._surplus = list(zip(slots[num_games:], out[num_games:]))
        self.last_slots = slots[:num_games]
        return out[:num_games]


def gating_colors(num_games):
    """Colour of the challenger in each game of a gating match: alternating, Black first
    (train/parallel/evaluator.py:114-115)."""
    return [1, -1] * (num_games // 2) + [1] * (num_games % 2)


def gating_verdict(winners, colors_of_new):
    """(win_counter, num_not_tie, accepted) by the rule of train/parallel/evaluator.py:86-89,122:
    a game counts for the challenger when its winner equals the challenger's colour, draws are left
    out of the denominator, and the challenger is accepted iff wins > int(0.55 * num_not_tie)."""
    win_counter = sum(1 for w, c in zip(winners, colors_of_new) if w == c)
    num_not_tie = sum(1 for w in winners if w is not None and w != 0)
    return win_counter, num_not_tie, win_counter > int(0.55 * num_not_tie)


class BatchedMatch:
    """Gating match between a challenger and the best model, all games concurrently on one GPU with both
    weight sets resident: the role of Evaluator.run + eval_wrapper (train/parallel/evaluator.py:70-126),
    i.e. Game(chal, best, ...) when the challenger is Black and Game(best, chal, ...) otherwise.

    ext_config is the reference's `evaluator` section: {'num_games', 'num_worker', 'gameplay': {...}};
    `n_parallel` (default: every game at once) replaces num_worker as the concurrency limit."""

    def __init__(self, nn_eval_chal, nn_eval_best, game_config, ext_config, n_parallel=None, seed=0, device=0):
        self.nn_eval_chal, self.nn_eval_best = nn_eval_chal, nn_eval_best
        self.game_config, self.ext_config = game_config, ext_config
        self.num_games = ext_config['num_games']
        self.n_parallel = min(self.num_games, n_parallel or self.num_games)
        self.seed, self.device = seed, device
        self.eng = None

    def _engine(self):
        if self.eng is None:
            gp = self.ext_config['gameplay']
            ev = self.nn_eval_chal
            self.eng = Engine(n_slots=self.n_parallel, max_playout=gp['player']['max_playout'],
                              dirichlet_before=gp['dirichlet_before'], max_turn=gp['max_turn'], record_trajectories=True,
                              seed=self.seed, num_blocks=ev.num_blocks, channels=ev.channels, tower_dtype=ev.dtype,
                              device=self.device, tree_full_policy=1, **game_config_to_engine_kwargs(self.game_config))
        return self.eng

    def run(self, rounds_per_poll=64):
        """Plays ext_config['num_games'] games; returns a dict with the reference's counters
        (win_counter, num_not_tie), the acceptance decision and the per-game winners / colours."""
        eng = self._engine()
        eng.tower_load(self.nn_eval_chal._tower_blob, net=0)   # network 0 = challenger
        eng.tower_load(self.nn_eval_best._tower_blob, net=1)
        colors = gating_colors(self.num_games)
        winners = [None] * self.num_games
        game_of_slot = list(range(self.n_parallel))
        next_game = self.n_parallel
        eng.selfplay_start()
        eng.match_assign(list(range(self.n_parallel)), [0 if colors[g] == 1 else 1 for g in game_of_slot])
        finished = 0
        while finished < self.num_games:
            eng.selfplay_run(rounds_per_poll)
            plies, done = eng.selfplay_status()
            grow_pool_when_used(eng)
            fin = [int(s) for s in np.flatnonzero(done) if game_of_slot[int(s)] is not None]
            n_more = min(len(fin), self.num_games - next_game)
            again, last = fin[:n_more], fin[n_more:]      # slots that take another game / that stay empty
            for group, restart in ((again, True), (last, False)):
                for s, tr in zip(group, eng.trajectory_pop_many(group, plies[group], restart=restart)):
                    winners[game_of_slot[s]] = tr["winner"]
                    finished += 1
            for s in again:
                game_of_slot[s] = next_game
                next_game += 1
            if again:
                eng.match_assign(again, [0 if colors[game_of_slot[s]] == 1 else 1 for s in again])
            for s in last:
                game_of_slot[s] = None
        win_counter, num_not_tie, accepted = gating_verdict(winners, colors)
        return {"win_counter": win_counter, "num_not_tie": num_not_tie, "win_rate": win_counter / (num_not_tie + 1e-9),
                "accepted": accepted, "winners": winners, "colors_of_new": colors}
