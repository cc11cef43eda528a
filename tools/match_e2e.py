#!/usr/bin/env python3
"""Gating match throughput through BatchedMatch (two resident weight sets): games/s and moves/s.
    python tools/match_e2e.py [game] [board] [sims] [blocks] [parallel] [games]"""
import os
import sys
import time

import yaml

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from alphazero_amd.evaluator.nn_eval import NNEvaluator  # noqa: E402
from alphazero_amd.game.gameplay import BatchedMatch  # noqa: E402

game = sys.argv[1] if len(sys.argv) > 1 else "reversi"
H = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sims = int(sys.argv[3]) if len(sys.argv) > 3 else 100
blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 6
par = int(sys.argv[5]) if len(sys.argv) > 5 else 1024
games = int(sys.argv[6]) if len(sys.argv) > 6 else par
with open(os.path.join(ROOT, "alphazero_amd", "config", game + ".yaml")) as f:
    cfg = yaml.safe_load(f)
O = H * H + (0 if game == "mnk" else 1)
cfg.update(board_height=H, board_width=H, flat_move_output=O, transform_types=8)
chal = NNEvaluator(cfg, num_blocks=blocks, channels=256, dtype="bf16", max_batch_size=1, seed=1)
best = NNEvaluator(cfg, num_blocks=blocks, channels=256, dtype="bf16", max_batch_size=1, seed=2)
ext = {"num_games": games, "num_worker": par,
       "gameplay": {"dirichlet_before": 30, "log_iter": 30, "max_turn": 400, "player": {"max_playout": sims}}}
m = BatchedMatch(chal, best, cfg, ext, n_parallel=par)
t0 = time.perf_counter()
out = m.run()
dt = time.perf_counter() - t0
c = m.eng.counters()
gpu_s = (c["select_ms"] + c["tower_ms"] + c["finish_ms"]) * 1e-3
print("%s %dx%d, %d sims, %d blocks: %d games on %d slots in %.1f s; win_counter %d / not tie %d -> accepted %s"
      % (game, H, H, sims, blocks, games, par, dt, out["win_counter"], out["num_not_tie"], out["accepted"]))
print("  %d moves: %.0f moves/s end to end, %.0f moves/s of GPU phase time" % (c["moves"], c["moves"] / dt, c["moves"] / gpu_s))
