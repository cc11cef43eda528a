#!/usr/bin/env python3
"""Phase times per round of a BatchedSelfplay run, for comparing two checkouts of this repository on one box:
    python tools/phase_probe.py <repo root> [game] [board] [sims] [blocks] [parallel] [games]"""
import os
import sys
import time

root = os.path.abspath(sys.argv[1])
sys.path.insert(0, root)
import yaml  # noqa: E402
from alphazero_amd.evaluator.nn_eval import NNEvaluator  # noqa: E402
from alphazero_amd.game.gameplay import BatchedSelfplay  # noqa: E402

game = sys.argv[2] if len(sys.argv) > 2 else "reversi"
H = int(sys.argv[3]) if len(sys.argv) > 3 else 8
sims = int(sys.argv[4]) if len(sys.argv) > 4 else 100
blocks = int(sys.argv[5]) if len(sys.argv) > 5 else 6
par = int(sys.argv[6]) if len(sys.argv) > 6 else 1024
want = int(sys.argv[7]) if len(sys.argv) > 7 else 4000
with open(os.path.join(root, "alphazero_amd", "config", game + ".yaml")) as f:
    cfg = yaml.safe_load(f)
O = H * H + (0 if game == "mnk" else 1)
cfg.update(board_height=H, board_width=H, flat_move_output=O, transform_types=8)
ev = NNEvaluator(cfg, num_blocks=blocks, channels=256, dtype="bf16", max_batch_size=1)
sp = BatchedSelfplay(ev, cfg, {"player": {"max_playout": sims}, "dirichlet_before": 30, "max_turn": 400}, n_games=par)
t0 = time.perf_counter()
hist = sp.run(want)
dt = time.perf_counter() - t0
c = sp.eng.counters()
r = max(c["steps"], 1)
print("%s: %s %dx%d %d sims %d blocks %d parallel, %d games in %.1f s: %d rounds, per round select %.4f tower %.4f finish+move+re-root %.4f ms; "
      "%.0f moves/s engine, %.0f moves/s wall" % (os.path.basename(root) or root, game, H, H, sims, blocks, par, len(hist), dt, r,
                                                 c["select_ms"] / r, c["tower_ms"] / r, c["finish_ms"] / r,
                                                 c["moves"] / ((c["select_ms"] + c["tower_ms"] + c["finish_ms"]) * 1e-3), c["moves"] / dt))
