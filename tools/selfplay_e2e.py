#!/usr/bin/env python3
"""End-to-end self-play through the drop-in API (BatchedSelfplay.run -> Game.get_history() tuples), host work
included: games/s and moves/s as a training loop would see them.
    python tools/selfplay_e2e.py [game] [board] [sims] [blocks] [n_games_parallel] [games_to_collect]"""
import os
import sys
import time

import numpy as np
import yaml

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from alphazero_amd.evaluator.nn_eval import NNEvaluator  # noqa: E402
from alphazero_amd.game.gameplay import BatchedSelfplay  # noqa: E402

game = sys.argv[1] if len(sys.argv) > 1 else "reversi"
H = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sims = int(sys.argv[3]) if len(sys.argv) > 3 else 100
blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 6
par = int(sys.argv[5]) if len(sys.argv) > 5 else 1024
want = int(sys.argv[6]) if len(sys.argv) > 6 else par
with open(os.path.join(ROOT, "alphazero_amd", "config", game + ".yaml")) as f:
    cfg = yaml.safe_load(f)
O = H * H + (0 if game == "mnk" else 1)
cfg.update(board_height=H, board_width=H, flat_move_output=O, transform_types=8)
ev = NNEvaluator(cfg, num_blocks=blocks, channels=256, dtype="bf16", max_batch_size=1)
sp = BatchedSelfplay(ev, cfg, {"player": {"max_playout": sims}, "dirichlet_before": 30, "max_turn": 400}, n_games=par)
t0 = time.perf_counter()
hist = sp.run(want)
dt = time.perf_counter() - t0
plies = sum(h[0].shape[0] for h in hist)
c = sp.eng.counters()
gpu_s = (c["select_ms"] + c["tower_ms"] + c["finish_ms"]) * 1e-3
print("%s %dx%d, %d sims, %d blocks, %d parallel: %d games (%d plies) in %.1f s" % (game, H, H, sims, blocks, par, len(hist), plies, dt))
print("  engine: %d moves, %.1f s of GPU phases -> %.0f moves/s; end to end incl. host: %.0f moves/s (all %d moves / wall)"
      % (c["moves"], gpu_s, c["moves"] / gpu_s, c["moves"] / dt, c["moves"]))
print("  history bytes returned: %.1f MB" % (sum(h[0].nbytes + h[1].nbytes + h[2].nbytes for h in hist) / 1e6))
# the shared pool of tree edges after thousands of games started and ended in these slots: every page in use must belong to
# exactly one tree (a page lost at a restart or a re-root would show here as pages in use that no tree holds)
page = c["tree_page_edges"]
held = sum(max(1, -(-sp.eng.tree_size(s, side)[0] // page)) for s in range(par) for side in (0, 1))
print("  tree edge pool: %d pages of %d edges, peak %d (%.1f %%), in use now %d, held by the trees %d -> %s; expansions skipped %d"
      % (c["tree_pool_pages"], page, c["tree_pool_pages_peak"], 100.0 * c["tree_pool_pages_peak"] / c["tree_pool_pages"],
         c["tree_pool_pages_in_use"], held, "no page lost" if held == c["tree_pool_pages_in_use"] else "PAGES LOST", c["expansions_skipped"]))
assert held == c["tree_pool_pages_in_use"]
