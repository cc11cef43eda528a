#!/usr/bin/env python3
"""How full the fixed per-tree arenas get over WHOLE self-play games (tree reuse from move to move), and how many
expansions the "leave it a leaf" policy of BatchedSelfplay skipped, for a flat and for a sharpened policy head.
The reference's tree has no limit (search/mcts.py:336-346 keeps the chosen child's subtree for the next search), so
a skipped expansion is a departure from it; this prints how often it happens.

    python tools/whole_game_arena.py [board] [sims] [parallel] [sharpen] [blocks]
sharpen multiplies the policy head's last layer: 1 = the random network as it is (flat priors), 30-100 = a policy that puts
most of its mass on a few points, as a trained network does."""
import os
import sys
import time

import numpy as np
import torch
import yaml

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from alphazero_amd.evaluator.nn_eval import NNEvaluator  # noqa: E402
from alphazero_amd.game.gameplay import BatchedSelfplay  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 19
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 800
par = int(sys.argv[3]) if len(sys.argv) > 3 else 32
sharpen = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
blocks = int(sys.argv[5]) if len(sys.argv) > 5 else 2
with open(os.path.join(ROOT, "alphazero_amd", "config", "go.yaml")) as f:
    cfg = yaml.safe_load(f)
O = H * H + 1
cfg.update(board_height=H, board_width=H, flat_move_output=O, transform_types=8)
ev = NNEvaluator(cfg, num_blocks=blocks, channels=256, dtype="bf16", max_batch_size=1)
if sharpen != 1.0:
    with torch.no_grad():
        ev.net.p_fc.weight.mul_(sharpen)
        ev.net.p_fc.bias.mul_(sharpen)
    ev._refold()
sp = BatchedSelfplay(ev, cfg, {"player": {"max_playout": sims}, "dirichlet_before": 30, "max_turn": 400}, n_games=par)
t0 = time.perf_counter()
hist = sp.run(par)
dt = time.perf_counter() - t0
c = sp.eng.counters()
plies = [h[0].shape[0] for h in hist]
pi_max = np.concatenate([h[1].max(axis=1) for h in hist])
print("go %dx%d, %d sims/move, %d games side by side, policy head x%g: %d games, %d plies (%.0f per game) in %.0f s"
      % (H, H, sims, par, sharpen, len(hist), sum(plies), np.mean(plies), dt))
print("  visit share of the most visited move: mean %.3f, 90th percentile %.3f" % (pi_max.mean(), np.percentile(pi_max, 90)))
print("  evaluations %d, expansions skipped %d (%.2e of evaluations)" % (c["evals"], c["expansions_skipped"],
                                                                         c["expansions_skipped"] / max(c["evals"], 1)))
print("  fullest tree: %d of %d edges (%.1f %%), %d of %d blocks (%.1f %%)" % (
    c["peak_tree_edges"], c["tree_edge_cap"], 100.0 * c["peak_tree_edges"] / c["tree_edge_cap"],
    c["peak_tree_blocks"], c["tree_block_cap"], 100.0 * c["peak_tree_blocks"] / c["tree_block_cap"]))
print("  shared edge pool: at most %d of %d pages of %d edges in use at once (%.1f %%; %.1f %% of what one arena per tree would take)" % (
    c["tree_pool_pages_peak"], c["tree_pool_pages"], c["tree_page_edges"], 100.0 * c["tree_pool_pages_peak"] / c["tree_pool_pages"],
    100.0 * c["tree_pool_pages_peak"] * c["tree_page_edges"] / (2.0 * par * c["tree_edge_cap"])))
