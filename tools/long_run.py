#!/usr/bin/env python3
"""A long self-play run on the headline network, watched ply by ply: what the opening-move bench cannot show.

    python tools/long_run.py [games=1452] [plies=40] [sims=800] [blocks=20] [dtype=bf16] [pool=0.3333] [sharpen=1]

Go 19x19, stem + `blocks` residual blocks x 256 channels, device RNG, transforms and root noise as in bench.py, trees kept
from move to move (search/mcts.py:336-346).  The run is cut into chunks of `sims` rounds (about one ply of every game); after
each chunk the engine's counters are read and reset: plies played so far, moves/s and simulations/s of the chunk, tree
levels and children scanned per simulation, the fullest tree of the chunk against the per-tree limit, the pages the trees
hold in the shared edge pool, expansions skipped (tree_full_policy 1: a skipped expansion would be a departure from the
reference's unlimited tree, so the count says whether the run was exact).  `sharpen` multiplies the policy head's last
layer (1 = the flat priors of a random network; 30 = most visits on one move, as late in training: deeper, narrower trees
and larger kept subtrees)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from alphazero_amd.engine import Engine  # noqa: E402
from alphazero_amd.network.model import fold_to_blob, random_net  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1452
PLIES = int(sys.argv[2]) if len(sys.argv) > 2 else 40
SIMS = int(sys.argv[3]) if len(sys.argv) > 3 else 800
NB = int(sys.argv[4]) if len(sys.argv) > 4 else 20
DT = sys.argv[5] if len(sys.argv) > 5 else "bf16"
POOL = float(sys.argv[6]) if len(sys.argv) > 6 else 1.0 / 3.0
SHARPEN = float(sys.argv[7]) if len(sys.argv) > 7 else 1.0
H, C = 19, 256
O = H * H + 1

net = random_net(H, H, O, NB, C, seed=0)
if SHARPEN != 1.0:
    with torch.no_grad():
        net.p_fc.weight.mul_(SHARPEN)
        net.p_fc.bias.mul_(SHARPEN)
eng = Engine("go", H, H, n_slots=G, max_playout=SIMS, transform_types=8, dirichlet_before=30, max_turn=400, seed=1000,
             record_trajectories=True, num_blocks=NB, channels=C, tower_dtype=DT, tree_full_policy=1,
             tree_pool_edges=max(1, int(POOL * 2 * G * 3 * (SIMS + 1) * O)))
eng.tower_load(fold_to_blob(net))
eng.selfplay_start()
c0 = eng.counters()
print("# Go %dx%d self-play, %d games on one MI355X, %d simulations per move, stem + %d residual blocks x %d channels (%s), policy head x%g,"
      % (H, H, G, SIMS, NB, C, DT, SHARPEN))
print("# device RNG, dihedral transforms on, root noise on the first 31 plies, trees kept from move to move; one line per %d rounds." % SIMS)
print("# per-tree limit %d edges / %d expanded nodes; shared edge pool %d pages of %d edges = %.1f GB (%.0f %% of what one arena per tree would take)"
      % (c0["tree_edge_cap"], c0["tree_block_cap"], c0["tree_pool_pages"], c0["tree_page_edges"],
         c0["tree_pool_pages"] * c0["tree_page_edges"] * 18 / 1e9,
         100.0 * c0["tree_pool_pages"] * c0["tree_page_edges"] / (2.0 * G * c0["tree_edge_cap"])))
print("# %6s %11s %9s %10s %9s %11s %13s %13s %12s %12s %8s" % ("rounds", "plies/game", "moves/s", "sims/s", "ms/round", "levels/sim",
                                                              "children/sim", "fullest tree", "pool in use", "pool peak", "skipped"))
t_all = time.perf_counter()
rounds = 0
total = dict(moves=0, evals=0, skipped=0, games=0)
peak_tree, peak_pool = 0, 0
while True:
    plies, done = eng.selfplay_status()
    live = done == 0   # (a game that two passes ended early keeps its ply count)
    if not live.any() or plies[live].min() >= PLIES:
        break
    eng.counters_reset()
    t0 = time.perf_counter()
    eng.selfplay_run(SIMS)
    dt = time.perf_counter() - t0
    rounds += SIMS
    c = eng.counters()
    plies, done = eng.selfplay_status()
    ev = max(c["evals"], 1)
    for k, name in (("moves", "moves"), ("evals", "evals"), ("skipped", "expansions_skipped"), ("games", "games_finished")):
        total[k] += c[name]
    peak_tree = max(peak_tree, c["peak_tree_edges"]); peak_pool = max(peak_pool, c["tree_pool_pages_peak"])
    print("  %6d %5.1f-%-5d %9.1f %10.0f %9.2f %11.2f %13.1f %8d %3.0f%% %8d %3.0f%% %8d %3.0f%% %8d" % (
        rounds, plies.mean(), plies.max(), c["moves"] / dt, c["evals"] / dt, dt * 1e3 / SIMS, c["levels"] / ev, c["children_scanned"] / ev,
        c["peak_tree_edges"], 100.0 * c["peak_tree_edges"] / c["tree_edge_cap"],
        c["tree_pool_pages_in_use"], 100.0 * c["tree_pool_pages_in_use"] / c["tree_pool_pages"],
        c["tree_pool_pages_peak"], 100.0 * c["tree_pool_pages_peak"] / c["tree_pool_pages"], c["expansions_skipped"]), flush=True)
wall = time.perf_counter() - t_all
c = eng.counters()
print("# whole run: %d rounds in %.0f s, %d moves (%.1f per game), %d simulations -> %.1f moves/s, %.0f sims/s; games finished %d"
      % (rounds, wall, total["moves"], total["moves"] / G, total["evals"], total["moves"] / wall, total["evals"] / wall, total["games"]))
print("# fullest tree of the run %d edges (%.1f %% of the per-tree limit); most pool pages in use at once %d (%.1f %% of the pool, %.1f GB); "
      "expansions skipped %d of %d (%s)"
      % (peak_tree, 100.0 * peak_tree / c["tree_edge_cap"], peak_pool, 100.0 * peak_pool / c["tree_pool_pages"],
         peak_pool * c["tree_page_edges"] * 18 / 1e9, total["skipped"], total["evals"],
         "the run is exact" if total["skipped"] == 0 else "not exact"))
eng.close()
