#!/usr/bin/env python3
"""Round time of the device loop with the tower run on every leaf row against the compacted rows only.
    python tools/compaction_ab.py [game] [board] [sims] [blocks] [slots] [rounds] [skip_rounds]"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from alphazero_amd.engine import Engine  # noqa: E402
from alphazero_amd.network.model import fold_to_blob, random_net  # noqa: E402

game = sys.argv[1] if len(sys.argv) > 1 else "reversi"
H = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sims = int(sys.argv[3]) if len(sys.argv) > 3 else 100
blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 6
G = int(sys.argv[5]) if len(sys.argv) > 5 else 4096
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 2000
skip = int(sys.argv[7]) if len(sys.argv) > 7 else 0
O = H * H + (0 if game == "mnk" else 1)
blob = fold_to_blob(random_net(H, H, O, blocks, 256, seed=0))
for compact in (False, True, False, True):
    eng = Engine(game, H, H, 5, n_slots=G, max_playout=sims, transform_types=8, dirichlet_before=30, max_turn=400, seed=3,
                 num_blocks=blocks, channels=256, tower_dtype="bf16")
    eng.tower_load(blob)
    eng.tower_leaf_compaction(compact)
    eng.selfplay_start()
    eng.selfplay_run(skip + 5)
    eng.counters_reset()
    eng.sync()
    t0 = time.perf_counter()
    eng.selfplay_run(rounds)
    eng.sync()
    dt = time.perf_counter() - t0
    c = eng.counters()
    print("%s %dx%d %d sims, %d slots, compaction %-5s: %.3f ms/round, %d playouts, %d boards through the tower (%.1f %%), %d moves"
          % (game, H, H, sims, G, compact, dt * 1e3 / rounds, c["evals"], c["tower_evals"], 100.0 * c["tower_evals"] / max(rounds * G, 1), c["moves"]),
          flush=True)
    eng.close()
