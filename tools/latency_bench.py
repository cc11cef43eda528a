#!/usr/bin/env python3
"""Small-batch latency of the tower and of a single-game search (the play_cmd / GTP style of use):
    python tools/latency_bench.py [blocks]
Prints ms per tower forward for a few batch sizes and simulations per second of one Go 19x19 game."""
import sys
import time

import numpy as np

import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from alphazero_amd.engine import Engine  # noqa: E402
from alphazero_amd.network.model import fold_to_blob, random_net  # noqa: E402

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 20
H = 19
O = H * H + 1
blob = fold_to_blob(random_net(H, H, O, NB, 256, seed=0))
eng = Engine("go", H, H, n_slots=64, num_blocks=NB, channels=256, tower_dtype="bf16")
eng.tower_load(blob)
rng = np.random.RandomState(0)
for B in (1, 2, 4, 8, 16, 17, 32, 64):
    x = (rng.rand(B, 17, H, H) < 0.3).astype(np.int8)
    eng.tower_forward(x)
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        eng.tower_forward(x)
    dt = (time.perf_counter() - t0) / n
    print("tower forward B=%3d: %7.3f ms  (%8.1f evals/s)" % (B, dt * 1e3, B / dt), flush=True)
eng.close()

for K in (1, 4, 8, 16, 32, 64):
    e = Engine("go", H, H, n_slots=1, max_playout=800, transform_types=8, num_blocks=NB, channels=256, tower_dtype="bf16", seed=3,
               leaves_per_round=K)
    e.tower_load(blob)
    e.search_begin([0], [0], [1])
    t0 = time.perf_counter()
    sims = 0
    while True:
        n = e.step_select()
        if n == 0:
            break
        e.step_finish()
        sims += n
    dt = time.perf_counter() - t0
    print("single game search, %2d leaves per round: %d simulations in %.3f s = %.0f sims/s" % (K, sims, dt, sims / dt), flush=True)
    e.close()

one = Engine("go", H, H, n_slots=1, max_playout=400, transform_types=8, num_blocks=NB, channels=256, tower_dtype="bf16", seed=3)
one.tower_load(blob)
one.search_begin([0], [0], [1])
t0 = time.perf_counter()
sims = 0
while one.step_select() > 0:
    one.step_finish()
    sims += 1
dt = time.perf_counter() - t0
print("single game search: %d simulations in %.3f s = %.0f sims/s" % (sims, dt, sims / dt))
one.close()
