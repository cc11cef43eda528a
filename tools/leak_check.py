#!/usr/bin/env python3
"""Creates, uses and closes engines in a loop and checks that free device memory does not drift.
    python tools/leak_check.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from alphazero_amd.engine import Engine
from alphazero_amd.network.model import fold_to_blob, random_net
blob = fold_to_blob(random_net(9, 9, 82, 2, 256, seed=0))
x = (np.random.rand(40, 17, 9, 9) < 0.3).astype(np.int8)
free0 = None
for it in range(40):
    e = Engine("go", 9, 9, n_slots=300, max_playout=20, transform_types=8, num_blocks=2, channels=256, tower_dtype="bf16", record_trajectories=True, seed=it)
    e.tower_load(blob); e.tower_load(blob, net=1)
    e.tower_forward(x[:3]); e.tower_forward(x)
    e.selfplay_start(); e.selfplay_run(30)
    e.match_assign(np.arange(300), np.arange(300) % 2)   # two towers side by side: the second stream and its events
    e.tower_profile(8)                                   # sampled per-launch timing: the tower's event pool
    e.selfplay_run(20)
    pl, dn = e.selfplay_status()
    e.grow_trees(); e.grow_pool()                        # limits, page tables and the edge pool re-allocated
    e.rng_inject(0, [1, 2, 3], [0.5], [0.1])
    e.close()
    free, tot = torch.cuda.mem_get_info(0)
    if it == 2: free0 = free
    if it % 10 == 9: print(it, "free MB", free >> 20, flush=True)
assert abs(free - free0) < 64 << 20, (free0, free)
print("no leak:", (free0 - free) >> 20, "MB drift")
