#!/usr/bin/env python3
"""Cost of the board groups that turn out empty: 3993 slots of Go 19x19, 20 blocks, only the first `live` slots searching
(leaf compaction lists them on the device; the host launches every group's kernels, the empty ones exit at once).
    AZ_TOWER_GROUP_BOARDS=-1|0 python tools/empty_groups.py [live]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from alphazero_amd.engine import Engine  # noqa: E402
from alphazero_amd.network.model import fold_to_blob, random_net  # noqa: E402

live = int(sys.argv[1]) if len(sys.argv) > 1 else 363
G, H = 3993, 19
eng = Engine("go", H, H, n_slots=G, max_playout=60, transform_types=8, num_blocks=20, channels=256, tower_dtype="bf16", seed=1)
eng.tower_load(fold_to_blob(random_net(H, H, H * H + 1, 20, 256, seed=0)))
slots = np.arange(live, dtype=np.int32)
eng.search_begin(slots, np.zeros(live, np.int32), np.zeros(live, np.int32))
for _ in range(3):
    eng.step_select(); eng.step_finish()
eng.sync()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    eng.step_select(); eng.step_finish()
eng.sync()
dt = (time.perf_counter() - t0) / n
print("groups %s: %d of %d slots searching: %.3f ms per round" % (os.environ.get("AZ_TOWER_GROUP_BOARDS", "0"), live, G, dt * 1e3))
eng.close()
