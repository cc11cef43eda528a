#!/usr/bin/env python3
"""Where one wavefront of k_select spends its cycles (development build with shader-clock stamps).
    make -C alphazero_amd/csrc stamp      # builds tools/stamp/libazamd.so with -DAZ_STAMP_SELECT
    python tools/select_stamps.py [board] [sims]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import alphazero_amd._lib as _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, "tools", "stamp", "libazamd.so")
from alphazero_amd.engine import Engine  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 19
P = int(sys.argv[2]) if len(sys.argv) > 2 else 800
G = int(sys.argv[3]) if len(sys.argv) > 3 else 1
O = H * H + 1
eng = Engine("go", H, H, n_slots=G, max_playout=P, transform_types=8, seed=5)
eng._L.az_debug_select_stamps.argtypes = [C.c_void_p, C.c_void_p]
rng = np.random.RandomState(0)
names = ["state load", "search start", "child scan + arg-max", "rule move", "visit store + child ptr", "transform id",
         "legal mask", "mask store + winner", "feature planes"]
for ply in range(3):
    eng.search_begin(list(range(G)), [ply & 1] * G, [1] * G)
    out = np.zeros(16, np.uint64)
    eng._L.az_debug_select_stamps(eng._h, out.ctypes.data)
    n = 0
    while eng.step_select() > 0:
        pri = rng.dirichlet(np.ones(O) * 0.3, size=G).astype(np.float32)
        eng.step_finish(pri, rng.uniform(-1, 1, G).astype(np.float32))
        n += 1
    eng._L.az_debug_select_stamps(eng._h, out.ctypes.data)
    tot = float(out[:9].sum())
    c = eng.counters()
    print("ply %d: %d playouts, %.0f cycles per k_select of slot 0 (%.2f levels per playout so far)" % (ply, n, tot / n, c["levels"] / max(c["evals"], 1)))
    for i, nm in enumerate(names):
        print("   %-26s %8.0f cycles  %5.1f %%" % (nm, out[i] / n, 100.0 * out[i] / tot))
    a, nn, _, _, _ = eng.root_stats(0, ply & 1)
    mv = int(a[np.argmax(nn)])
    eng.advance(list(range(G)), [mv] * G)
    eng.do_move(list(range(G)), [mv] * G)
eng.close()
