"""TEST / MEASUREMENT INFRASTRUCTURE - NOT PRODUCT CODE.

CPU self-play baseline with the C oracle (oracle/az_oracle.c, the bit-exact restatement of the reference's rules and
sequential search) driving the fp32 PyTorch-CPU restatement of the network one leaf per evaluation - what one game process
of the reference sees (train/parallel/selfplay.py:71-89: one process per game, config/rl_sys_config.yaml:27 num_worker).
Measured as P single-thread game processes side by side (the reference's shape: many game processes per host) and as one
process with all threads given to the network.  Used by bench.py's cpu_baseline leg (runs on the GPU box's host cores) and
by oracle/ref_cpu_baseline.py (build container, beside the reference itself)."""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONFIGS = {
    # name: game, H, k, sims, blocks, channels, transform_types
    "ttt": ("mnk", 3, 3, 100, 2, 256, 0),
    "reversi": ("reversi", 8, 0, 400, 6, 256, 8),
    "go9": ("go", 9, 0, 400, 10, 256, 8),
    "gomoku": ("mnk", 15, 5, 800, 10, 256, 8),
    "go19": ("go", 19, 0, 800, 20, 256, 8),
}


def play_for(game, H, k, sims, blocks, channels, tt, budget_s, threads, evaluator="torch_fp32", seed=0):
    """Self-play games with the C oracle for about budget_s seconds: searches of `sims` playouts, move chosen from the
    visit counts, both trees re-rooted, new game when one ends.  Returns (evaluations, plies, seconds)."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    torch.set_num_threads(threads)
    from alphazero_amd.network.model import random_net
    from oracle import oracle as oz
    from oracle import tower_ref
    O = H * H + (0 if game == "mnk" else 1)
    if evaluator == "uniform":
        cb = oz.fn_ptr("oz_uniform_eval")
    else:
        sd = random_net(H, H, O, blocks, channels, seed=0).state_dict()

        def ev(planes, O_):
            x = torch.from_numpy(np.ascontiguousarray(planes)).float()[None]
            p, v = tower_ref.forward(sd, x, blocks)
            return p[0].numpy(), float(v[0])
        cb = oz.make_eval(ev)
    rng = oz.Rng(seed=seed + 1)
    evals = plies = 0
    t0 = time.perf_counter()
    # a search is not interruptible: cap its length so that a slow configuration still returns near the budget
    while time.perf_counter() - t0 < budget_s:
        st = oz.State(game, H, H, k) if game == "mnk" else oz.State(game, H, H)
        trees = [oz.Tree(O, tt), oz.Tree(O, tt)]
        cur = 0
        while not (st.s.is_end or st.s.turns > 400) and time.perf_counter() - t0 < budget_s:
            left = sims
            while left > 0 and time.perf_counter() - t0 < budget_s:   # one search, in slices of at most 25 playouts
                n = min(left, 25)
                if left == sims:
                    evals += trees[cur].search(st, n, st.s.turns <= 30, cb, rng)
                else:
                    for _ in range(n):
                        trees[cur].playout(st, cb, rng)
                    evals += n
                left -= n
            if left > 0:
                break
            mv = trees[cur].choose(float(np.random.RandomState(plies + seed).rand()))
            st.do_move(mv)
            trees[0].update_with_move(mv); trees[1].update_with_move(mv)
            cur ^= 1
            plies += 1
    return evals, plies, time.perf_counter() - t0


def _worker(args):
    return play_for(*args)


def measure(config, budget_s, processes, evaluator="torch_fp32"):
    """{'processes': P single-thread game processes ..., 'single': one process with P threads}: sims/s and moves/s."""
    game, H, k, sims, blocks, channels, tt = CONFIGS[config] if isinstance(config, str) else config
    out = {"sims_per_move": sims, "evals_per_move": sims + 1}
    ctx = mp.get_context("spawn")
    with ctx.Pool(processes) as pool:
        res = pool.map(_worker, [(game, H, k, sims, blocks, channels, tt, budget_s, 1, evaluator, 17 * i) for i in range(processes)])
    ev = sum(r[0] for r in res); wall = max(r[2] for r in res)
    out["processes"] = {"n": processes, "threads_each": 1, "evals": ev, "seconds": wall, "sims_per_sec": ev / wall,
                        "moves_per_sec": ev / wall / (sims + 1)}
    with ctx.Pool(1) as pool:
        e1, p1, w1 = pool.map(_worker, [(game, H, k, sims, blocks, channels, tt, budget_s, processes, evaluator, 0)])[0]
    out["single"] = {"n": 1, "threads": processes, "evals": e1, "seconds": w1, "sims_per_sec": e1 / w1, "moves_per_sec": e1 / w1 / (sims + 1)}
    return out
