"""TEST / MEASUREMENT INFRASTRUCTURE - NOT PRODUCT CODE.

Times the reference's own Python self-play search (/root/reference, read-only; this container only)
on the headline workload - Go 19x19, MCTSearch._calc_move from the empty board - and writes the result
as a data fixture, tests/golden/cpu_baseline_reference.json, that bench.py quotes beside the number it
measures with the C port on the GPU box (the reference itself cannot travel there).

Two evaluators, both honouring the reference's evaluator contract (eval(state) -> ([(action, prob)], value)):
  * "uniform"  - no network at all: isolates rules + search + feature cost of the reference
  * "torch_fp32" - the fp32 PyTorch-CPU restatement of the dual-head ResNet (TensorFlow is absent),
                   one leaf per call, as a single reference game process sees its evaluator

    python oracle/ref_cpu_baseline.py [--sims 200] [--net-sims 40]
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np
import yaml

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

np.int = int
np.float = float
_yl = yaml.load
yaml.load = lambda f, Loader=yaml.SafeLoader: _yl(f, Loader=Loader)

import AlphaZero.env.go as go  # noqa: E402
import AlphaZero.processing.state_converter as sc  # noqa: E402
import AlphaZero.search.mcts as mcts  # noqa: E402


def ttt(args):
    """Whole tic-tac-toe games through the reference's Game class (gameplay.py:38-71), 100 sims per move, a 2-block
    256-channel fp32 PyTorch-CPU network: the reference's own CPU-runnable configuration."""
    import torch
    import AlphaZero.env.mnk as mnk
    import AlphaZero.game.gameplay as gameplay
    from alphazero_amd.network.model import random_net
    from oracle import tower_ref
    gameplay.printlog = lambda *a, **k: None
    with open(os.path.join(REF, "AlphaZero", "config", "mnk.yaml")) as f:
        cfg = yaml.load(f)
    cfg.update(board_height=3, board_width=3, k=3, flat_move_output=9, transform_types=0)
    mnk.game_config.update(board_height=3, board_width=3, k=3)
    s2t, t2a = sc.StateTensorConverter(cfg), sc.TensorActionConverter(cfg)
    net = random_net(3, 3, 9, 2, 256, seed=0)
    sd = net.state_dict()

    class Ev:
        def eval(self, state):
            x = torch.from_numpy(np.ascontiguousarray(s2t.state_to_tensor(state))).float()
            p, v = tower_ref.forward(sd, x, 2)
            return t2a.tensor_to_action(p[0].numpy()), np.float32(v[0])
    ext = {"player": {"max_playout": 100}, "dirichlet_before": 30, "log_iter": 10 ** 9, "max_turn": 400}
    np.random.seed(0)
    t0 = time.perf_counter()
    games = plies = 0
    while time.perf_counter() - t0 < 20.0:
        g = gameplay.Game(Ev(), Ev(), cfg, ext)
        g.start()
        plies += len(g.get_history()[1])
        games += 1
    dt = time.perf_counter() - t0
    out = {"workload": "mnk 3x3x3 self-play through gameplay.Game, 100 sims/move, 2-block 256-ch fp32 net on CPU",
           "games": games, "plies": plies, "seconds": dt, "moves_per_sec": plies / dt, "sims_per_sec": plies * 101 / dt,
           "torch_threads": torch.get_num_threads(), "processes": 1}
    print(out)
    path = os.path.join(HERE, "..", "tests", "golden", "cpu_baseline_reference_ttt.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sims", type=int, default=200)
    ap.add_argument("--net-sims", type=int, default=40)
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--ttt", action="store_true", help="the tic-tac-toe configuration (mnk 3x3x3, 100 sims, 2 blocks) instead of Go 19x19")
    args = ap.parse_args()
    if args.ttt:
        return ttt(args)
    import torch
    from alphazero_amd.network.model import random_net
    from oracle import tower_ref

    with open(os.path.join(REF, "AlphaZero", "config", "go.yaml")) as f:
        cfg = yaml.load(f)
    H = W = 19
    O = H * W + 1
    s2t = sc.StateTensorConverter(cfg)
    t2a = sc.TensorActionConverter(cfg)

    def uniform_eval(state):
        s2t.state_to_tensor(state)  # the reference's evaluator always converts the state
        p = np.full(O, 1.0 / O, dtype=np.float32)
        return t2a.tensor_to_action(p), np.float32(0.0)

    net = random_net(H, W, O, args.blocks, 256, seed=0)
    sd = net.state_dict()
    threads = torch.get_num_threads()

    def torch_eval(state):
        x = torch.from_numpy(np.ascontiguousarray(s2t.state_to_tensor(state))).float()
        p, v = tower_ref.forward(sd, x, args.blocks)
        return t2a.tensor_to_action(p[0].numpy()), np.float32(v[0])

    out = {"workload": "Go 19x19, MCTSearch._calc_move from the empty board, Dirichlet noise on, 8 transforms",
           "host": {"cpu": platform.processor() or platform.machine(), "cores_visible": os.cpu_count(), "torch_threads": threads},
           "measured_with": "oracle/ref_cpu_baseline.py (reference Python search, NumPy %s)" % np.__version__}
    for name, ev, n in (("uniform", uniform_eval, args.sims), ("torch_fp32", torch_eval, args.net_sims)):
        np.random.seed(0)
        search = mcts.MCTSearch(ev, cfg, max_playout=n)
        st = go.GameState(size=19)
        t0 = time.perf_counter()
        search.calc_move_with_probs(st, dirichlet=True)
        dt = time.perf_counter() - t0
        sims = n + 1
        out[name] = {"sims": sims, "seconds": dt, "sims_per_sec": sims / dt, "moves_per_sec_at_800": sims / dt / 801.0,
                     "processes": 1}
        print(name, out[name])
    # the C port (oracle/liboracle.so) on the same host under the same two evaluators: its ratio to the reference
    # converts a port timing taken elsewhere (the GPU box's host) into a reference-equivalent estimate
    from oracle import oracle as oz

    def port_torch(planes, O_):
        x = torch.from_numpy(np.ascontiguousarray(planes)).float()[None]
        p, v = tower_ref.forward(sd, x, args.blocks)
        return p[0].numpy(), float(v[0])

    for name, ev, n in (("uniform", oz.fn_ptr("oz_uniform_eval"), 20 * args.sims), ("torch_fp32", oz.make_eval(port_torch), args.net_sims)):
        tree = oz.Tree(O, 8)
        st = oz.State("go", H, W)
        t0 = time.perf_counter()
        evals = tree.search(st, n, True, ev, oz.Rng(seed=1))
        dt = time.perf_counter() - t0
        out["port_" + name] = {"sims": int(evals), "seconds": dt, "sims_per_sec": evals / dt}
        out["reference_over_port_" + name] = out[name]["sims_per_sec"] / (evals / dt)
        print("port", name, out["port_" + name], "reference/port", out["reference_over_port_" + name])
    path = os.path.join(HERE, "..", "tests", "golden", "cpu_baseline_reference.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
