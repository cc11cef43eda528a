#!/usr/bin/env python3
"""Self-play throughput of the HIP engine on the headline configuration:
Go 19x19, 800 sims/move, 20-block 256-channel dual-head ResNet (bf16 MFMA tower), games sharded
over the GPUs of one node with the weights broadcast over RCCL.

    python bench.py --gpus N --steps K --warmup W

A "round" is one pass of the hot path over every game slot of a GPU: PUCT select (rule replay down to
one leaf per game) -> feature planes -> tower forward over the leaf batch -> expand/backup ->
(when a search is complete) move choice, real move, re-root of both trees; it consumes one network
evaluation (= one MCTS simulation) per game.  A "step" is a fixed block of rounds, one twentieth of a
move's search rounded up (41 rounds at 800 sims/move), so that the default 20 timed steps carry every
game through the end of a search, the move and the re-root of both trees: `value` is the number of
moves completed inside the timed region per second.  Inputs are synthetic: seeded glorot-uniform
weights, games start from the empty board, device RNG seed 0; everything is resident in HBM before
the timed region.  Rank 0 prints one JSON line.

`--gpus N` with N > 1 and no RANK in the environment (the plain `python bench.py --gpus N` form) makes this process a
launcher: before anything touches the GPU it starts N ranks of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one
GPU each, RCCL between them), lets rank 0 print the line and returns non-zero if any rank fails.  Under torchrun the
ranks exist already and run directly.  After the timed region (outside it) every run also exercises the two exchanges of
a multi-GPU self-play job - the weight update and the gather of packed trajectories to rank 0 - and reports their bytes
and times in `exchange`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS_BF16 = 2500.0  # dense bf16 MFMA peak of one MI355X, TFLOP/s
VALU_PEAK_TFLOPS_FP32 = 157.3   # float32 vector peak (the fp32 parity tower runs on the vector ALU)


def measured_traffic(games, board, channels):
    """HBM bytes per launch of the dominant conv kernel from the committed PMC passes (they cannot be
    collected inside a timed run); None when the workload is not the one the counters were taken on."""
    d = None
    for name in ("round3_traffic.json", "round2_traffic.json"):   # the newest committed counter passes
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            break
        except (OSError, ValueError):
            continue
    if d is None:
        return None, None
    if (d.get("board"), d.get("channels")) != (board, channels):
        return None, None
    # per-launch traffic is proportional to the number of boards of the launch (activations in/out; the weights, 1.2 MB, are
    # 0.6 % of a 363-board launch)
    return (int(d["traffic_bytes_per_launch"] * games / d["boards"]),
            d["source"] + "; measured at %.0f boards per launch, scaled to %.0f" % (d["boards"], games))


class PowerSampler:
    """Reads the GPU's own telemetry (rocm-smi --showmetrics, read-only) a few times while the timed region runs, from a
    thread of its own: the timed call blocks in the engine with the GIL released.  What limits the conv kernels is the
    package power limit, so the line carries socket power, the XCDs' shader clocks and the power-limit residency counter
    beside the roofline figure.  Never fails the bench: no tool / no permission / odd output -> None."""

    KEYS = {"current_socket_power": "socket_power_w", "current_gfxclks": "gfxclk_mhz", "ppt_residency_acc": "power_limit_residency",
            "socket_thm_residency_acc": "thermal_residency", "average_umc_activity": "memory_controller_activity_pct",
            "temperature_hotspot": "hotspot_c"}

    def __init__(self, device, delays=(6.0, 14.0, 22.0)):
        import threading
        self.device, self.delays, self.samples, self.stop = device, delays, [], threading.Event()
        self.thread = threading.Thread(target=self._run, daemon=True)

    SMI = "/opt/rocm/libexec/rocm_smi/rocm_smi.py"

    @classmethod
    def _smi(cls, *args):
        """rocm-smi started as `python3 rocm_smi.py` (its own first line goes through `env`, one more exec), with a
        profiler's preload and tool variables removed from the child's environment."""
        import subprocess
        env = {k: v for k, v in os.environ.items() if not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS"))}
        if "rocprof" in env.get("LD_PRELOAD", ""):   # only the profiler's own entries go; anything else preloaded stays
            env["LD_PRELOAD"] = " ".join(x for x in env["LD_PRELOAD"].replace(":", " ").split() if "rocprof" not in x)
        cmd = [sys.executable, cls.SMI] if os.path.exists(cls.SMI) else ["rocm-smi"]
        return subprocess.run(cmd + list(args), capture_output=True, text=True, timeout=20, env=env).stdout

    @staticmethod
    def under_profiler():
        e = os.environ
        return any(k.startswith(("ROCP_", "ROCPROF")) for k in e) or "rocprof" in e.get("LD_PRELOAD", "") or "HSA_TOOLS_LIB" in e

    def _read(self):
        return self.parse(self._smi("-d", str(self.device), "--showmetrics", "--json"))

    @classmethod
    def parse(cls, text):
        """The fields of KEYS out of `rocm-smi --showmetrics --json` output (one card; warnings may precede the JSON)."""
        doc = json.loads(text[text.index("{"):])
        card = next(iter(doc.values()))
        rec = {}
        for k, v in card.items():
            for pre, name in cls.KEYS.items():
                if k.startswith(pre):
                    try:
                        if isinstance(v, str) and v.strip().startswith("["):
                            v = json.loads(v.replace("'", '"'))
                        rec[name] = [float(x) for x in v] if isinstance(v, (list, tuple)) else float(v)
                    except (TypeError, ValueError):
                        pass
        return rec

    def _run(self):
        t0 = time.perf_counter()
        for d in self.delays:
            if self.stop.wait(max(0.0, d - (time.perf_counter() - t0))):
                return
            try:
                rec = self._read()
                rec["t_s"] = round(time.perf_counter() - t0, 1)
                self.samples.append(rec)
            except Exception:  # noqa: BLE001 - telemetry is optional
                return

    def start(self):
        self.thread.start()

    def finish(self):
        self.stop.set()
        self.thread.join(timeout=25)
        if not self.samples:
            return None
        out = {"source": "rocm-smi --showmetrics sampled during the timed region", "samples": self.samples}
        try:
            lim = self._smi("-d", str(self.device), "--showmaxpower", "--json")
            card = next(iter(json.loads(lim[lim.index("{"):]).values()))
            out["power_limit_w"] = float(next(iter(card.values())))
        except Exception:  # noqa: BLE001
            pass
        return out


def cpu_baseline(game, H, k, num_blocks, channels, sims, budget_s=12.0):
    """The CPU oracle (C port of the reference's rules/search) driving an fp32 PyTorch-CPU restatement of the network,
    one leaf per evaluation as the reference's per-game search sees it.  Two shapes, a bounded sample each: P single-thread
    game processes side by side (how the reference runs self-play: one process per game, num_worker of them), which is
    `value`, and one process with all P threads in the network.  The reference's own Python numbers for the same shapes,
    taken in the build container by oracle/ref_cpu_baseline.py, travel as a data fixture and are quoted beside them."""
    from oracle import port_baseline

    # the GPU box gives a one-GPU job 16 host cores (os.cpu_count() reports the whole machine)
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    cfgt = (game, H, k, sims, num_blocks, channels, 8)
    m = port_baseline.measure(cfgt, budget_s, cores)
    ref = None
    try:
        with open(os.path.join(ROOT, "tests", "golden", "cpu_baseline_reference.json")) as f:
            ref = json.load(f)
    except OSError:
        pass
    name = {tuple(v[:5]): n for n, v in port_baseline.CONFIGS.items()}.get((game, H, k if game == "mnk" else 0, sims, num_blocks))
    rec = (ref or {}).get("configs", {}).get(name) if name else None
    est = None
    if rec:
        # port timing on this host x (reference / port) ratio measured side by side, same shape, in the build container
        est = m["processes"]["sims_per_sec"] * rec["torch_fp32"]["reference_over_port_processes"]
    return {
        "value": m["processes"]["moves_per_sec"], "unit": "moves/s", "sims_per_sec": m["processes"]["sims_per_sec"], "cores": cores,
        "kind": "port",
        "sample": "%d single-thread game processes for %.0f s each on %s %dx%d, %d sims/move (%d-block %d-ch fp32 net on CPU, one leaf "
                  "per evaluation, searches from the opening); moves/s = sims/s / %d" % (cores, budget_s, game, H, H, sims, num_blocks,
                                                                                       channels, sims + 1),
        "single_process_all_threads": m["single"],
        "reference_equivalent_sims_per_sec_estimate": est,
        "reference_python_in_build_container": {"host": (ref or {}).get("host"), "config": name, "torch_fp32": rec["torch_fp32"] if rec else None,
                                                "uniform_evaluator": rec["uniform"] if rec else None} if ref else None,
    }


def hbm_kernels(c, cells, O, steps, peak_gbs=8000.0):
    """Achieved GB/s of the search-side phases against the HBM peak.  Algorithmic bytes: per simulation the 3 KiB
    state ring read, 12 B per child scanned (N, W, P) + 8 B per level, 17 int8 planes + the legal mask written
    (select); the priors row and mask read, 18 B per edge created, 8 B per level backed up (finish)."""
    ev = max(c["evals"], 1)
    sel_bytes = ev * (8 * 384 + 17 * cells + O) + 12 * c["children_scanned"] + 8 * c["levels"]
    fin_bytes = ev * (4 * O + O) + 18 * c["edges_created"] + 8 * c["levels"]
    out = []
    for name, nbytes, ms in (("k_select", sel_bytes, c["select_ms"]), ("k_finish+k_move+k_reroot", fin_bytes, c["finish_ms"])):
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out.append({"kernel": name, "bound": "hbm", "achieved": gbs, "peak": peak_gbs, "unit": "GB/s", "frac": gbs / peak_gbs,
                    "bytes_per_sim": nbytes / ev, "ms_per_round": ms / steps})
    return out


class PostRegionGuard:
    """The measured line is complete when the timed region ends; the exercises that run after it (exchanges between ranks,
    parity check, other tower format, CPU baseline) only add blocks to it.  None of them may cost the measurement: if they
    have not finished after `seconds` (a collective that never completes on some fabric), rank 0 prints the line as it
    stands, saying so in `post_region_note`, and the process leaves - every rank runs its own guard, so all of them do."""

    def __init__(self, out, rank, seconds, emit=None, leave=os._exit):
        import threading
        self.out, self.rank, self.seconds = out, rank, seconds
        self._emit_fn = emit if emit is not None else _emit
        self._leave = leave
        self._lock = threading.Lock()
        self.emitted = False
        self._timer = threading.Timer(seconds, self._expired)
        self._timer.daemon = True
        self._timer.start()

    def emit(self, note=None):
        with self._lock:
            if self.emitted:
                return
            self.emitted = True
            if self.rank == 0:
                if note:
                    self.out["post_region_note"] = note
                self._emit_fn(json.dumps(self.out))

    def _expired(self):
        self.emit("the work after the timed region did not finish within %.0f s and was abandoned" % self.seconds)
        self._leave(0)

    def record(self, key, fn):
        """Runs one exercise; its result (or the error it raised) becomes out[key] unless the line has gone out already."""
        try:
            val = fn()
        except Exception as e:  # noqa: BLE001 - a failed exercise is reported in its place
            val = {"error": "%s: %s" % (type(e).__name__, e)}
        if self.rank == 0:
            with self._lock:
                if not self.emitted:
                    self.out[key] = val

    def done(self):
        self._timer.cancel()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rounds-per-step", type=int, default=0,
                    help="rounds (one simulation per game each) in a step; 0 = ceil((sims + 1) / 20), i.e. 20 steps are a "
                         "little more than one whole search per game (about 45 s on one MI355X at the default workload)")
    ap.add_argument("--profile-every", type=int, default=16,
                    help="per-launch HIP-event timing of the tower's conv launches on every N-th round of the timed region")
    ap.add_argument("--games", type=int, default=0,
                    help="concurrent games per GPU; 0 = 4096, or fewer if the GPU's free memory does not hold their trees")
    ap.add_argument("--game", default="go", choices=["go", "reversi", "mnk"],
                    help="rule set; the headline workload is go (the other games are the smaller configurations)")
    ap.add_argument("--k", type=int, default=5, help="stones in a row for mnk")
    ap.add_argument("--board", type=int, default=19)
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--blocks", type=int, default=20, help="residual blocks after the stem")
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "fp32"],
                    help="storage type of the tower's weights and activations: bf16 (the headline), f16 (IEEE half: same matrix-core "
                         "rate, 8 times finer rounding), fp32 (vector-ALU parity tower)")
    ap.add_argument("--tree-pool", type=float, default=1.0 / 3.0,
                    help="edges in the pool the search trees share, as a fraction of games x 2 trees x the per-tree limit of "
                         "3 * (sims + 1) * actions (the fullest tree of a search from the opening reaches a third of its limit, "
                         "and only one of a game's two trees is being searched)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-telemetry", action="store_true", help="no rocm-smi sampling during the timed region (always off under a profiler)")
    ap.add_argument("--no-exchange", action="store_true", help="skip the weight-update / trajectory-gather exercise after the timed region")
    ap.add_argument("--no-tower-parity", action="store_true", help="skip the bf16-against-fp32 tower comparison after the timed region")
    ap.add_argument("--no-other-format", action="store_true", help="skip the short run with the other 16-bit tower format after the timed region")
    ap.add_argument("--post-timeout", type=float, default=300.0,
                    help="seconds the work after the timed region (exchanges, parity check, CPU baseline) may take before the "
                         "measured line is printed without it and the ranks leave")
    ap.add_argument("--exchange-games", type=int, default=1024, help="synthetic finished games (400 plies each) every rank packs and sends")
    return ap.parse_args(argv)


def launch_ranks(args, argv, script=None):
    """`python bench.py --gpus N` without a launcher around it: start N ranks of this script, one per GPU.  This process never
    touches the GPU (no torch.cuda call, no engine): it only starts children, and a child is a fresh interpreter that finds
    RANK set and runs the bench itself.  Rank 0 inherits stdout, so its one JSON line is this command's output; the other
    ranks' stdout goes to stderr.  Returns the first non-zero exit status of any rank (after ending the others), else 0."""
    import socket
    import subprocess
    n = args.gpus
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []

    def die_with_parent():
        # (runs in the child between fork and exec) a rank must not outlive a launcher that was killed: it would keep its GPU
        try:
            import ctypes
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, 9)   # PR_SET_PDEATHSIG, SIGKILL
        except Exception:  # noqa: BLE001 - not Linux / no libc: the signal handlers below still cover an orderly end
            pass

    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr, preexec_fn=die_with_parent))

    def ended(signum, frame):   # the launcher is told to stop: so are its ranks
        for p in procs:
            if p.poll() is None:
                p.terminate()
        sys.exit(128 + signum)

    import signal
    for sig in (signal.SIGTERM, signal.SIGINT):
        try:
            signal.signal(sig, ended)
        except ValueError:      # not the main thread (tests call this function directly)
            pass
    status = 0
    live = set(range(n))
    while live:
        for r in list(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 128 - rc
                print("bench.py: rank %d exited with status %d; ending the other ranks" % (r, rc), file=sys.stderr)
                for q in live:           # they would wait for the lost rank in the next collective
                    procs[q].terminate()
        time.sleep(0.2)
    return status


def exchange_exercise(eng, parallel, torch, dist, rank, world, dev, blob_t, O, n_games, backend):
    """The two exchanges of a multi-GPU self-play job, once each, outside the timed region (a generation's worth):
    the model update (rank 0's folded weights to every rank's tower, device memory to device memory) and the trajectory
    gather (every rank's finished games, packed sparse, to rank 0 where the replay buffer lives).  The games are synthetic
    and shaped like finished ones: 400 plies, a few dozen visited children per ply."""
    out = {"backend": backend if dist is not None else "none (one rank)", "world": world}

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # 1. model update
    sync()
    times = []
    for _ in range(2):
        t0 = time.perf_counter()
        parallel.update_weights(eng, blob_t if rank == 0 else None, dist, dev)
        sync()
        times.append((time.perf_counter() - t0) * 1e3)
    out["weights"] = {"bytes": int(eng.tower_blob_floats()) * 4, "ms_first": times[0], "ms": times[1],
                      "path": "broadcast of a GPU tensor, then az_tower_load_weights_dev (no host copy)"}
    # 2. trajectories
    plies = 400
    rows = n_games * plies
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    visits = torch.zeros((rows, O), dtype=torch.int32, device=dev)
    mask = torch.rand((rows, O), device=dev, generator=g) < (40.0 / O)
    visits[mask] = torch.randint(1, 200, (int(mask.sum()),), device=dev, generator=g, dtype=torch.int32)
    del mask
    moves = torch.randint(0, O, (rows,), device=dev, generator=g, dtype=torch.int32).to(torch.int16)
    sync()
    t0 = time.perf_counter()
    packed = parallel.pack_sparse(moves, visits, [plies] * n_games, [1] * n_games)
    torch.cuda.synchronize()
    t_pack = (time.perf_counter() - t0) * 1e3
    dense = rows * (O * 4 + 2)
    del visits
    sync()
    t0 = time.perf_counter()
    got = parallel.gather_packed(packed, dist, dst=0)
    sync()
    t_gather = (time.perf_counter() - t0) * 1e3
    ok = True
    if rank == 0:
        ok = len(got) == world
        for b in got:
            head = b[:16].view(torch.int32).tolist()
            ok = ok and head[0] == parallel._MAGIC and head[1] == n_games and head[2] == rows and b.device.type == "cuda"
    recv = sum(parallel.packed_nbytes(b) for b in got[1:]) if rank == 0 else 0
    out["trajectories"] = {"games_per_rank": n_games, "plies_per_game": plies, "dense_bytes_per_rank": dense,
                           "packed_bytes_per_rank": parallel.packed_nbytes(packed), "pack_ms": t_pack, "gather_ms": t_gather,
                           "bytes_into_rank0": recv, "gather_gb_per_s": (recv / (t_gather * 1e-3) / 1e9) if t_gather > 0 and recv else None,
                           "received_ok": bool(ok), "wire": "uint8 view of the packed int16 tensor, point to point, 64 MiB chunks"}
    return out


def other_16bit_format(args, Engine, fold_to_blob, random_net, H, O, device, rps):
    """The same workload for a few steps with the other 16-bit format of the tower (half precision when the run is bf16 and the
    other way round): the bf16 tower is the faster one, the f16 tower the one whose value output stays within 1e-2 of
    float32 at 20 blocks (`tower_parity`), and this says on the same card, minutes apart, what that costs."""
    other = "f16" if args.dtype == "bf16" else "bf16"
    eng = Engine(args.game, H, H, args.k, n_slots=args.games, max_playout=args.sims, transform_types=8, dirichlet_before=30,
                 max_turn=400, d_alpha=0.03, d_epsilon=0.25, seed=1000, num_blocks=args.blocks, channels=args.channels,
                 tower_dtype=other, device=device, tree_pool_edges=max(1, int(args.tree_pool * 2 * args.games * 3 * (args.sims + 1) * O)))
    try:
        eng.tower_load(fold_to_blob(random_net(H, H, O, args.blocks, args.channels, seed=0)))
        eng.selfplay_start()
        eng.tower_profile(8)
        eng.selfplay_run(rps)
        eng.counters_reset()
        steps = 3
        t0 = time.perf_counter()
        eng.selfplay_run(steps * rps)
        eng.sync()
        dt = time.perf_counter() - t0
        c = eng.counters()
        conv_ms = c["conv_ms"] / max(c["conv_launches"], 1)
        return {"dtype": other, "steps": steps, "rounds": steps * rps, "sims_per_sec": c["evals"] / dt, "ms_per_round": dt * 1e3 / (steps * rps),
                "steady_state_moves_per_sec": c["evals"] / dt / (args.sims + 1),
                "conv_tflops": c["conv_flops_per_launch"] / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0, "kernel": eng.tower_conv_kernel()}
    finally:
        eng.close()


def tower_parity(torch, np, eng, H, O, blocks, channels, planes, device):
    """The 16-bit towers (bf16 and half precision) against the fp32 tower of the same library on the same boards and weights:
    the bench's architecture with batch-norm statistics calibrated on data (as a trained network's are; the timed run itself
    uses untouched statistics, with which a random 20-block tower has logits of magnitude 50 and no tolerance means
    anything).  96 boards, so the 16-bit side runs the tiled kernels of large batches."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.network.model import calibrate_bn, fold_to_blob, random_net
    B = min(96, planes.shape[0])
    x = np.ascontiguousarray(planes[:B])
    net = random_net(H, H, O, blocks, channels, seed=0)
    # statistics from random boards of mixed occupancy (what tools/bf16_search_effect.py and the depth tests use); this run's
    # leaf boards are all a few plies from the empty board and next to identical, which makes degenerate statistics
    calibrate_bn(net, torch.from_numpy((np.random.RandomState(7).rand(6, 17, H, H) < 0.3).astype(np.float32)))
    blob = fold_to_blob(net)
    res = {}
    for dt in ("bf16", "f16", "fp32"):
        e = Engine(eng.game, H, H, eng.cfg.k, n_slots=B, num_blocks=blocks, channels=channels, tower_dtype=dt, device=device)
        e.tower_load(blob)
        res[dt] = e.tower_forward(x)
        if dt == "bf16":
            res["kernel"] = e.tower_conv_kernel()
        e.close()
    out = {"what": "16-bit towers against the fp32 tower (same library, same weights, batch-norm statistics calibrated) on %d leaf boards of this run" % B,
           "kernel": res["kernel"], "bound_16bit": 1e-2}
    for dt in ("bf16", "f16"):
        dp = np.abs(res[dt][0] - res["fp32"][0]); dv = np.abs(res[dt][1] - res["fp32"][1])
        out[dt] = {"priors_max_abs_diff": float(dp.max()), "value_max_abs_diff": float(dv.max()), "value_mean_abs_diff": float(dv.mean()),
                   "priors_within_bound": bool(dp.max() <= 1e-2), "value_within_bound": bool(dv.max() <= 1e-2)}
    try:
        with open(os.path.join(ROOT, "profiles", "round3_bf16_search_effect.json")) as f:
            out["search_effect"] = json.load(f)
    except (OSError, ValueError):
        out["search_effect"] = None
    return out


def main(args):
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # rehearsal on a box with fewer GPUs than ranks: AZ_BENCH_DEVICE pins every rank to one card and
    # AZ_BENCH_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one device)
    if "AZ_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["AZ_BENCH_DEVICE"])
    backend = os.environ.get("AZ_BENCH_DIST_BACKEND", "nccl")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d wants GPU %d but only %d are visible (--gpus %d)" % (rank, local_rank, torch.cuda.device_count(), args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # under a launcher the RCCL path is taken even with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from alphazero_amd.engine import Engine, slots_for_whole_tile_rounds
    from alphazero_amd.network.model import fold_to_blob, random_net
    from alphazero_amd import parallel

    H = args.board
    O = H * H + (0 if args.game == "mnk" else 1)
    if args.games <= 0:
        # two trees per game: their share of the edge pool (18 B per edge) and 8*(sims+1) blocks of 24 B each, plus state,
        # trajectories and activations (about 0.7 MB); larger batches keep the tower's tiles flowing (fewer tails)
        per_game = 2 * (int(args.tree_pool * 3 * (args.sims + 1) * O) * 18 + 8 * (args.sims + 1) * 24) + (402 * O * 4 + 402 * 2) + 3 * 441 * 256 * 2 + 64 * 1024
        free, _ = torch.cuda.mem_get_info(local_rank)
        gmax = int(max(256, min(4096, int(0.85 * free) // per_game)))
        # the tower's conv tiles are 256 board points; pick the largest game count whose tiles fill a whole number of
        # rounds over the CUs (no partly idle last round in each of the 40 conv launches): 3993 games of 19x19 = 22 rounds
        cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
        args.games = slots_for_whole_tile_rounds(gmax, H, H, cus)
    eng = Engine(args.game, H, H, args.k, n_slots=args.games, max_playout=args.sims, transform_types=8, dirichlet_before=30,
                 max_turn=400, d_alpha=0.03, d_epsilon=0.25, seed=1000 + rank, record_trajectories=True,
                 num_blocks=args.blocks, channels=args.channels, tower_dtype=args.dtype, device=local_rank,
                 tree_pool_edges=max(1, int(args.tree_pool * 2 * args.games * 3 * (args.sims + 1) * O)))
    # weights: rank 0 builds them, every rank receives them over RCCL and loads them from GPU memory (model update path)
    blob_t = None
    if rank == 0:
        blob_t = torch.from_numpy(fold_to_blob(random_net(H, H, O, args.blocks, args.channels, seed=0))).to(dev)
    parallel.update_weights(eng, blob_t, dist, dev)
    eng.selfplay_start()
    rps = args.rounds_per_step if args.rounds_per_step > 0 else -(-(args.sims + 1) // 20)
    rounds, warm_rounds = args.steps * rps, args.warmup * rps
    # the headline rate is measured in the production mode (leaf compaction on, rounds queued ahead of the host); every
    # profile_every-th round additionally carries HIP events around each of its conv launches for the roofline figure
    eng.tower_profile(max(8, args.profile_every))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    eng.selfplay_run(warm_rounds)
    eng.counters_reset()
    barrier()
    sample = rank == 0 and not args.no_telemetry and not PowerSampler.under_profiler()
    sampler = PowerSampler(local_rank) if sample else None
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    eng.selfplay_run(rounds)
    barrier()
    dt = time.perf_counter() - t0
    gpu_telemetry = sampler.finish() if sampler else None
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    c = eng.counters()
    evals = torch.tensor([float(c["evals"]), float(c["moves"]), float(c["expansions_skipped"]), float(args.games)],
                         dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(evals, op=dist.ReduceOp.SUM)
    total_evals, total_moves, total_skipped, total_games = (float(x) for x in evals.tolist())
    sims_per_s = total_evals / dt
    evals_per_move = args.sims + 1  # a search costs max_playout evaluations, +1 when its root is still a leaf
    steady_moves_per_s = sims_per_s / evals_per_move
    # moves that really completed (search end, move choice, real move, both re-roots) inside the timed region; a region
    # shorter than one search completes none and only the evaluation-count figure exists
    whole_searches = total_moves >= total_games
    moves_per_s = total_moves / dt if whole_searches else steady_moves_per_s

    # the evaluator outputs the last timed round's finish step used: finite, priors a distribution (a 16-bit tower whose
    # activations overflowed would show here as inf / nan)
    lp, lv = eng.leaf_outputs()
    outputs_ok = bool(np.isfinite(lp).all() and np.isfinite(lv).all() and np.abs(lp.sum(axis=1) - 1.0).max() < 1e-3 and np.abs(lv).max() <= 1.0)
    del lp, lv

    # ---- the measured line is complete here (rank 0); what follows only adds to it
    out = None
    if rank == 0:
        conv_ms = c["conv_ms"] / max(c["conv_launches"], 1)
        achieved = c["conv_flops_per_launch"] / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        # the tower takes a large batch through its layers in groups of boards: a launch covers one group
        boards_per_launch = c["conv_flops_per_launch"] / (2.0 * H * H * 9.0 * args.channels * args.channels)
        traffic, traffic_src = measured_traffic(boards_per_launch, H, args.channels) if args.dtype == "bf16" and boards_per_launch > 0 else (None, None)
        # the 16-bit kernels run on the matrix cores (bf16 and f16 have the same dense peak); the fp32 tower (parity mode) is a
        # vector-ALU kernel
        peak, bound = (MFMA_PEAK_TFLOPS_BF16, "mfma") if args.dtype != "fp32" else (VALU_PEAK_TFLOPS_FP32, "valu")
        out = {
            "metric": "self-play moves/sec + MCTS sims/sec (node), Go 19x19 800 sims, 1/2/4/8 GPUs",
            "value": moves_per_s, "unit": "moves/s", "sims_per_sec": sims_per_s,
            "value_kind": "moves completed in the timed region / time" if whole_searches else
                          "sims/s / (sims + 1): the timed region is shorter than one search",
            "steady_state_moves_per_sec": steady_moves_per_s,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps,
            "rounds_per_step": rps, "ms_per_round": dt * 1e3 / rounds,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {
                "workload": "%s %dx%d self-play, %d sims/move, stem + %d residual blocks x %d ch dual-head ResNet, "
                            "%d concurrent games per GPU, dihedral transforms on, Dirichlet noise on the first 31 plies"
                            % ({"go": "Go", "reversi": "Reversi", "mnk": "m,n,k (k=%d)" % args.k}[args.game], H, H, args.sims,
                               args.blocks, args.channels, args.games),
                "games_per_gpu": args.games, "sims_per_move": args.sims, "evals_per_move_used": evals_per_move,
                "residual_blocks": args.blocks, "channels": args.channels, "parallelism": "games sharded, dp%d" % world,
                "tower_gflop_per_eval": eng.tower_flops_per_eval() / 1e9,
            },
            "moves_completed_in_timed_region": total_moves, "completed_moves_per_sec": total_moves / dt,
            "phase_ms_per_round": {"select": c["select_ms"] / rounds, "tower": c["tower_ms"] / rounds,
                                   "finish_move_reroot": c["finish_ms"] / rounds},
            # tree_full_policy 0 here: a full tree or an empty edge pool would have stopped the run with an error, so 0
            # skipped expansions is by construction; the peaks show how far the fullest tree and the pool of rank 0 got
            "leaf_outputs_finite_and_normalised": outputs_ok,
            "expansions_skipped": total_skipped,
            "tree_arena_peak": {"edges": c["peak_tree_edges"], "edge_cap": c["tree_edge_cap"],
                                "blocks": c["peak_tree_blocks"], "block_cap": c["tree_block_cap"],
                                "frac": max(c["peak_tree_edges"] / max(c["tree_edge_cap"], 1),
                                            c["peak_tree_blocks"] / max(c["tree_block_cap"], 1))},
            "tree_edge_pool": {"page_edges": c["tree_page_edges"], "pages": c["tree_pool_pages"], "pages_peak": c["tree_pool_pages_peak"],
                               "pages_in_use_at_end": c["tree_pool_pages_in_use"],
                               "bytes": c["tree_pool_pages"] * c["tree_page_edges"] * 18,
                               "bytes_if_every_tree_had_its_own_arena": 2 * args.games * c["tree_edge_cap"] * 18,
                               "frac_of_pool_used_at_peak": c["tree_pool_pages_peak"] / max(c["tree_pool_pages"], 1)},
            "tree_stats_per_sim": {"levels": c["levels"] / max(c["evals"], 1), "children_scanned": c["children_scanned"] / max(c["evals"], 1)},
            "roofline": {
                "bound": bound, "kernel": eng.tower_conv_kernel(),
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": conv_ms, "launches": c["conv_launches"], "flops_per_launch": c["conv_flops_per_launch"],
                "boards_per_launch": boards_per_launch,
                "timing": "HIP events on the launch stream around every conv launch of every %d-th round of the timed "
                          "region (all games evaluated in those rounds); flops_per_launch is the mean over the timed launches" % max(8, args.profile_every),
            },
            # HBM-side kernels: algorithmic bytes from the engine's counters / kernel time (HIP events around each phase)
            "hbm_kernels": hbm_kernels(c, H * H, O, rounds),
            "gpu_telemetry": gpu_telemetry,
            "tower_tflops_end_to_end": (c["tower_evals"] * eng.tower_flops_per_eval()) / (c["tower_ms"] * 1e-3) / 1e12 if c["tower_ms"] > 0 else 0.0,
            "exchange": None, "tower_parity": None, "other_16bit_format": None,
        }

    # ---- after the timed region: the exchanges of a multi-GPU job, the tower's 16-bit error on this run's boards, the CPU
    # baseline.  None of it may cost the measurement: if it has not finished after --post-timeout seconds (a collective that
    # never completes on some fabric), rank 0 prints the line as it stands, saying so, and every rank leaves.
    guard = PostRegionGuard(out, rank, args.post_timeout)
    record = guard.record

    if not args.no_exchange:
        record("exchange", lambda: exchange_exercise(eng, parallel, torch, dist, rank, world, dev, blob_t, O,
                                                     max(1, min(args.exchange_games, args.games)), backend))
    if rank == 0 and args.dtype != "fp32" and not args.no_tower_parity:
        def parity():
            eng.selfplay_select()   # a fresh round's leaves (the finish step of the last timed round consumed its own)
            planes, valid = eng.leaf_batch()
            rows = np.flatnonzero(valid)
            return tower_parity(torch, np, eng, H, O, args.blocks, args.channels, planes[rows if rows.size else slice(None)], local_rank)
        record("tower_parity", parity)
    if rank == 0 and world == 1 and args.dtype != "fp32" and not args.no_other_format:   # at N=1 only, like the CPU baseline
        record("other_16bit_format", lambda: other_16bit_format(args, Engine, fold_to_blob, random_net, H, O, local_rank, rps))
    if rank == 0 and not args.no_cpu_baseline and world == 1:  # reported at N=1 only; at N>1 the other ranks would idle behind it
        record("cpu_baseline", lambda: cpu_baseline(args.game, H, args.k, args.blocks, args.channels, args.sims))
    guard.emit()
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    guard.done()


# The one JSON line must be the only thing on stdout, but libraries write there too (RCCL prints a version banner when
# its first communicator is made): file descriptor 1 is pointed at stderr for the whole run and the result line goes
# to the saved descriptor.
_REAL_STDOUT = None


def _emit(line):
    sys.stdout.flush()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (line + "\n").encode())


if __name__ == "__main__":
    _args = parse_args()
    if _args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(_args, sys.argv[1:]))
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    main(_args)
