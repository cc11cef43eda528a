"""Two self-play ranks (two processes on the one GPU, gloo) through the model-update and trajectory-gather sequence."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_model_update_and_gather_with_two_ranks():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_model_update.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=100)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, err[-2000:]
        outs.append(json.loads(out.strip().splitlines()[-1]))
    res = {o["rank"]: o for o in outs}
    for rank in (0, 1):
        r = res[rank]
        assert r["mid_game"] and r["used"] == {"a": False, "b": True} and r["finished"] == 24 and r["same_local"], r
        assert r["weights_equal"] and r["gather_ok"] and r["packed_bytes"] < r["dense_bytes"], r
    assert res[0]["n_received"] == 48 and res[1]["n_received"] == 0


def test_plain_bench_command_with_two_gpus_runs_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (here both on the one card, gloo in
    place of RCCL, which refuses two ranks per device), prints one JSON line with n_gpus 2, and runs both exchanges -
    weights to every rank's tower from GPU memory, packed trajectories to rank 0 - after the timed region."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(AZ_BENCH_DEVICE="0", AZ_BENCH_DIST_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "25", "--warmup", "2", "--board", "9", "--sims", "19",
           "--blocks", "2", "--games", "96", "--exchange-games", "8", "--no-cpu-baseline", "--no-telemetry", "--no-tower-parity"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [x for x in r.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["games_per_gpu"] == 96 and d["value"] > 0
    assert d["moves_completed_in_timed_region"] >= 2 * 96          # both ranks' games are in the total
    ex = d["exchange"]
    assert "error" not in ex, ex
    assert ex["world"] == 2 and ex["weights"]["bytes"] > 0 and ex["trajectories"]["received_ok"]
    # rank 1's packed games arrived at rank 0 (its random games pack to about, not exactly, rank 0's size)
    assert 0.9 < ex["trajectories"]["bytes_into_rank0"] / ex["trajectories"]["packed_bytes_per_rank"] < 1.1
    # (40 of a 9x9 board's 82 actions visited per ply: packed is about half of dense here, a tenth on 19x19)
    assert ex["trajectories"]["packed_bytes_per_rank"] < ex["trajectories"]["dense_bytes_per_rank"]


_NCCL_ONE_RANK = """
import os, sys, json
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
# the tensor types the two exchanges hand to RCCL (alphazero_amd/parallel.py): float32 broadcast, int64 all_gather of the
# sizes, and the packed blob as uint8 (RCCL has no 16-bit integer type: int16 must be refused, uint8 taken)
t = torch.arange(1000, dtype=torch.float32, device=dev); dist.broadcast(t, src=0)
n = torch.tensor([5], dtype=torch.int64, device=dev); out = [torch.zeros(1, dtype=torch.int64, device=dev)]; dist.all_gather(out, n)
b = torch.arange(100, dtype=torch.int16, device=dev)
u8 = b.view(torch.uint8); g8 = [torch.zeros_like(u8)]; dist.all_gather(g8, u8)
try:
    g16 = [torch.zeros_like(b)]; dist.all_gather(g16, b); torch.cuda.synchronize(); int16_ok = True
except Exception as e:
    int16_ok = False
torch.cuda.synchronize()
print(json.dumps({"bcast": bool((t.cpu().numpy() == np.arange(1000)).all()), "sizes": int(out[0].item()),
                  "u8": bool(torch.equal(g8[0].view(torch.int16), b)), "int16_ok": int16_ok}))
dist.destroy_process_group()
"""


def test_rccl_takes_the_tensor_types_of_the_exchanges():
    """One rank on the real RCCL backend: the dtypes parallel.py puts on the wire are accepted (float32, int64, uint8)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, "-c", _NCCL_ONE_RANK % (ROOT, str(port))], env=env, capture_output=True, text=True, timeout=150)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["bcast"] and d["sizes"] == 5 and d["u8"], d
