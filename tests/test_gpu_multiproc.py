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
