import sys
import numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from alphazero_amd.engine import Engine
from alphazero_amd.network.model import fold_to_blob, random_net
H = 19; O = 362; NB = 20
eng = Engine("go", H, H, n_slots=4, num_blocks=NB, channels=256, tower_dtype="bf16")
eng.tower_load(fold_to_blob(random_net(H, H, O, NB, 256, seed=0)))
x = (np.random.RandomState(0).rand(1, 17, H, H) < 0.3).astype(np.int8)
for _ in range(20):
    eng.tower_forward(x)
eng.close()
