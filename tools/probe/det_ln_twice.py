"""Every layernorm_bwd of a deterministic-mode step is launched TWICE on the same inputs (second result discarded by the
step); at the end the two dx are compared.  MODE=sync puts a device synchronisation in front of every call.
    python tools/probe/det_ln_twice.py"""
import copy
import os
import sys
import warnings

warnings.filterwarnings("ignore")
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import torch
from peppa_amd import hip as H
from peppa_amd import layers as L
from peppa_amd.data import synthetic_batch
import test_deterministic_gpu as T

H.set_deterministic(True)
net = T._net(T._cfg())
state = copy.deepcopy(net.state_dict())
batch = synthetic_batch(2, 16, 112, 36800).to("cuda")
PAIRS = []
mode = os.environ.get("MODE", "")
orig = L.layernorm_bwd


def twice(dy, x, ln, saved):
    if mode == "sync":
        torch.cuda.synchronize()
    r = orig(dy, x, ln, saved)
    r2 = orig(dy, x, ln, saved)
    r3 = orig(dy, x, ln, saved)
    PAIRS.append((r[0], r2[0].clone(), r3[0].clone(), dy, dy.clone()))
    return r


L.layernorm_bwd = twice
for rep in range(int(os.environ.get("REPS", "16"))):
    PAIRS.clear()
    T._run(net, state, batch, steps=1)
    torch.cuda.synchronize()
    bad = [(i, not torch.equal(a, b), not torch.equal(b, c), not torch.equal(d, e)) for i, (a, b, c, d, e) in enumerate(PAIRS)
           if not (torch.equal(a, b) and torch.equal(b, c) and torch.equal(d, e))]
    print(f"rep {rep}: {len(PAIRS)} calls; (call, first != second, second != third, dy changed afterwards): {bad}", flush=True)
