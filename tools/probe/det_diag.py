"""Which kernel option breaks run-to-run bitwise equality in deterministic mode?  Runs the real-geometry batch-2 case of
tests/test_deterministic_gpu.py twice per option setting and names the tensors that differ.
    python tools/probe/det_diag.py"""
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
from peppa_amd.data import synthetic_batch
import test_deterministic_gpu as T

H.set_deterministic(True)
net = T._net(T._cfg())
state = copy.deepcopy(net.state_dict())
batch = synthetic_batch(2, 16, 112, 36800).to("cuda")
from peppa_amd import audio as PA
PA.GROUP_WGRAD = os.environ.get("GROUP_WGRAD", "1") == "1"
for opts in ({"win_producers": 2, "tw_producers": 1}, {"win_producers": 0, "tw_producers": 0}):
    for k, v in opts.items():
        H.set_option(k, v)
    for rep in range(int(os.environ.get('REPS', '6'))):
        a = T._run(net, state, batch)
        b = T._run(net, state, batch)
        badl = [i for i, (x, y) in enumerate(zip(a[0], b[0])) if not torch.equal(x, y)]
        badg = [n for n in a[1] if not torch.equal(a[1][n], b[1][n])]
        print(opts, "rep", rep, "losses differ:", badl, "gradients differ:", len(badg), [n.replace("audio_encoder.audio.", "") for n in badg], flush=True)
