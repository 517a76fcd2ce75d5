"""Localise a run-to-run difference of the audio backward in deterministic mode: every layernorm_bwd / linear_dgrad /
gelu_bwd call of a step records clones of its tensor arguments (made on the calling stream right after the launch); two
runs of the same step are then compared call by call.      python tools/probe/det_trace.py"""
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
if "LN_BWD_ALONE" in os.environ:
    H.set_option("ln_bwd_alone", int(os.environ["LN_BWD_ALONE"]))
net = T._net(T._cfg())
state = copy.deepcopy(net.state_dict())
batch = synthetic_batch(2, 16, 112, 36800).to("cuda")
TRACE = []


def wrap(mod, name, outs_of):
    orig = getattr(mod, name)

    def f(*a, **k):
        r = orig(*a, **k)
        flat = []
        for t in list(a) + list(k.values()):
            flat += list(t) if isinstance(t, (tuple, list)) else [t]
        ins = [t.detach().clone() for t in flat if torch.is_tensor(t)]
        outs = [t.detach().clone() for t in outs_of(r, a, k) if torch.is_tensor(t)]
        TRACE.append((name, ins, outs))
        return r
    setattr(mod, name, f)


wrap(L, "layernorm_bwd", lambda r, a, k: r)
wrap(L, "layernorm_fwd", lambda r, a, k: [r[0], r[1][0], r[1][1]])
wrap(L, "linear_dgrad", lambda r, a, k: [r])
wrap(H, "gelu_bwd", lambda r, a, k: [a[2]])
wrap(H, "attention_bwd", lambda r, a, k: [a[-1]])


def run():
    TRACE.clear()
    out = T._run(net, state, batch, steps=1)
    return out, list(TRACE)


for rep in range(int(os.environ.get("REPS", "8"))):
    (a, ta), (b, tb) = run(), run()
    badg = [n for n in a[1] if not torch.equal(a[1][n], b[1][n])]
    msg = []
    if len(ta) != len(tb):
        msg.append(f"trace lengths {len(ta)} / {len(tb)}")
    for i, (x, y) in enumerate(zip(ta, tb)):
        din = [j for j, (p, q) in enumerate(zip(x[1], y[1])) if not torch.equal(p, q)]
        dout = [j for j, (p, q) in enumerate(zip(x[2], y[2])) if not torch.equal(p, q)]
        if din or dout:
            d = x[2][dout[0]].float() - y[2][dout[0]].float() if dout else None
            nz = (d != 0).nonzero() if d is not None else None
            msg.append(f"call {i} {x[0]}: inputs differing {din}, outputs differing {dout}" +
                       (f", {len(nz)} elements, first at {nz[0].tolist()} last at {nz[-1].tolist()}, max |d| {d.abs().max().item():.3e}" if dout else ""))
            if len(msg) > 2:
                break
    print("rep", rep, "gradients differ:", len(badg), "|", " ; ".join(msg) if msg else "traces identical", flush=True)
