"""A/B of step-level switches on ONE box, alternating inside one process (boxes of the pool differ by +-5 %):
    python tools/ab_step.py fuse_bnr wgrad_side        # each named switch is toggled against the default"""
import os
import sys
import time
import warnings

warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yaml
import pig.models
from peppa_amd import layers as L
from peppa_amd import video as PV
from peppa_amd import models as PM
from peppa_amd.data import synthetic_batch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = yaml.safe_load(open(os.path.join(root, "hparams_base.yaml")))
cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
torch.manual_seed(0)
net = pig.models.PeppaPig(cfg).cuda().train()
opt = net.configure_optimizers()
b = synthetic_batch(64, 16, 112, 36800).to("cuda")


def step(i):
    opt.zero_grad(set_to_none=True)
    net.training_step(b, i).backward()
    opt.step()


def timed(n=12):
    step(0); step(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step(2 + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


class _Paired:
    """wgrad_side: the weight-gradient side stream also when the towers are paired (round-1 behaviour)."""
    orig = PM.VideoTrunkFn.forward


def set_switch(name, on):
    if name == "fuse_bnr":
        L.FUSE_BN_BWD_REDUCE = on
    elif name == "wgrad_side":
        PM.FORCE_WGRAD_SIDE = on
    elif name == "out_nt":
        from peppa_amd import hip as H
        H.set_option("win_out_nt", 1 if on else 0)
    elif name == "bn_tuned":
        from peppa_amd import hip as H
        H.set_option("bn_nt", 2 if on else 0)
        H.set_option("bn_grid", 32768 if on else 4096)
    elif name == "paired_stem":
        PV.PAIRED_STEM = on
    elif name == "fuse_bn_apply":
        PV.FUSE_BN_APPLY = on
    elif name == "group_wgrad":
        from peppa_amd import audio as PA
        PA.GROUP_WGRAD = on
    elif name == "win_producers_tw":          # the temporal form of the window kernel with producer waves as well (level 3)
        from peppa_amd import hip as H
        H.set_option("win_producers", 4 if on else 2)   # (4 is the default)
    elif name == "win_producers_all":         # also the 144-column spatial tiles (level 4) against level 3
        from peppa_amd import hip as H
        H.set_option("win_producers", 4 if on else 3)
    elif name == "win_producers":
        from peppa_amd import hip as H
        H.set_option(name, 4 if on else 0)
    elif name in ("tw_producers", "tw_narrow", "ring_producers"):
        from peppa_amd import hip as H
        H.set_option(name, 1 if on else 0)
    elif name == "win_tall":                  # 512-row window tiles for the narrow-output data gradient (1) against 256-row tiles (0)
        from peppa_amd import hip as H
        H.set_option("win_tall", 1 if on else 0)
    elif name == "win_s2d":                   # stride-2 spatial data gradients as one window-kernel launch
        L.WIN_S2D = on
    elif name == "wgrad_flat":
        from peppa_amd import hip as H
        H.set_option("wgrad_flat", 1 if on else 0)
    elif name == "r4_all":                    # every RUNTIME switch of round 4 at once (on = this round's defaults, off = round 3's)
        from peppa_amd import hip as H
        L.WIN_S2D = on
        H.set_option("wgrad_flat", 1 if on else 0)
        H.set_option("win_tall", 0 if on else 1)
        H.set_option("win_partial", 1 if on else 0)
        H.set_option("win_kpb", 2 if on else 1)
    elif name == "stem_window":     # the paired-pixel stem's forward as a window kernel
        L.STEM_WINDOW = bool(on)
    elif name == "igemm_big":       # 256 x 256 forward / data-gradient tiles (plain epilogues)
        from peppa_amd import hip as H
        H.set_option("igemm_big", 8192 if on else 0)
    elif name == "wgrad_big":       # 256 x 256 weight-gradient tiles
        from peppa_amd import hip as H
        H.set_option("wgrad_big", 32768 if on else 0)
    elif name == "prep_plan":       # one launch for a tower's convolution weight operands
        L.PREP_PLAN = bool(on)
    elif name == "win_stagger":
        from peppa_amd import hip as H
        H.set_option("win_stagger", 1 if on else 0)
    elif name.startswith("persist_cus"):       # persist_cus224: the persistent conv / GEMM kernels leave 32 CUs to the other stream
        from peppa_amd import hip as H
        H.set_option("persist_cus", int(name[len("persist_cus"):]) if on else 256)
    else:
        raise SystemExit(f"unknown switch {name}")


defaults = {"r4_all": True, "prep_plan": True, "wgrad_big": True, "igemm_big": False, "stem_window": True, "win_tall": False, "win_s2d": True, "wgrad_flat": True, "win_producers_all": True, "ring_producers": True, "win_producers_tw": True, "win_producers": True, "tw_producers": True, "tw_narrow": True, "group_wgrad": True, "win_stagger": False, "paired_stem": True, "fuse_bn_apply": True, "persist_cus248": False, "persist_cus240": False, "persist_cus224": False, "fuse_bnr": False, "wgrad_side": False, "bn_tuned": True, "out_nt": True}
for _ in range(3):
    step(0)
for name in sys.argv[1:]:
    for rep in range(3):
        for on in (defaults[name], not defaults[name]):
            set_switch(name, on)
            print(f"{name}={on}: {timed():.2f} ms/step", flush=True)
    set_switch(name, defaults[name])
