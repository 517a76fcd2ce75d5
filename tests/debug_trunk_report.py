"""Debug helper: per-unit activation error of the HIP video trunk vs the CPU oracle."""
import copy, sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch
from oracle import model as O
import pig.models
from pig.execution import default_config
from peppa_amd import video as V
from peppa_amd.data import synthetic_batch

B, T, S = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 4, int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = copy.deepcopy(default_config); cfg["video"]["pretrained"] = False; cfg["audio"]["pretrained"] = False
torch.manual_seed(0)
ref = O.PeppaPigOracle(cfg)
net = pig.models.PeppaPig(cfg)
net.load_state_dict(ref.state_dict(), strict=False)
net = net.cuda()
batch = synthetic_batch(B, T, S, 4000)
acts = []
def hook(m, i, o):
    acts.append(o.detach())
rv = ref.video_encoder.video
plan = net.video_encoder.video.units()
# oracle: record BN outputs (pre-ReLU for non-final, we compare post-activation z instead via relu modules)
x = O.normalize_video(batch.video, "peppa")
ref.train()
outs = []
cur = x
def run_block(blk, inp):
    res = inp if blk.downsample is None else blk.downsample(inp)
    c1 = blk.conv1
    if isinstance(c1[0], torch.nn.Sequential):
        a = torch.relu(c1[0][1](c1[0][0](inp))); outs.append(a)
        b = torch.relu(c1[1](c1[0][3](a))); outs.append(b)
        c2 = blk.conv2
        c = torch.relu(c2[0][1](c2[0][0](b))); outs.append(c)
        d = torch.relu(c2[1](c2[0][3](c)) + res); outs.append(d)
        return d
with torch.no_grad():
    s = rv.stem
    a = torch.relu(s[1](s[0](cur))); outs.append(a)
    b = torch.relu(s[4](s[3](a))); outs.append(b)
    cur = b
    for layer in (rv.layer1, rv.layer2, rv.layer3, rv.layer4):
        for blk in layer:
            cur = run_block(blk, cur)
    V.FUSE_BN_APPLY = False      # (this report looks at every unit's activated output z)
    z, thw, tape = V.trunk_forward(net.video_encoder.video, batch.video.cuda(), "peppa", True, True)
torch.cuda.synchronize()
recs = []
for item in tape:
    if item[0] == "unit": recs.append(item[1])
    elif item[0] == "block_last": recs.append(item[1])
print(len(recs), len(outs))
for i, (r, o) in enumerate(zip(recs, outs)):
    C = o.shape[1]
    zz = r.z.float().cpu()[:, :C].reshape(o.shape[0], *o.shape[2:], C).permute(0, 4, 1, 2, 3)
    err = (zz - o).norm() / (o.norm() + 1e-9)
    yy = r.y.float().cpu()[:, :C]
    print(f"unit {i:2d} C={C:4d} shape={tuple(o.shape[2:])} rel err {err:.4f}  max {((zz-o).abs().max()):.4f} |o| {o.abs().max():.3f}")
