"""Where the host spends its time while issuing one training step (cProfile over 5 steps, no syncs inside)."""
import cProfile, copy, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pig.models
from pig.execution import default_config
from peppa_amd.data import synthetic_batch

cfg = copy.deepcopy(default_config)
cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
torch.manual_seed(0)
net = pig.models.PeppaPig(cfg).cuda()
batch = synthetic_batch(64, 16, 112, 36800).to("cuda")
opt = net.configure_optimizers()


def step(i):
    loss = net.training_step(batch, i)
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for i in range(3):
    step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(5):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
