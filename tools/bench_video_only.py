"""Video tower alone (fwd + bwd, C2 shapes) for rocprofv3 --kernel-trace --stats."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch, yaml
import pig.models
from peppa_amd.data import synthetic_batch
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "hparams_base.yaml")))
cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
torch.manual_seed(0)
net = pig.models.PeppaPig(cfg).cuda().train()
b = synthetic_batch(64, 16, 112, 36800).to("cuda")
R = torch.randn(64, 512, device="cuda")
for _ in range(int(os.environ.get("N", "5"))):
    net.zero_grad(set_to_none=True)
    (net.encode_video(b.video) * R).sum().backward()
torch.cuda.synchronize()
