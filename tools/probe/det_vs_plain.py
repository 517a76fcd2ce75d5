"""How far apart are the deterministic and the plain (atomic) mode on the test's tiny step?  (margins of
tests/test_deterministic_gpu.py::test_the_mode_changes_results_only_within_rounding)"""
import copy, os, sys, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import test_deterministic_gpu as T
from peppa_amd import hip as H
from peppa_amd.data import synthetic_batch
H.set_deterministic(True)
net = T._net(T._cfg())
state = copy.deepcopy(net.state_dict())
batch = synthetic_batch(4, 4, 32, 4000).to("cuda")
det = T._run(net, state, batch, steps=1)
names = ("audio_encoder.project.weight", "video_encoder.project.weight", "audio_encoder.audio.encoder.transformer.layers.11.feed_forward.output_dense.weight")
for rep in range(4):
    H.set_deterministic(False)
    plain = T._run(net, state, batch, steps=1)
    H.set_deterministic(True)
    print("loss diff %.2e" % abs(det[0][0].item() - plain[0][0].item()), " rel grad diffs:", ["%.3f" % ((det[1][n] - plain[1][n]).norm().item() / plain[1][n].norm().item()) for n in names], flush=True)
