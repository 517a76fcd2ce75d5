"""Time the video tower and the audio tower of the C2 step separately (fwd+bwd), and the optimizer."""
import os, sys, copy, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch, yaml
import pig.models
from peppa_amd.data import synthetic_batch
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "hparams_base.yaml")))
cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
torch.manual_seed(0)
net = pig.models.PeppaPig(cfg).cuda().train()
opt = net.configure_optimizers()
b = synthetic_batch(64, 16, 112, 36800).to("cuda")
R = torch.randn(64, 512, device="cuda")
def t(fn, n=5):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def video():
    net.zero_grad(set_to_none=True); (net.encode_video(b.video) * R).sum().backward()
def video_fwd():
    with torch.no_grad(): net.encode_video(b.video)
def audio():
    net.zero_grad(set_to_none=True); (net.encode_audio(b.audio) * R).sum().backward()
def audio_fwd():
    with torch.no_grad(): net.encode_audio(b.audio)
def full():
    opt.zero_grad(set_to_none=True); net.training_step(b, 1).backward(); opt.step()
def optim():
    opt.step()
print(f"video fwd+bwd {t(video):.1f} ms | video fwd (no grad) {t(video_fwd):.1f} ms")
print(f"audio fwd+bwd {t(audio):.1f} ms | audio fwd (no grad) {t(audio_fwd):.1f} ms")
print(f"full step {t(full):.1f} ms | optimizer {t(optim):.2f} ms")
# host issue time vs device time: if the host needs as long to issue a step as the GPU to run it, the step is launch-bound
def issue_time(fn, n=5):
    fn(); fn(); torch.cuda.synchronize(); tot = 0.0
    for _ in range(n):
        t0 = time.perf_counter(); fn(); tot += time.perf_counter() - t0; torch.cuda.synchronize()
    return tot / n * 1e3
print(f"host issue time: video {issue_time(video):.1f} ms | audio {issue_time(audio):.1f} ms | full {issue_time(full):.1f} ms")
if os.environ.get("AB_WGRAD_OVERLAP"):
    from peppa_amd import video as PV
    for flag in (True, False, True, False):
        PV.OVERLAP_WGRAD = flag
        print(f"overlap_wgrad={flag}: video fwd+bwd {t(video):.2f} ms | full step {t(full):.2f} ms")
    PV.OVERLAP_WGRAD = True
if os.environ.get("AB_FUSED_ATTENTION"):
    from peppa_amd import audio as PA
    for flag in (True, False, True, False):
        PA.FUSED_ATTENTION = flag
        print(f"fused_attention={flag}: audio fwd+bwd {t(audio):.2f} ms | full step {t(full, 8):.2f} ms")
    PA.FUSED_ATTENTION = True
