"""CPU-side checks (no GPU): the C ABI library loads and exports every symbol the header declares,
host logic of the pig.* mirror, config files, and the no-fallback rule."""
import copy
import os
import re
import numpy as np
import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from peppa_amd import _lib
    header = open(os.path.join(ROOT, "include", "peppa_hip.h")).read()
    declared = set(re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", header))
    declared -= {"pp_status", "pp_act", "pp_gather_mode"}
    assert len(declared) >= 45
    h = _lib.lib()
    for name in sorted(declared):
        assert hasattr(h, name), f"{name} declared in include/peppa_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert set(_lib.SIGNATURES) <= declared, set(_lib.SIGNATURES) - declared
    assert h.pp_version() >= 100
    assert isinstance(h.pp_last_error(), bytes)


def test_struct_layouts_match_header_order():
    from peppa_amd import _lib
    header = open(os.path.join(ROOT, "include", "peppa_hip.h")).read()
    body = header[header.index("typedef struct pp_gather {"):header.index("} pp_gather;")]
    names = re.findall(r"\b([A-Za-z_]+)\s*[,;]", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    assert [n for n in names if n != "int"] == [f[0] for f in _lib.Gather._fields_]


def test_no_cpu_fallback():
    import pig.models
    import pig.loss
    from peppa_amd._lib import PeppaHipError
    with pytest.raises(PeppaHipError):
        pig.loss.TripletLoss(0.2)(torch.randn(4, 8), torch.randn(4, 8))
    from peppa_amd import audio
    m = audio.wav2vec2_base(28)
    with pytest.raises(PeppaHipError):
        m(torch.zeros(1, 4000))
    assert "oracle" not in " ".join(sorted(k for k in __import__("sys").modules if k.startswith("peppa_amd")))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "peppa_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
    for f in ("run.py", "pig/__init__.py"):
        assert not re.search(r"^\s*(from|import)\s+oracle", open(os.path.join(ROOT, f)).read(), flags=re.M)


def test_state_dict_names_match_oracle_and_reference_layout():
    import warnings
    warnings.filterwarnings("ignore")
    import pig.models
    from pig.execution import default_config
    from oracle import model as O
    cfg = copy.deepcopy(default_config)
    cfg["video"]["pretrained"] = False
    cfg["audio"]["pretrained"] = False
    net, ref = pig.models.PeppaPig(cfg), O.PeppaPigOracle(cfg)
    a, b = net.state_dict(), ref.state_dict()
    assert set(a) == set(b)
    assert all(a[k].shape == b[k].shape for k in a)
    assert "video_encoder.video.layer1.0.conv1.0.0.weight" in a
    assert "audio_encoder.audio.encoder.transformer.layers.0.attention.k_proj.weight" in a
    assert sum(p.numel() for p in net.parameters()) == 126314341
    # freeze config: feature extractor + 12 layers frozen, the rest of the audio tower trains
    cfg2 = copy.deepcopy(cfg)
    cfg2["audio"]["freeze_feature_extractor"] = True
    cfg2["audio"]["freeze_encoder_layers"] = 12
    n2 = pig.models.PeppaPig(cfg2)
    trainable = sum(p.numel() for p in n2.audio_encoder.parameters() if p.requires_grad)
    assert trainable == 5159736  # ~5.16 M
    assert n2.audio_encoder.audio.encoder.transformer.pos_conv_embed.conv.weight_v.requires_grad


def test_config_errors_and_yaml_files():
    import pig.models
    from pig.execution import default_config, conditions
    for name, cond in conditions().items():
        on_disk = yaml.safe_load(open(os.path.join(ROOT, f"hparams_{name}.yaml")))
        if name == "static":
            cond = copy.deepcopy(cond)
            cond["video"]["pretrained"] = False   # as committed in the reference's hparams_static.yaml
        assert on_disk == cond, name
    bad = copy.deepcopy(default_config)
    bad["video"]["pretrained"] = bad["audio"]["pretrained"] = False
    bad["audio"]["pooling"] = "max"
    with pytest.raises(ValueError, match="Invalid pooling"):
        pig.models.PeppaPig(bad)
    # pretrained: true cannot be honoured offline and must not silently become random init
    with pytest.raises(RuntimeError, match="Kinetics"):
        pig.models.PeppaPig(copy.deepcopy(default_config))
    only_audio = copy.deepcopy(default_config)
    only_audio["video"]["pretrained"] = False
    with pytest.raises(RuntimeError, match="fairseq"):
        pig.models.PeppaPig(only_audio)


def test_bertadam_host_side(golden_dir):
    import pig.optimization as opt
    d = np.load(os.path.join(golden_dir, "ref_bertadam.npz"))
    for s, m in zip(d["sched_steps"], d["sched_mult"]):
        assert abs(opt.warmup_linear(s / 15000, 0.1) - m) < 1e-12
    assert opt.warmup_constant(0.5, 0.1) == 1.0 and abs(opt.warmup_cosine(0.05, 0.1) - 0.5) < 1e-12
    p = torch.nn.Parameter(torch.zeros(3))
    for kw in (dict(lr=-1.0), dict(lr=1e-3, schedule="nope"), dict(lr=1e-3, warmup=1.5), dict(lr=1e-3, b1=1.0),
               dict(lr=1e-3, e=-1.0)):
        with pytest.raises(ValueError):
            opt.BertAdam([p], **kw)
    o = opt.BertAdam([p], lr=1e-3, warmup=0.1, t_total=10)
    assert o.get_lr() == [0]
    p.grad = torch.ones(3)
    from peppa_amd._lib import PeppaHipError
    with pytest.raises(PeppaHipError):  # CPU parameters: no fallback
        o.step()


def test_triplet_pairing_and_geometry():
    import random
    import pig.triplet as T
    from peppa_amd.layers import ConvGeom, cpad
    from peppa_amd.audio import n_frames
    assert T.pairs([1, 2, 3, 4, 5]) == [[1, 2], [3, 4]]
    random.seed(0)
    dur = [1, 1, 2, 2, 2, 3, 1, 1]
    trip = list(T._triplets(range(len(dur)), lambda i: dur[i]))
    assert len(trip) == 3 and all(dur[a] == dur[b] and a != b for a, b in trip)
    assert [n_frames(n) for n in (16000, 36800, 73600, 101429)] == [49, 114, 229, 316]
    g = ConvGeom(64, (16, 56, 56), 64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    assert (g.M, g.Kf, g.Kd, g.out_cstride) == (64 * 16 * 56 * 56, 576, 9 * 144, 144)
    g = ConvGeom(2, (8, 28, 28), 64, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    assert g.out_thw == (8, 14, 14) and g.out_cstride == 240 and cpad(921) == 928
    from peppa_amd.data import Clip, collate
    # the reference's Clip fields (pig/data.py:28-37, built by featurize :72-76); durations are SECONDS
    clips = [Clip(video=torch.zeros(3, 4, 8, 8), audio=torch.zeros(1, 100), video_duration=0.4, audio_duration=0.41,
                  filename="a.avi"),
             Clip(torch.zeros(3, 6, 8, 8), torch.zeros(1, 80), 0.6, 0.59, "b.avi", offset=1.5, index=3)]
    b = collate(clips)
    assert b.video.shape == (2, 3, 6, 8, 8) and b.audio.shape == (2, 1, 100)
    assert torch.equal(b.video_duration, torch.tensor([0.4, 0.6])) and torch.equal(b.audio_duration, torch.tensor([0.41, 0.59]))
    assert clips[1].duration == 0.59 and clips[1].filename == "b.avi"
    trip = list(T.triplets([Clip(torch.zeros(1), torch.ones(1), 1.0, 2.0, "x"), Clip(torch.zeros(1), torch.ones(1), 1.1, 2.0, "y")]))
    assert len(trip) == 1 and isinstance(trip[0], T.Triplet)


# ---- the build's correctness requirements are enforced, not trusted ------------------
def test_shipped_code_objects_contain_no_packed_fp32():
    """v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 return wrong results on gfx950 when MFMA-issuing
    waves of another kernel share the SIMD (tools/probe/pk_mfma_min.hip).  Both shipped libraries are disassembled
    (llvm-objdump -d on every gfx950 code object): no build recipe, ROCm upgrade or float2 expression may bring them back."""
    from peppa_amd import build, _lib
    for prec, path in _lib.LIB_PATHS.items():
        assert os.path.exists(path), path
        assert build.packed_fp32_instructions(path) == {}, (prec, build.packed_fp32_instructions(path))
    # ... and the detector does see them where they exist (an object compiled without the flags)
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "pk.hip")
        with open(src, "w") as f:
            f.write("#include <hip/hip_runtime.h>\ntypedef float v2f __attribute__((ext_vector_type(2)));\n"
                    "__global__ void k(v2f* p) { p[threadIdx.x] = p[threadIdx.x] * p[threadIdx.x + 64] + p[threadIdx.x + 128]; }\n")
        obj = os.path.join(tmp, "pk.o")
        subprocess.run([build.HIPCC, "--offload-arch=gfx950", "-O3", "-c", src, "-o", obj], check=True, capture_output=True)
        assert build.packed_fp32_instructions(obj), "the detector must find v_pk_*_f32 in a float2 kernel built without the flags"
        obj2 = os.path.join(tmp, "pk2.o")
        subprocess.run([build.HIPCC] + build.FLAGS + ["-c", src, "-o", obj2], check=True, capture_output=True)
        assert build.packed_fp32_instructions(obj2) == {}, "the build flags must keep even explicit float2 arithmetic scalar"


def test_shipped_libraries_are_not_experiment_builds():
    from peppa_amd import _lib
    for prec in _lib.LIB_PATHS:
        assert _lib.lib(prec).pp_experimental_build() == 0


def test_build_staleness_covers_flags_and_compiler(tmp_path):
    from peppa_amd import build
    obj, src = tmp_path / "a.o", tmp_path / "a.hip"
    src.write_text("x")
    obj.write_text("o")
    cmd = ["hipcc", "-O3", "-c", str(src)]
    assert build._stale(str(obj), [str(src)], build._stamp(cmd))            # no stamp yet: a build/ that predates the stamps
    (tmp_path / "a.o.stamp").write_text(build._stamp(cmd))
    assert not build._stale(str(obj), [str(src)], build._stamp(cmd))
    assert build._stale(str(obj), [str(src)], build._stamp(cmd + ["-fno-slp-vectorize"]))   # a flag change rebuilds
    with pytest.raises(RuntimeError):
        build.build_library(extra_flags=["-DPP_TW_ABLATE=15"])             # never into the shipped library


def test_bertadam_scratch_is_size_checked():
    from peppa_amd import hip as H
    from peppa_amd._lib import TensorList, PeppaHipError
    tl = TensorList()
    tl.n_tensors = 5
    prev = H.DETERMINISTIC
    try:
        H.DETERMINISTIC = True     # the flag alone: the check runs before any device call
        with pytest.raises(PeppaHipError, match="norms"):
            H.bertadam_step(tl, None, None, 12, 16, torch.empty(5), 1e-3, 0.9, 0.999, 1e-6, 0.01, 1.0)
        H.DETERMINISTIC = False
        with pytest.raises(PeppaHipError, match="norms"):
            H.bertadam_step(tl, None, None, 12, 16, torch.empty(4), 1e-3, 0.9, 0.999, 1e-6, 0.01, 1.0)
    finally:
        H.DETERMINISTIC = prev
