"""On-device collate: oracle vs the reference's golden batch (CPU), HIP kernels vs the oracle (GPU).
Byte / integer work: every comparison is bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import collate as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_collate.npz")


def _case(d, tag):
    Hh, W = d[f"{tag}_hw"]
    frames, off = [], 0
    for t in d[f"{tag}_T"]:
        n = int(t) * Hh * W * 3
        frames.append(d[f"{tag}_frames"][off:off + n].reshape(int(t), Hh, W, 3))
        off += n
    audio, off = [], 0
    for l in d[f"{tag}_L"]:
        audio.append(d[f"{tag}_audio"][off:off + int(l)].reshape(1, int(l)))
        off += int(l)
    return frames, audio


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_matches_reference_batch(tag):
    d = np.load(GOLDEN)
    frames, audio = _case(d, tag)
    video, aud = O.collate(frames, audio)
    assert video.dtype == np.float32 and np.array_equal(video, d[f"{tag}_video_batch"])
    assert np.array_equal(aud, d[f"{tag}_audio_batch"])
    assert video[0, :, 0, 0, 0].tolist() == [0.0, 1.0, np.float32(128 / 255)]
    # padded frames / samples are exact zeros
    for i, t in enumerate(d[f"{tag}_T"]):
        assert not video[i, :, int(t):].any()
    with pytest.raises(ValueError, match="zero frames"):
        O.featurize_frames(np.zeros((0, 4, 4, 3), dtype=np.uint8))


def test_oracle_bf16_rounding_matches_torch():
    rng = np.random.default_rng(0)
    x = rng.random((2, 3, 2, 4, 4), dtype=np.float32)
    mean, std = (0.43, 0.39, 0.37), (0.22, 0.21, 0.2)
    got = O.normalize_ndhwc_bf16(x, mean, std)
    t = torch.from_numpy(x)
    m = torch.tensor(mean).view(1, 3, 1, 1, 1)
    inv = (1.0 / torch.tensor(std)).view(1, 3, 1, 1, 1)
    want = ((t - m) * inv).permute(0, 2, 3, 4, 1).contiguous().to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(got[..., :3], want) and not got[..., 3:].any()


# ------------------------------------------------------------------------------------------------------------- GPU
def _clips(rng, Ts, Ls, Hh, W):
    from peppa_amd.data import RawClip
    return [RawClip(frames=torch.from_numpy(rng.integers(0, 256, size=(t, Hh, W, 3), dtype=np.uint8)),
                    audio=torch.from_numpy((0.1 * rng.standard_normal((1, l))).astype(np.float32)),
                    video_duration=t / 10, audio_duration=l / 16000) for t, l in zip(Ts, Ls)]


CASES = [
    ([3, 5, 1, 4], [100, 257, 1, 64], 8, 8),          # ragged, vector path
    ([2, 1, 3], [33, 7, 50], 5, 7),                   # H*W % 4 != 0: scalar path, rows not multiples of 16 bytes
    ([4], [4096], 16, 16),                            # a single clip (nothing to pad)
    ([1, 1], [1, 1], 4, 4),                           # shortest possible clips
    ([16, 9, 16, 12, 16, 1, 7, 16, 16], [36800, 20000, 36799, 5, 36800, 12345, 36800, 16, 3], 112, 112),  # C2 frames
]


@pytest.mark.gpu
@pytest.mark.parametrize("Ts,Ls,Hh,W", CASES)
def test_collate_device_matches_oracle(Ts, Ls, Hh, W):
    from peppa_amd.data import collate_device
    rng = np.random.default_rng(len(Ts) * 1000 + Hh)
    clips = _clips(rng, Ts, Ls, Hh, W)
    want_v, want_a = O.collate([c.frames.numpy() for c in clips], [c.audio.numpy() for c in clips])
    batch = collate_device(clips, "cuda")
    assert batch.video.dtype == torch.float32 and tuple(batch.video.shape) == want_v.shape
    assert np.array_equal(batch.video.cpu().numpy(), want_v)
    assert np.array_equal(batch.audio.cpu().numpy(), want_a)
    assert batch.video_duration.tolist() == pytest.approx([t / 10 for t in Ts])
    assert batch.audio_duration.tolist() == pytest.approx([l / 16000 for l in Ls])
    u8 = collate_device(clips, "cuda", video_dtype=torch.uint8)
    assert u8.video.dtype == torch.uint8
    assert np.array_equal(u8.video.cpu().numpy(), O.pad_frames_u8([c.frames.numpy() for c in clips]))
    assert np.array_equal(u8.audio.cpu().numpy(), want_a)


@pytest.mark.gpu
def test_collate_device_matches_reference_golden():
    from peppa_amd.data import RawClip, collate_device
    d = np.load(GOLDEN)
    for tag in ("a", "b"):
        frames, audio = _case(d, tag)
        clips = [RawClip(torch.from_numpy(f.copy()), torch.from_numpy(a.copy())) for f, a in zip(frames, audio)]
        batch = collate_device(clips, "cuda")
        assert np.array_equal(batch.video.cpu().numpy(), d[f"{tag}_video_batch"])
        assert np.array_equal(batch.audio.cpu().numpy(), d[f"{tag}_audio_batch"])
        # clips without durations: seconds at the reference's 10 fps / 44.1 kHz (pig/preprocess.py:45-47, pig/data.py:26)
        assert batch.video_duration.tolist() == pytest.approx([t / 10 for t in d[f"{tag}_T"].tolist()])
        assert batch.audio_duration.tolist() == pytest.approx([a.shape[1] / 44100 for a in audio])


@pytest.mark.gpu
def test_unaligned_clip_pointers_take_the_scalar_path():
    """The C-ABI accepts any clip pointer; only 4-byte (video) / 16-byte (rows) aligned ones use vector loads."""
    from peppa_amd import hip as H
    rng = np.random.default_rng(3)
    Hh = W = 8
    Ts = [3, 2]
    frames = [rng.integers(0, 256, size=(t, Hh, W, 3), dtype=np.uint8) for t in Ts]
    flat = torch.zeros(1 + sum(f.size for f in frames) + 8, dtype=torch.uint8)
    offs, o = [], 1                                   # first clip at byte offset 1, second right behind it
    for f in frames:
        flat[o:o + f.size] = torch.from_numpy(f.reshape(-1))
        offs.append(o)
        o += f.size
    dev = flat.cuda()
    table = torch.tensor([[dev.data_ptr() + off, t] for off, t in zip(offs, Ts)], dtype=torch.int64).cuda()
    out = torch.empty(2, 3, 3, Hh, W, device="cuda")
    H.collate_video_u8(table, 2, 3, Hh, W, out)
    assert np.array_equal(out.cpu().numpy(), O.collate(frames, [np.zeros((1, 1), np.float32)] * 2)[0])
    rows = torch.tensor([[dev.data_ptr() + off, f.size] for off, f in zip(offs, frames)], dtype=torch.int64).cuda()
    padded = torch.empty(2, 3, Hh, W, 3, dtype=torch.uint8, device="cuda")
    H.collate_rows(rows, 2, 3 * Hh * W * 3, padded)
    assert np.array_equal(padded.cpu().numpy(), O.pad_frames_u8(frames))


@pytest.mark.gpu
@pytest.mark.parametrize("Hh,W", [(8, 8), (5, 7)])
def test_uint8_stem_input_is_bit_identical_to_the_fp32_route(Hh, W):
    from peppa_amd import hip as H
    from peppa_amd.data import collate_device
    from peppa_amd.video import VIDEO_STATS
    rng = np.random.default_rng(11)
    clips = _clips(rng, [3, 2, 4], [10, 20, 30], Hh, W)
    f32 = collate_device(clips, "cuda").video
    u8 = collate_device(clips, "cuda", video_dtype=torch.uint8).video
    for kind in ("peppa", "kinetics"):
        mean, std = VIDEO_STATS[kind]
        a = torch.empty(f32.numel() // 3, 8, dtype=torch.bfloat16, device="cuda")
        b = torch.empty_like(a)
        H.video_normalize_ndhwc(f32, a, mean, std)
        H.video_normalize_u8_ndhwc(u8, b, mean, std)
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
        want = O.normalize_ndhwc_bf16(f32.cpu().numpy(), mean, std).reshape(-1, 8)
        assert np.array_equal(b.view(torch.int16).cpu().numpy().view(np.uint16), want)
        # four channels per pixel (the paired-pixel stem's input): the same three values and a zero, from either route
        a4 = torch.empty(f32.numel() // 3, 4, dtype=torch.bfloat16, device="cuda")
        b4 = torch.empty_like(a4)
        H.video_normalize_ndhwc(f32, a4, mean, std)
        H.video_normalize_u8_ndhwc(u8, b4, mean, std)
        assert torch.equal(a4.view(torch.int16), b4.view(torch.int16)) and torch.equal(a4, a[:, :4])
    with pytest.raises(H.PeppaHipError):
        H.video_normalize_u8_ndhwc(u8.permute(0, 4, 1, 2, 3), b, mean, std)


@pytest.mark.gpu
def test_encode_video_accepts_the_uint8_batch():
    import copy
    import pig.models
    from pig.execution import default_config
    from peppa_amd.data import collate_device
    cfg = copy.deepcopy(default_config)
    cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
    torch.manual_seed(0)
    net = pig.models.PeppaPig(cfg).cuda().eval()
    clips = _clips(np.random.default_rng(2), [8, 5, 8, 3], [16000, 9000, 16000, 4000], 64, 64)
    with torch.no_grad():
        a = net.encode_video(collate_device(clips, "cuda").video)
        b = net.encode_video(collate_device(clips, "cuda", video_dtype=torch.uint8).video)
    assert a.shape == (4, 512) and torch.equal(a, b)


@pytest.mark.gpu
def test_collate_device_rejects_bad_batches():
    from peppa_amd.data import RawClip, collate_device
    ok = RawClip(torch.zeros(2, 4, 4, 3, dtype=torch.uint8), torch.zeros(1, 5))
    with pytest.raises(ValueError, match="zero frames"):
        collate_device([ok, RawClip(torch.zeros(0, 4, 4, 3, dtype=torch.uint8), torch.zeros(1, 5))])
    with pytest.raises(ValueError, match="one frame size"):
        collate_device([ok, RawClip(torch.zeros(2, 4, 6, 3, dtype=torch.uint8), torch.zeros(1, 5))])
    with pytest.raises(ValueError, match="uint8"):
        collate_device([RawClip(torch.zeros(2, 4, 4, 3), torch.zeros(1, 5))])
    with pytest.raises(ValueError, match="empty"):
        collate_device([])
