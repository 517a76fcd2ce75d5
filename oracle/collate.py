"""CPU restatement of the reference's clip collation.  TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench cpu_baseline).

Follows
  * pig/data.py:66-72 `featurize`: every decoded frame (uint8, H x W x 3) becomes `torch.tensor(frame / 255).float()`
    -- a float64 division rounded once to float32 -- the frames are stacked and permuted (T,H,W,C) -> (C,T,H,W);
  * pig/util.py:30-33 `pad_video_batch`: zero frames appended up to the longest clip of the batch, then stacked;
  * pig/util.py:20-23 `pad_audio_batch`: zero samples appended up to the longest clip, then stacked;
  * pig/models.py:327-342 `build_transform`: per-channel (x - mean) / std on the padded batch (so padded frames become
    -mean/std, not zero).
Pinned by tests/golden/ref_collate.npz, which oracle/make_golden.py writes with the live `pig.util` padding functions.
"""
import numpy as np


def featurize_frames(frames_u8):
    """uint8 (T,H,W,3) -> float32 (3,T,H,W) in [0,1]."""
    frames_u8 = np.asarray(frames_u8)
    assert frames_u8.dtype == np.uint8 and frames_u8.ndim == 4 and frames_u8.shape[-1] == 3
    if frames_u8.shape[0] == 0:
        raise ValueError("Clip has zero frames.")
    return np.ascontiguousarray((frames_u8 / 255).astype(np.float32).transpose(3, 0, 1, 2))


def pad_video_batch(videos):
    """list of (3,T_i,H,W) -> (B,3,Tmax,H,W), zeros after T_i."""
    size = max(v.shape[1] for v in videos)
    out = np.zeros((len(videos),) + videos[0].shape[:1] + (size,) + videos[0].shape[2:], dtype=videos[0].dtype)
    for i, v in enumerate(videos):
        out[i, :, :v.shape[1]] = v
    return out


def pad_audio_batch(audios):
    """list of (1,L_i) -> (B,1,Lmax), zeros after L_i."""
    size = max(a.shape[1] for a in audios)
    out = np.zeros((len(audios), audios[0].shape[0], size), dtype=audios[0].dtype)
    for i, a in enumerate(audios):
        out[i, :, :a.shape[1]] = a
    return out


def collate(frames_list, audio_list):
    return pad_video_batch([featurize_frames(f) for f in frames_list]), pad_audio_batch(list(audio_list))


def pad_frames_u8(frames_list):
    """The uint8 batch (B,Tmax,H,W,3) the fused route keeps on the device."""
    size = max(f.shape[0] for f in frames_list)
    out = np.zeros((len(frames_list), size) + frames_list[0].shape[1:], dtype=np.uint8)
    for i, f in enumerate(frames_list):
        out[i, :f.shape[0]] = f
    return out


def normalize_ndhwc_bf16(video_f32, mean, std):
    """fp32 (B,3,T,H,W) -> bf16 bit patterns (uint16) [B][T][H][W][8] of (x - mean) * (1/std), channels 3..7 zero;
    float32 arithmetic in the order the stem's input kernel uses, round-to-nearest-even to bf16."""
    x = np.asarray(video_f32, dtype=np.float32)
    m = np.asarray(mean, dtype=np.float32).reshape(1, 3, 1, 1, 1)
    inv = (np.float32(1.0) / np.asarray(std, dtype=np.float32)).reshape(1, 3, 1, 1, 1)
    y = ((x - m) * inv).astype(np.float32).transpose(0, 2, 3, 4, 1)
    bits = np.ascontiguousarray(y).view(np.uint32)
    bits = ((bits + np.uint32(0x7FFF) + ((bits >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)
    out = np.zeros(y.shape[:-1] + (8,), dtype=np.uint16)
    out[..., :3] = bits
    return out
