"""Oracle (test infrastructure): fp32 CPU restatement of torchaudio 0.9.1 wav2vec2_base.

Call sites in the reference: `pig/models.py:70-74` (construction via
`A.wav2vec2_base(num_out=28)`), `pig/models.py:101-105` (`self.audio(x)` when
`full` else `self.audio.extract_features(x)`), `pig/grsa.py:448-452`
(`feature_extractor(x, None)`).  torchaudio 0.9.1 (requirements.txt:74) is not
installed; the architecture is restated from its published definition
 and cross-checked against HF transformers' Wav2Vec2Model in
`tests/test_oracle_audio_hf.py`.  Module paths equal torchaudio's.
"""
import torch
from torch import nn
import torch.nn.functional as F

CONV_SPEC = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2


def n_frames(length):
    for _, k, s in CONV_SPEC:
        length = (length - k) // s + 1
    return length


class ConvLayerBlock(nn.Module):
    def __init__(self, inp, out, k, s, norm):
        super().__init__()
        self.conv = nn.Conv1d(inp, out, k, s, bias=False)
        self.layer_norm = nn.GroupNorm(out, out, affine=True) if norm else None

    def forward(self, x):
        x = self.conv(x)
        if self.layer_norm is not None:
            x = self.layer_norm(x)
        return F.gelu(x)


class FeatureExtractor(nn.Module):
    def __init__(self):
        super().__init__()
        blocks, inp = [], 1
        for i, (out, k, s) in enumerate(CONV_SPEC):
            blocks.append(ConvLayerBlock(inp, out, k, s, norm=(i == 0)))
            inp = out
        self.conv_layers = nn.ModuleList(blocks)

    def forward(self, x, length=None):
        x = x.unsqueeze(1)
        for blk in self.conv_layers:
            x = blk(x)
        return x.transpose(1, 2), length


class FeatureProjection(nn.Module):
    def __init__(self, inp, out, dropout):
        super().__init__()
        self.layer_norm = nn.LayerNorm(inp)
        self.projection = nn.Linear(inp, out)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        return self.dropout(self.projection(self.layer_norm(x)))


class ConvolutionalPositionalEmbedding(nn.Module):
    def __init__(self, dim=768, kernel=128, groups=16):
        super().__init__()
        self.conv = nn.Conv1d(dim, dim, kernel, padding=kernel // 2, groups=groups)
        self.conv = torch.nn.utils.weight_norm(self.conv, name="weight", dim=2)
        self.num_remove = 1 if kernel % 2 == 0 else 0

    def forward(self, x):
        x = self.conv(x.transpose(-2, -1))
        if self.num_remove:
            x = x[..., :-self.num_remove]
        return F.gelu(x).transpose(-2, -1)


class SelfAttention(nn.Module):
    def __init__(self, dim, heads, dropout):
        super().__init__()
        self.num_heads, self.head_dim = heads, dim // heads
        self.scaling = self.head_dim ** -0.5
        self.dropout = nn.Dropout(dropout)
        self.k_proj = nn.Linear(dim, dim)
        self.v_proj = nn.Linear(dim, dim)
        self.q_proj = nn.Linear(dim, dim)
        self.out_proj = nn.Linear(dim, dim)

    def forward(self, x, attention_mask=None):
        B, T, D = x.shape
        shape = (B, T, self.num_heads, self.head_dim)
        q = self.q_proj(x).view(*shape).transpose(2, 1)
        k = self.k_proj(x).view(*shape).permute(0, 2, 3, 1)
        v = self.v_proj(x).view(*shape).transpose(2, 1)
        w = self.scaling * (q @ k)
        if attention_mask is not None:
            w = w + attention_mask
        w = self.dropout(F.softmax(w, dim=-1))
        out = (w @ v).transpose(2, 1).reshape(B, T, D)
        return self.out_proj(out)


class FeedForward(nn.Module):
    def __init__(self, dim, inner, p_inter, p_out):
        super().__init__()
        self.intermediate_dense = nn.Linear(dim, inner)
        self.intermediate_dropout = nn.Dropout(p_inter)
        self.output_dense = nn.Linear(inner, dim)
        self.output_dropout = nn.Dropout(p_out)

    def forward(self, x):
        x = self.intermediate_dropout(F.gelu(self.intermediate_dense(x)))
        return self.output_dropout(self.output_dense(x))


class EncoderLayer(nn.Module):
    """Post-LN layer (wav2vec2-base: layer_norm_first=False)."""

    def __init__(self, dim, heads, inner, p):
        super().__init__()
        self.attention = SelfAttention(dim, heads, p)
        self.dropout = nn.Dropout(p)
        self.layer_norm = nn.LayerNorm(dim)
        self.feed_forward = FeedForward(dim, inner, p, p)
        self.final_layer_norm = nn.LayerNorm(dim)

    def forward(self, x, attention_mask=None):
        x = self.layer_norm(x + self.dropout(self.attention(x, attention_mask)))
        return self.final_layer_norm(x + self.feed_forward(x))


class Transformer(nn.Module):
    def __init__(self, dim, heads, inner, n_layers, p, layer_drop):
        super().__init__()
        self.pos_conv_embed = ConvolutionalPositionalEmbedding(dim)
        self.layer_norm = nn.LayerNorm(dim)
        self.layer_drop = layer_drop
        self.dropout = nn.Dropout(p)
        self.layers = nn.ModuleList([EncoderLayer(dim, heads, inner, p) for _ in range(n_layers)])

    def forward(self, x, attention_mask=None):
        x = self.dropout(self.layer_norm(x + self.pos_conv_embed(x)))
        for layer in self.layers:
            if not (self.training and torch.rand(1).item() <= self.layer_drop):
                x = layer(x, attention_mask)
        return x


class Encoder(nn.Module):
    def __init__(self, num_out, p, layer_drop):
        super().__init__()
        self.feature_projection = FeatureProjection(512, 768, p)
        self.transformer = Transformer(768, 12, 3072, 12, p, layer_drop)
        self.readout = nn.Linear(768, num_out)

    def forward(self, features, lengths=None):
        return self.readout(self.transformer(self.feature_projection(features)))


class Wav2Vec2Model(nn.Module):
    """`forward(wave (B,L)) -> ((B,T,num_out), None)`; `extract_features -> ((B,T,512), None)`."""

    def __init__(self, num_out=28, dropout=0.1, layer_drop=0.1):
        super().__init__()
        self.feature_extractor = FeatureExtractor()
        self.encoder = Encoder(num_out, dropout, layer_drop)

    def extract_features(self, waveforms, lengths=None):
        return self.feature_extractor(waveforms, lengths)

    def forward(self, waveforms, lengths=None):
        x, lengths = self.feature_extractor(waveforms, lengths)
        return self.encoder(x, lengths), lengths


def wav2vec2_base(num_out, dropout=0.1, layer_drop=0.1):
    return Wav2Vec2Model(num_out, dropout, layer_drop)
