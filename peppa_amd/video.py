"""Video trunks (r2plus1d_18 / r3d_18 / mc3_18) on the HIP path.

Module tree and parameter names equal torchvision 0.10.1 `models.video.resnet.VideoResNet`
(`stem`, `layer1..4`, `avgpool`, `fc`), which is what `pig/models.py:122-127,141-150` builds and
what its checkpoints store.  The torch.nn modules below are parameter containers only: the
forward/backward of the trunk is a hand-scheduled chain of HIP kernels (implicit-GEMM convs with a
BatchNorm-statistics epilogue, streaming BN apply / backward passes) driven by `TrunkFn`.
"""
import torch
from torch import nn

from . import hip as H
from . import layers as L
from .hip import act16, f32

VIDEO_STATS = {  # data/out/stats.pt, data/out/kinetics-stats.pt, pig/models.py:335-336
    "peppa": ((0.62745821, 0.66273642, 0.66865104), (0.24167268, 0.20884572, 0.27490067)),
    "kinetics": ((0.43216, 0.394666, 0.37645), (0.22803, 0.22145, 0.216989)),
    "imagenet": ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225)),
}


def _conv(ci, co, k, s, p):
    return nn.Conv3d(ci, co, k, stride=s, padding=p, bias=False)


def _conv2plus1d(ci, co, mid, stride=1):
    return nn.Sequential(_conv(ci, mid, (1, 3, 3), (1, stride, stride), (0, 1, 1)), nn.BatchNorm3d(mid),
                         nn.ReLU(inplace=True), _conv(mid, co, (3, 1, 1), (stride, 1, 1), (1, 0, 0)))


def _conv3dsimple(ci, co, mid=None, stride=1):
    return _conv(ci, co, (3, 3, 3), stride, 1)


def _conv3dnotemporal(ci, co, mid=None, stride=1):
    return _conv(ci, co, (1, 3, 3), (1, stride, stride), (0, 1, 1))


class BasicBlock(nn.Module):
    def __init__(self, ci, planes, builder, stride=1, downsample=None):
        super().__init__()
        mid = (ci * planes * 27) // (ci * 9 + 3 * planes)
        self.conv1 = nn.Sequential(builder(ci, planes, mid, stride), nn.BatchNorm3d(planes), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(builder(planes, planes, mid), nn.BatchNorm3d(planes))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample


class VideoResNet(nn.Module):
    def __init__(self, version="r2plus1d_18"):
        super().__init__()
        if version == "r2plus1d_18":
            builders = [_conv2plus1d] * 4
            self.stem = nn.Sequential(_conv(3, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3)), nn.BatchNorm3d(45),
                                      nn.ReLU(inplace=True), _conv(45, 64, (3, 1, 1), 1, (1, 0, 0)),
                                      nn.BatchNorm3d(64), nn.ReLU(inplace=True))
        elif version in ("r3d_18", "mc3_18"):
            builders = [_conv3dsimple] * 4 if version == "r3d_18" else [_conv3dsimple] + [_conv3dnotemporal] * 3
            self.stem = nn.Sequential(_conv(3, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3)), nn.BatchNorm3d(64),
                                      nn.ReLU(inplace=True))
        else:
            raise ValueError(f"Invalid version {version}")
        self.version = version
        self._inplanes = 64
        self.layer1 = self._make_layer(builders[0], 64, 1)
        self.layer2 = self._make_layer(builders[1], 128, 2)
        self.layer3 = self._make_layer(builders[2], 256, 2)
        self.layer4 = self._make_layer(builders[3], 512, 2)
        self.avgpool = nn.AdaptiveAvgPool3d((1, 1, 1))  # unused by the reference
        self.fc = nn.Linear(512, 400)                   # unused by the reference
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, builder, planes, stride):
        ds = None
        if stride != 1 or self._inplanes != planes:
            ds_stride = (1, stride, stride) if builder is _conv3dnotemporal else (stride, stride, stride)
            ds = nn.Sequential(_conv(self._inplanes, planes, 1, ds_stride, 0), nn.BatchNorm3d(planes))
        blocks = [BasicBlock(self._inplanes, planes, builder, stride, ds), BasicBlock(planes, planes, builder)]
        self._inplanes = planes
        return nn.Sequential(*blocks)

    # ---- execution plan ---------------------------------------------------------------------------
    @staticmethod
    def _cbr(seq):
        """Flatten a module into [conv, bn, relu?] units in execution order."""
        mods = [m for m in seq.modules() if isinstance(m, (nn.Conv3d, nn.BatchNorm3d, nn.ReLU))]
        out, i = [], 0
        while i < len(mods):
            conv, bn = mods[i], mods[i + 1]
            relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
            out.append([conv, bn, relu])
            i += 3 if relu else 2
        return out

    def stem_plan(self):
        return [("unit", *u) for u in self._cbr(self.stem)]

    @classmethod
    def block_plan(cls, blk):
        """('block_begin', ds) , inner ('unit', conv, bn, relu)..., ('block_last', conv, bn, ds)."""
        us = cls._cbr(blk.conv1) + cls._cbr(blk.conv2)
        return ([("block_begin", blk.downsample)] + [("unit", *u) for u in us[:-1]] +
                [("block_last", us[-1][0], us[-1][1], blk.downsample)])

    def units(self):
        plan = self.stem_plan()
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                plan += self.block_plan(blk)
        return plan

    def trunk_parameters(self):
        """Parameters used by the trunk, in plan order (fc is excluded)."""
        ps = []
        for m in self.modules():
            if isinstance(m, (nn.Conv3d, nn.BatchNorm3d)):
                ps += list(m.parameters(recurse=False))
        return ps


class BasicBlock2D(nn.Module):
    def __init__(self, ci, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(ci, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample


class ResNet18(nn.Module):
    """torchvision.models.resnet18 parameter tree (`conv1, bn1, layer1..4, fc`), used per frame by the
    static ImageEncoder (pig/models.py:156-200).  Executed as (1,k,k) 3-D convolutions over [B][T][H][W]."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self._inplanes = 64
        self.layer1 = self._make_layer(64, 1)
        self.layer2 = self._make_layer(128, 2)
        self.layer3 = self._make_layer(256, 2)
        self.layer4 = self._make_layer(512, 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, stride):
        ds = None
        if stride != 1 or self._inplanes != planes:
            ds = nn.Sequential(nn.Conv2d(self._inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        blocks = [BasicBlock2D(self._inplanes, planes, stride, ds), BasicBlock2D(planes, planes)]
        self._inplanes = planes
        return nn.Sequential(*blocks)

    def units(self):
        plan = [("unit", self.conv1, self.bn1, True), ("maxpool",)]
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                plan += [("block_begin", blk.downsample), ("unit", blk.conv1, blk.bn1, True),
                         ("block_last", blk.conv2, blk.bn2, blk.downsample)]
        return plan

    def trunk_parameters(self):
        ps = []
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.BatchNorm2d)):
                ps += list(m.parameters(recurse=False))
        return ps


def _t3(v):
    """Conv2d modules (resnet18, static ImageEncoder) run as 3-D convolutions with a unit time tap."""
    return tuple(v) if len(v) == 3 else (1,) + tuple(v)


PAIRED_STEM = True   # A/B switch: the first convolution over pixel pairs (layers.ConvGeom.paired_stem)


def stem_pairs(conv, thw):
    """Does the first convolution `conv` run over pixel pairs for an input of (T, H, W)?  Then the normalised input is
    written with four channels per pixel (trunk_forward) and _geom(first=True) gives the paired geometry."""
    pad = conv.padding if len(conv.padding) == 3 else (0,) + tuple(conv.padding)
    return PAIRED_STEM and L.ConvGeom.paired_stem(1, thw, conv.in_channels, conv.out_channels, _t3(conv.kernel_size),
                                                  _t3(conv.stride), pad) is not None


def _geom(conv, B, thw, first=False):
    ci = conv.in_channels
    pad = conv.padding if len(conv.padding) == 3 else (0,) + tuple(conv.padding)
    if first and stem_pairs(conv, thw):
        return L.ConvGeom.paired_stem(B, thw, ci, conv.out_channels, _t3(conv.kernel_size), _t3(conv.stride), pad)
    kw = dict(in_cstride=8, cg_in=8) if first else {}
    return L.ConvGeom(B, thw, ci, conv.out_channels, _t3(conv.kernel_size), _t3(conv.stride), pad, **kw)


class _Tape:
    pass


def normalized_input(x, norm_kind, first_conv):
    """x fp32 [B][3][T][H][W] (or uint8 [B][T][H][W][3]) -> (normalised channels-last 16-bit input of `first_conv`, (T,H,W), B):
    eight channels per pixel (three real), or four where the first convolution runs over pixel pairs (stem_pairs)."""
    mean, std = VIDEO_STATS[norm_kind]
    if x.dtype == torch.uint8:               # padded decoder frames (B,T,H,W,3) from data.collate_device
        B, T, Hh, W, _ = x.shape
    else:
        B, _, T, Hh, W = x.shape
    cpp = 4 if (first_conv is not None and stem_pairs(first_conv, (T, Hh, W))) else 8
    cur = L.empty((B * T * Hh * W, cpp), act16(), x)
    if x.dtype == torch.uint8:
        H.video_normalize_u8_ndhwc(x, cur, mean, std)
    else:
        H.video_normalize_ndhwc(x, cur, mean, std)
    return cur, (T, Hh, W), B


def trunk_forward(net, x, norm_kind, training, save):
    """x fp32 [B][3][T][H][W] (or uint8 [B][T][H][W][3]) -> (z 16-bit [B*T'*H'*W'][512], (T',H',W'), tape)."""
    plan = net.units()
    cur, (T, Hh, W), B = normalized_input(x, norm_kind, plan[0][1] if plan and plan[0][0] == "unit" else None)
    with L.PrepPlan(("video", id(net), tuple(x.shape), bool(save))):      # every convolution's operands in one launch
        return run_plan(plan, cur, (T, Hh, W), B, training, save, first=True)


FUSE_BN_APPLY = True   # A/B switch (tools/ab_step.py fuse_bn_apply): see run_plan


def run_plan(plan, cur, thw, B, training, save, first=False):
    """Run a (partial) unit plan on a channels-last 16-bit activation [B*T*H*W][Cp].

    A unit whose activated output z = relu(bn(y)) feeds ONE convolution that can apply scale / shift / ReLU itself on its
    LDS windows (layers.can_fuse_bn_apply: the layer-1 temporal convolutions of r2plus1d_18 at the C2 shapes) skips its
    apply pass: the consumer -- and later its weight gradient -- reads the raw y, and z is never written (VERDICT r1
    item 5 ii; bit-identical, one read and one write of the 144-channel tensor less per unit)."""
    tape = []
    block_in = block_thw = None

    def run_unit(conv, bn, relu, inp, thw_in, residual=None, first=False, x_bn=None, next_conv=None):
        geom = _geom(conv, B, thw_in, first)
        need_dgrad = save and not first
        wf, wd = L.prep_conv_weights(conv.weight, geom, need_dgrad=need_dgrad)
        # train mode: batch statistics from the conv epilogue; eval mode: running statistics
        y, partials = L.conv_fwd(inp, geom, wf, stats=training, x_bn=x_bn)
        fuse = (FUSE_BN_APPLY and next_conv is not None and relu and residual is None and
                L.can_fuse_bn_apply(_geom(next_conv, B, geom.out_thw, False)))
        z, sv = L.bn_fwd(y, partials, partials.shape[0] if partials is not None else geom.nblk, geom.M, bn, relu=relu, residual=residual, eps=bn.eps,
                         momentum=bn.momentum, update_running=training, apply=not fuse)
        rec = None
        if save:
            rec = _Tape()
            rec.conv, rec.bn, rec.relu, rec.geom, rec.wd = conv, bn, relu, geom, wd
            rec.x, rec.y, rec.z, rec.sv, rec.first = inp, y, z, sv, first
            rec.x_bn = x_bn
            rec.has_res = residual is not None
        # (fused: the next unit reads y and applies (scale, shift, relu) itself)
        return (y if fuse else z), geom.out_thw, rec, ((sv.scale, sv.shift, relu) if fuse else None)

    pending = None            # (scale, shift, relu) the next convolution must apply to its input
    for k, item in enumerate(plan):
        nxt = plan[k + 1] if k + 1 < len(plan) else None
        if item[0] == "unit":
            _, conv, bn, relu = item
            next_conv = nxt[1] if nxt is not None and nxt[0] in ("unit", "block_last") else None
            cur, thw, rec, pending = run_unit(conv, bn, relu, cur, thw, first=first, x_bn=pending, next_conv=next_conv)
            first = False
            tape.append(("unit", rec))
        elif item[0] == "maxpool":
            assert pending is None
            T, Hh, W = thw
            Cp = cur.shape[1]
            Ho, Wo = (Hh - 1) // 2 + 1, (W - 1) // 2 + 1
            out = L.empty((B * T * Ho * Wo, Cp), act16(), cur)
            H.maxpool3x3s2_fwd(cur, out, B * T, Hh, W, Cp)
            tape.append(("maxpool", (cur, B * T, Hh, W, Cp) if save else None))
            cur, thw = out, (T, Ho, Wo)
        elif item[0] == "block_begin":
            assert pending is None
            block_in, block_thw = cur, thw
            tape.append(("block_begin", None))
        else:  # block_last
            _, conv, bn, ds = item
            ds_rec = None
            res = block_in
            if ds is not None:
                res, _, ds_rec, _ = run_unit(ds[0], ds[1], False, block_in, block_thw)
            cur, thw, rec, pending = run_unit(conv, bn, True, cur, thw, residual=res, x_bn=pending)
            tape.append(("block_last", rec, ds_rec))
    assert pending is None
    return cur, thw, tape


_WGRAD_STREAMS = {}
OVERLAP_WGRAD = True   # bench.py clears this for its isolated (one kernel at a time) roofline pass


def _wgrad_stream(device):
    device = torch.device(device)
    if device not in _WGRAD_STREAMS:
        _WGRAD_STREAMS[device] = torch.cuda.Stream(device=device)
    return _WGRAD_STREAMS[device]


_TOWER_STREAMS = {}


def tower_stream(device):
    """Side stream of the audio tower (PeppaPig.encode_pair)."""
    device = torch.device(device)
    if device not in _TOWER_STREAMS:
        _TOWER_STREAMS[device] = torch.cuda.Stream(device=device)
    return _TOWER_STREAMS[device]


def ensure_streams(device):
    """Create the step's side streams NOW (weight-gradient stream, audio-tower stream) and touch each once.

    ROCclr maps streams onto four hardware queues in creation order.  Created lazily, the weight-gradient stream comes
    to life in the first backward pass -- after RCCL has made its own stream for the first all-gather -- and then shares
    a hardware queue with a busy stream: +5 ms per step measured on one GPU (tools/probe/bench_dbg.py).  Data-parallel
    entry points call this before the first collective."""
    device = torch.device(device)
    if device.type != "cuda":
        return
    for st in (_wgrad_stream(device), tower_stream(device)):
        with torch.cuda.stream(st):
            torch.zeros(8, device=device).add_(1)
    torch.cuda.synchronize(device)


def trunk_backward(tape, dz, grads, overlap_wgrad=True):
    """dz 16-bit grad of the trunk output; fills `grads[param] = tensor`.

    The weight gradient of a unit is off the critical chain (dz -> BN backward -> dy -> data gradient -> next
    unit), so it is issued on a side stream: the MFMA/latency-bound wgrad kernels then overlap the HBM-bound
    BatchNorm backward passes and the data-gradient GEMMs of the following units."""
    main = torch.cuda.current_stream()
    side = _wgrad_stream(dz.device) if (overlap_wgrad and OVERLAP_WGRAD) else None

    def unit_bwd(rec, dz_in, relu, want_dres, dgrad_residual=None, need_dx=True, consumer=None):
        # units without a residual input recompute the ReLU mask from y (rec.has_res False -> z not read)
        dy, dres, dg, db = L.bn_bwd(dz_in, rec.y, rec.z if rec.has_res else None, rec.sv, rec.bn.weight, relu=relu,
                                    want_dres=want_dres)
        if rec.bn.weight.requires_grad:
            grads[rec.bn.weight], grads[rec.bn.bias] = dg, db
        if rec.conv.weight.requires_grad:
            if side is not None:
                side.wait_stream(main)              # dy (and everything before it) is ready
                with torch.cuda.stream(side):
                    grads[rec.conv.weight] = L.conv_wgrad(rec.x, dy, rec.geom, rec.conv.weight.shape, x_bn=rec.x_bn)
                dy.record_stream(side)              # keep the allocator from recycling them under the side stream
                rec.x.record_stream(side)
            else:
                grads[rec.conv.weight] = L.conv_wgrad(rec.x, dy, rec.geom, rec.conv.weight.shape, x_bn=rec.x_bn)
        dx = None
        if need_dx and not rec.first:
            dx = L.conv_dgrad(dy, rec.geom, rec.wd, residual=dgrad_residual, consumer=consumer)
        return dx, dres

    # the main chain in backward execution order: ("unit" | "block_last" | "block_first" | "maxpool", payload...)
    items = []
    i = len(tape) - 1
    while i >= 0:
        kind = tape[i][0]
        if kind == "unit":
            items.append(("unit", tape[i][1]))
            i -= 1
        elif kind == "maxpool":
            items.append(("maxpool", tape[i][1]))
            i -= 1
        elif kind == "block_last":
            _, rec, ds_rec = tape[i]
            j = i - 1
            inner = []
            while tape[j][0] != "block_begin":
                inner.append(tape[j][1])
                j -= 1
            if not inner:
                raise RuntimeError("block without inner units")
            items.append(("block_last", rec, ds_rec))
            items += [("unit", r) for r in inner[:-1]]
            items.append(("block_first", inner[-1]))
            i = j - 1
        else:
            i -= 1

    def consumer_of(k):
        """The BatchNorm unit that receives item k's data gradient as its dz: (y, z or None, saved statistics, relu) --
        the window data-gradient kernels take its backward sums in their epilogue (layers.conv_dgrad)."""
        if k + 1 >= len(items) or items[k + 1][0] == "maxpool":
            return None
        nxt = items[k + 1]
        r = nxt[1]
        return (r.y, r.z if r.has_res else None, r.sv, True if nxt[0] == "block_last" else r.relu)

    cur, skip = dz, None
    for k, item in enumerate(items):
        kind = item[0]
        if kind == "unit":
            cur, _ = unit_bwd(item[1], cur, item[1].relu, False, consumer=consumer_of(k))
        elif kind == "maxpool":
            x, N, Hh, W, Cp = item[1]
            dx = L.empty(x.shape, act16(), x)
            H.maxpool3x3s2_bwd(x, cur, dx, N, Hh, W, Cp)
            cur = dx
        elif kind == "block_last":
            _, rec, ds_rec = item
            cur, skip = unit_bwd(rec, cur, True, True, consumer=consumer_of(k))
            if ds_rec is not None:
                skip, _ = unit_bwd(ds_rec, skip, False, False)
        else:  # block_first: the skip connection's gradient joins in the data-gradient epilogue
            cur, _ = unit_bwd(item[1], cur, item[1].relu, False, dgrad_residual=skip, consumer=consumer_of(k))
            skip = None
    if side is not None:
        main.wait_stream(side)   # every weight gradient is complete before autograd hands them on
    return cur
