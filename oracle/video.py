"""Oracle (test infrastructure): fp32 CPU restatement of the video backbones.

Follows the call sites `pig/models.py:113-154` (R3DEncoder: `.stem`, `.layer1-4`
of `torchvision.models.video.{r3d_18,mc3_18,r2plus1d_18}`) and `pig/models.py:156-200`
(ImageEncoder: `resnet18` trunk).  torchvision 0.10.1 (requirements.txt:76) is not
installed here, so the architecture is restated from its published definition
 as a straight composition of torch.nn.Conv3d / BatchNorm3d.  Module
paths equal torchvision's, so state-dicts interchange with the product encoder.
"""
import torch
from torch import nn


def _mid(inp, planes):
    # torchvision video/resnet.py BasicBlock: (in*planes*27) // (in*9 + 3*planes)
    return (inp * planes * 27) // (inp * 9 + 3 * planes)


def conv2plus1d(inp, out, mid, stride=1):
    return nn.Sequential(
        nn.Conv3d(inp, mid, (1, 3, 3), stride=(1, stride, stride), padding=(0, 1, 1), bias=False),
        nn.BatchNorm3d(mid),
        nn.ReLU(inplace=True),
        nn.Conv3d(mid, out, (3, 1, 1), stride=(stride, 1, 1), padding=(1, 0, 0), bias=False))


def conv3dsimple(inp, out, mid=None, stride=1):
    return nn.Conv3d(inp, out, 3, stride=stride, padding=1, bias=False)


def conv3dnotemporal(inp, out, mid=None, stride=1):
    return nn.Conv3d(inp, out, (1, 3, 3), stride=(1, stride, stride), padding=(0, 1, 1), bias=False)


def _ds_stride(builder, stride):
    return (1, stride, stride) if builder is conv3dnotemporal else (stride, stride, stride)


class BasicBlock3D(nn.Module):
    def __init__(self, inp, planes, builder, stride=1, downsample=None):
        super().__init__()
        mid = _mid(inp, planes)
        self.conv1 = nn.Sequential(builder(inp, planes, mid, stride), nn.BatchNorm3d(planes),
                                   nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(builder(planes, planes, mid), nn.BatchNorm3d(planes))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        res = x if self.downsample is None else self.downsample(x)
        return self.relu(self.conv2(self.conv1(x)) + res)


class VideoResNet18(nn.Module):
    """`.stem`, `.layer1..4`, `.avgpool`, `.fc` like torchvision's VideoResNet."""

    def __init__(self, version="r2plus1d_18"):
        super().__init__()
        if version == "r2plus1d_18":
            builders = [conv2plus1d] * 4
            self.stem = nn.Sequential(
                nn.Conv3d(3, 45, (1, 7, 7), stride=(1, 2, 2), padding=(0, 3, 3), bias=False),
                nn.BatchNorm3d(45), nn.ReLU(inplace=True),
                nn.Conv3d(45, 64, (3, 1, 1), stride=1, padding=(1, 0, 0), bias=False),
                nn.BatchNorm3d(64), nn.ReLU(inplace=True))
        elif version in ("r3d_18", "mc3_18"):
            builders = [conv3dsimple] * 4 if version == "r3d_18" else \
                [conv3dsimple] + [conv3dnotemporal] * 3
            self.stem = nn.Sequential(
                nn.Conv3d(3, 64, (3, 7, 7), stride=(1, 2, 2), padding=(1, 3, 3), bias=False),
                nn.BatchNorm3d(64), nn.ReLU(inplace=True))
        else:
            raise ValueError(f"Invalid version {version}")
        self.inplanes = 64
        self.layer1 = self._make(builders[0], 64, 1)
        self.layer2 = self._make(builders[1], 128, 2)
        self.layer3 = self._make(builders[2], 256, 2)
        self.layer4 = self._make(builders[3], 512, 2)
        self.avgpool = nn.AdaptiveAvgPool3d((1, 1, 1))
        self.fc = nn.Linear(512, 400)
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)

    def _make(self, builder, planes, stride):
        ds = None
        if stride != 1 or self.inplanes != planes:
            ds = nn.Sequential(
                nn.Conv3d(self.inplanes, planes, 1, stride=_ds_stride(builder, stride), bias=False),
                nn.BatchNorm3d(planes))
        blocks = [BasicBlock3D(self.inplanes, planes, builder, stride, ds)]
        self.inplanes = planes
        blocks.append(BasicBlock3D(planes, planes, builder))
        return nn.Sequential(*blocks)

    def trunk(self, x):
        return self.layer4(self.layer3(self.layer2(self.layer1(self.stem(x)))))


class BasicBlock2D(nn.Module):
    def __init__(self, inp, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inp, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        res = x if self.downsample is None else self.downsample(x)
        out = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        return self.relu(out + res)


class ResNet18(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.inplanes = 64
        self.layer1 = self._make(64, 1)
        self.layer2 = self._make(128, 2)
        self.layer3 = self._make(256, 2)
        self.layer4 = self._make(512, 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make(self, planes, stride):
        ds = None
        if stride != 1 or self.inplanes != planes:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False),
                               nn.BatchNorm2d(planes))
        blocks = [BasicBlock2D(self.inplanes, planes, stride, ds)]
        self.inplanes = planes
        blocks.append(BasicBlock2D(planes, planes))
        return nn.Sequential(*blocks)
