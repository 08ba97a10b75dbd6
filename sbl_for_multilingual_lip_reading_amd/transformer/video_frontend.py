# coding: utf-8
"""Mirror of SBL_Multilingual_Lip_reading/transformer/video_frontend.py: Conv3d stem + ResNet-18 trunk.

The torch.nn conv / batch-norm modules below are parameter containers only (so state-dict keys, shapes,
.to(), pickling and the reference's init code behave identically); their own forward is never called.
Activations flow channels-last: (N*T, h, w, C)."""
import math

import torch
import torch.nn as nn

from ._env import config, ops


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride,
                     padding=1, bias=False)


def _conv_bn(x, conv, bn, res, relu, training, box_out=None, box_in=None, ctl=None):
    """conv -> BatchNorm2d -> (+res) -> (ReLU), NHWC, one tape node; BN side effects like nn.BatchNorm2d."""
    return ops.ConvBNFn.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, res, relu,
                              conv.stride[0], training, bn.momentum, bn.eps, box_out, box_in,
                              bn.num_batches_tracked if training else None, ctl)      # (+= 1 inside the finalize kernel)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(BasicBlock, self).__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # video_frontend.py:28-41 on NHWC activations
        box = {}      # lets conv2's backward do bn1's reduction pass in its epilogue (ops.ConvBNFn)
        # link: shared by this block's tape nodes; prev / pub: what the previous block published on its output tensor and
        # what this one publishes on its own (ops.ConvBNFn: residual gradient and the BatchNorm backward sums of the
        # previous block ride on conv1's input-gradient epilogue)
        link = {"identity": self.downsample is None}
        prev, pub = getattr(x, "_sbl_pub", None), {}
        out = _conv_bn(x, self.conv1, self.bn1, None, True, self.training, box_out=box,
                       ctl={"role": "conv1", "link": link, "prev": prev})
        if self.downsample is not None:
            residual = _conv_bn(x, self.downsample[0], self.downsample[1], None, False, self.training,
                                ctl={"role": "ds", "link": link, "pub": pub})
        else:
            residual = x
        y = _conv_bn(out, self.conv2, self.bn2, residual, True, self.training, box_in=box,
                     ctl={"role": "conv2", "link": link, "pub": pub})
        if self.training and torch.is_grad_enabled():
            y._sbl_pub = pub
        return y


class ResNet(nn.Module):

    def __init__(self, block, layers):
        self.inplanes = 64
        super(ResNet, self).__init__()
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion,
                          kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion),
            )

        layers = []
        layers.append(block(self.inplanes, planes, stride, downsample))
        self.inplanes = planes * block.expansion
        for i in range(1, blocks):
            layers.append(block(self.inplanes, planes))

        return nn.Sequential(*layers)

    def forward(self, x):
        """x: (N*T, h, w, 64) channels-last -> (N*T, 512)   (video_frontend.py:82-89)"""
        x = self.layer1(x)
        x = self.layer2(x)
        x = self.layer3(x)
        x = self.layer4(x)
        return ops.AvgPoolFn.apply(x)


class Lipreading(nn.Module):
    def __init__(self, hiddenDim=512, embedSize=256):
        super(Lipreading, self).__init__()
        self.inputDim = 512
        self.hiddenDim = hiddenDim
        self.embedSize = embedSize
        self.nLayers = 3
        # frontend3D (parameter containers; computed by the fused stem kernels)
        self.frontend3D = nn.Sequential(
            nn.Conv3d(1, 64, kernel_size=(5, 7, 7), stride=(1, 2, 2), padding=(2, 3, 3), bias=False),
            nn.BatchNorm3d(64),
            nn.ReLU(True),
            nn.MaxPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
        )
        # resnet
        self.resnet18 = ResNet(BasicBlock, [2, 2, 2, 2])
        # the reference's always-on F.dropout(p=0.5) (video_frontend.py:122); set to 0.0 for parity runs
        self.frontend_dropout_p = config.FRONTEND_DROPOUT_P
        self._initialize_weights()

    def _frontend_forward(self, x):
        """x: (N, 1, T, H, W) or (N, T, H, W) -> (N*T, 512)"""
        if x.dim() == 5:
            x = x[:, 0]
        conv, bn = self.frontend3D[0], self.frontend3D[1]
        x = ops.StemFn.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, self.training,
                             bn.momentum, bn.eps, bn.num_batches_tracked if self.training else None)
        return self.resnet18(x)

    def forward(self, x):
        frameLen = x.size(2) if x.dim() == 5 else x.size(1)
        x = self._frontend_forward(x)
        x = ops.dropout(x, self.frontend_dropout_p, True)      # active in eval too, like the reference
        x = x.view(-1, frameLen, self.inputDim)
        return x

    def _initialize_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv3d, nn.Conv2d, nn.Conv1d)):
                n = m.out_channels
                for ksz in m.kernel_size:
                    n *= ksz
                m.weight.data.normal_(0, math.sqrt(2. / n))
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, (nn.BatchNorm3d, nn.BatchNorm2d, nn.BatchNorm1d)):
                m.weight.data.fill_(1)
                m.bias.data.zero_()


device = config.device


def visual_frontend(pt=None):
    """video_frontend.py:176-190: build the frontend; with a local path, copy every name+shape-matching entry
    of that state dict (tensor-only load)."""
    model = Lipreading(hiddenDim=512, embedSize=256)
    if pt is not None:
        model_dict = model.state_dict()
        pretrained_dict = torch.load(pt, map_location=device, weights_only=True)
        print(len(pretrained_dict))
        pretrained_dict = {k: v for k, v in pretrained_dict.items()
                           if k in model_dict.keys() and v.size() == model_dict[k].size()}
        print('loaded params/tot params:{}/{}'.format(len(pretrained_dict), len(model_dict)))
        model_dict.update(pretrained_dict)
        model.load_state_dict(model_dict)
    return model
