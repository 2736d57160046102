"""``MinkowskiEngine.modules.resnet_block.BasicBlock`` (imported at utils/minkunet.py:30):
relu(norm2(conv2(relu(norm1(conv1(x))))) + (downsample(x) or x)), expansion 1."""
import torch.nn as nn

from .. import MinkowskiBatchNorm, MinkowskiConvolution, MinkowskiReLU


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1,
                 dimension=3):
        super().__init__()
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dilation=dilation,
                                          dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dilation=dilation,
                                          dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        residual = x
        out = self.norm1.forward_fused(self.conv1(x), relu=True)
        if self.downsample is not None:
            residual = self.downsample(x)
        return self.norm2.forward_fused(self.conv2(out), residual=residual.F, relu=True)
