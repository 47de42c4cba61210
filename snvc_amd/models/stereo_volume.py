"""Global scene stack: plane-sweep cost volume -> 3D convs -> hourglass -> classifier.

The public reference does not ship its global model (``snvc/models/__init__.py:1-2`` has the
``StereoNet`` imports commented out); what it does ship is the op (``build_cost_volume``), the
blocks (``convbn_3d``, ``hourglass``) and the composition pattern of VernierScale's '3D' branch
(vernier.py:128-142 constructor, :366-371 forward -- itself dead code upstream: it defines
``hg_conv`` but calls ``hg_conv3d``).  ``GlobalStack`` is that pattern on a concat cost volume
(BASELINE.json configs[0..1]):

    volume = build_cost_volume(left, right, shift, ds)          [N, 2C, D, H, W]
    v = relu(bn(conv3(volume)))   2C -> C                        conv1
    v = relu(bn(conv3(v)))        C  -> C                        conv2
    v = v + hourglass(v)[0]                                      hg_conv3d
    cost = conv1x1x1(v)           C  -> 1                        classifier
"""
import torch
import torch.nn as nn

from ..extension.build_cost_volume import build_cost_volume
from .submodule import ConvBNReLU3d, HipConv3d, convbn_3d, hourglass


class GlobalStack(nn.Module):
    def __init__(self, c=32, gn=False):
        super().__init__()
        self.conv1 = ConvBNReLU3d(convbn_3d(2 * c, c, 3, 1, 1, gn=gn), nn.ReLU(inplace=True))
        self.conv2 = ConvBNReLU3d(convbn_3d(c, c, 3, 1, 1, gn=gn), nn.ReLU(inplace=True))
        self.hg_conv3d = hourglass(c, gn=gn)
        self.classifier = HipConv3d(c, 1, kernel_size=1, padding=0, stride=1, bias=False)
        for m in self.modules():  # same init as VernierScale (vernier.py:38-46)
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, volume):
        v = self.conv2(self.conv1(volume))
        v, _, _ = self.hg_conv3d(v, None, None, residual=v)   # v + hourglass(v)[0], add fused in the epilogue
        return self.classifier(v)

    def forward_pair(self, left, right, shift, downsample=1):
        """cost-volume build + 3D CNN forward: the unit BASELINE.json's metric counts."""
        return self.forward(build_cost_volume(left, right, shift, downsample))
