"""PyTorch-CPU restatement of the reference's 3D module graphs -- TEST INFRASTRUCTURE.

The reference executes this part of the path as plain ``torch.nn`` modules (cuDNN on
its GPUs, ATen/oneDNN on CPU); the arithmetic lives in PyTorch, pinned by the reference
at pytorch 1.9.0 / cudnn 8.2.1 (spec-file.txt:20,50,253).  This file rebuilds the same
graphs, with the same state-dict keys, from stock ``torch.nn`` layers so that

  * ``tests/golden/make_golden.py`` can check it layer-for-layer against the imported
    reference modules in this container (where /root/reference exists), and
  * the GPU box (where the reference does not exist) has a CPU checker and a CPU
    baseline for the HIP kernels.

Nothing under ``snvc_amd/`` imports this module.

Reference locations restated here
  convbn_3d                 snvc/models/submodule.py:32-50
  hourglass                 snvc/models/submodule.py:85-168
  get_hg_down_sample        snvc/models/submodule.py:170-181
  get_hg_up_sample          snvc/models/submodule.py:197-208
  hourglass_downsample_16   snvc/models/submodule.py:223-268
  disparityregression       snvc/models/submodule.py:76-83
  convbn / hourglass2d / hourglass2d_downsample_16 (2D BEV neck)
                            snvc/models/submodule.py:11-29,183-221,270-361
  BasicBlock / basicdownsample (coord head)   snvc/models/hrnet.py:25-69
  VernierScale 3D trunk     snvc/models/vernier.py:249-313 (ctor), :414-458 (forward)
  _sample_2d_feat           snvc/models/vernier.py:323-349
  cfg-1/2 "global" stack    snvc/models/vernier.py:128-142,366-371 ('3D' branch pattern)
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- 3D blocks
def convbn_3d(cin, cout, kernel_size, stride, pad, dilation=1, gn=False, groups=32):
    return nn.Sequential(
        nn.Conv3d(cin, cout, kernel_size=kernel_size, padding=pad, dilation=dilation,
                  stride=stride, bias=False),
        nn.GroupNorm(groups, cout) if gn else nn.BatchNorm3d(cout))


def _deconv_norm(cin, cout, gn):
    return nn.Sequential(
        nn.ConvTranspose3d(cin, cout, kernel_size=3, padding=1, output_padding=1, stride=2,
                           bias=False),
        nn.GroupNorm(32, cout) if gn else nn.BatchNorm3d(cout))


class hourglass(nn.Module):
    def __init__(self, inplanes, gn=False):
        super().__init__()
        c = inplanes
        self.conv1 = nn.Sequential(convbn_3d(c, 2 * c, 3, 2, 1, gn=gn), nn.ReLU(inplace=True))
        self.conv2 = convbn_3d(2 * c, 2 * c, 3, 1, 1, gn=gn)
        self.conv3 = nn.Sequential(convbn_3d(2 * c, 2 * c, 3, 2, 1, gn=gn), nn.ReLU(inplace=True))
        self.conv4 = nn.Sequential(convbn_3d(2 * c, 2 * c, 3, 1, 1, gn=gn), nn.ReLU(inplace=True))
        self.conv5 = _deconv_norm(2 * c, 2 * c, gn)
        self.conv6 = _deconv_norm(2 * c, c, gn)

    def forward(self, x, presqu, postsqu):
        out = self.conv1(x)
        pre = self.conv2(out)
        pre = F.relu(pre + postsqu) if postsqu is not None else F.relu(pre)
        out = self.conv4(self.conv3(pre))
        post = F.relu(self.conv5(out) + (presqu if presqu is not None else pre))
        return self.conv6(post), pre, post


def get_hg_down_sample(cin, cout, gn, downsample=True):
    return nn.Sequential(convbn_3d(cin, cout, 3, 2 if downsample else 1, 1, gn=gn),
                         nn.ReLU(inplace=True))


def get_hg_up_sample(cin, cout, gn):
    return _deconv_norm(cin, cout, gn)


class hourglass_downsample_16(nn.Module):
    def __init__(self, inplanes, gn=False):
        super().__init__()
        c = inplanes
        self.conv1 = get_hg_down_sample(c, 2 * c, gn)
        self.conv2 = get_hg_down_sample(2 * c, 2 * c, gn, False)
        self.conv3 = get_hg_down_sample(2 * c, 2 * c, gn)
        self.conv4 = get_hg_down_sample(2 * c, 2 * c, gn, False)
        self.conv5 = get_hg_down_sample(2 * c, 2 * c, gn)
        self.conv6 = get_hg_down_sample(2 * c, 2 * c, gn, False)
        self.conv7 = get_hg_down_sample(2 * c, 2 * c, gn)
        self.conv8 = get_hg_down_sample(2 * c, 2 * c, gn, False)
        self.conv9 = get_hg_up_sample(2 * c, 2 * c, gn)
        self.conv10 = get_hg_up_sample(2 * c, 2 * c, gn)
        self.conv11 = get_hg_up_sample(2 * c, 2 * c, gn)
        self.conv12 = get_hg_up_sample(2 * c, c, gn)

    def forward(self, x):
        o2 = self.conv2(self.conv1(x))
        o4 = self.conv4(self.conv3(o2))
        o6 = self.conv6(self.conv5(o4))
        o8 = self.conv8(self.conv7(o6))
        o10 = self.conv10(self.conv9(o8) + o6)
        o11 = self.conv11(o10 + o4)
        return self.conv12(o11 + o2)


def disparityregression(x, depth):
    """snvc/models/submodule.py:81-83 (the ctor's .cuda() buffer is unused by forward)."""
    return torch.sum(x * depth[None, :, None, None], 1)


# ----------------------------------------------------------------------------- 2D BEV neck
def convbn(cin, cout, kernel_size, stride, pad, dilation, gn=False, groups=32):
    return nn.Sequential(
        nn.Conv2d(cin, cout, kernel_size=kernel_size, stride=stride,
                  padding=dilation if dilation > 1 else pad, dilation=dilation, bias=False),
        nn.GroupNorm(groups, cout) if gn else nn.BatchNorm2d(cout))


def _deconv2d_norm(cin, cout, gn):
    return nn.Sequential(
        nn.ConvTranspose2d(cin, cout, kernel_size=3, padding=1, output_padding=1, stride=2, bias=False),
        nn.GroupNorm(32, cout) if gn else nn.BatchNorm2d(cout))


class hourglass2d(nn.Module):
    def __init__(self, inplanes, gn=False):
        super().__init__()
        c = inplanes
        self.conv1 = nn.Sequential(convbn(c, 2 * c, 3, 2, 1, 1, gn=gn), nn.ReLU(inplace=True))
        self.conv2 = convbn(2 * c, 2 * c, 3, 1, 1, 1, gn=gn)
        self.conv3 = nn.Sequential(convbn(2 * c, 2 * c, 3, 2, 1, 1, gn=gn), nn.ReLU(inplace=True))
        self.conv4 = nn.Sequential(convbn(2 * c, 2 * c, 3, 1, 1, 1, gn=gn), nn.ReLU(inplace=True))
        self.conv5 = _deconv2d_norm(2 * c, 2 * c, gn)
        self.conv6 = _deconv2d_norm(2 * c, c, gn)

    def forward(self, x, presqu, postsqu):
        out = self.conv1(x)
        pre = self.conv2(out)
        pre = F.relu(pre + postsqu) if postsqu is not None else F.relu(pre)
        out = self.conv4(self.conv3(pre))
        post = F.relu(self.conv5(out) + (presqu if presqu is not None else pre))
        return self.conv6(post), pre, post


def _down2d(cin, cout, gn, downsample=True):
    return nn.Sequential(convbn(cin, cout, 3, 2 if downsample else 1, 1, 1, gn=gn), nn.ReLU(inplace=True))


class hourglass2d_downsample_16(nn.Module):
    def __init__(self, inplanes, gn=False):
        super().__init__()
        c = inplanes
        self.conv1 = _down2d(c, 2 * c, gn)
        self.conv2 = _down2d(2 * c, 2 * c, gn, False)
        self.conv3 = _down2d(2 * c, 2 * c, gn)
        self.conv4 = _down2d(2 * c, 2 * c, gn, False)
        self.conv5 = _down2d(2 * c, 2 * c, gn)
        self.conv6 = _down2d(2 * c, 2 * c, gn, False)
        self.conv7 = _down2d(2 * c, 2 * c, gn)
        self.conv8 = _down2d(2 * c, 2 * c, gn, False)
        self.conv9 = _deconv2d_norm(2 * c, 2 * c, gn)
        self.conv10 = _deconv2d_norm(2 * c, 2 * c, gn)
        self.conv11 = _deconv2d_norm(2 * c, 2 * c, gn)
        self.conv12 = _deconv2d_norm(2 * c, c, gn)

    def forward(self, x):
        o2 = self.conv2(self.conv1(x))
        o4 = self.conv4(self.conv3(o2))
        o6 = self.conv6(self.conv5(o4))
        o8 = self.conv8(self.conv7(o6))
        o10 = self.conv10(self.conv9(o8) + o6)
        o11 = self.conv11(o10 + o4)
        return self.conv12(o11 + o2)


class BasicBlock2d(nn.Module):
    """snvc/models/hrnet.py:25-54"""

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes, momentum=0.1)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, momentum=0.1)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        res = x if self.downsample is None else self.downsample(x)
        return self.relu(out + res)


def basicdownsample(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1, stride=2, bias=False), nn.BatchNorm2d(cout))


# ----------------------------------------------------------------------------- gather
def sample_2d_feat(left, right, l_pts, r_pts, resolution, grid_hwl):
    """snvc/models/vernier.py:323-349, aggregate="concat", WITHOUT the in-place side effect
    on the caller's coordinate tensors (the reference normalises through a view)."""
    nh, nw, nl = grid_hwl
    n, f = left.shape[0], left.shape[1]

    def norm(p):
        p = p.permute(0, 2, 1).reshape(n, nh, nw * nl, 2).clone()
        p[:, :, :, 0] = p[:, :, :, 0] / resolution[1] * 2 - 1
        p[:, :, :, 1] = p[:, :, :, 1] / resolution[0] * 2 - 1
        return p

    fl = F.grid_sample(left, norm(l_pts), mode="bilinear", padding_mode="zeros",
                       align_corners=False).reshape(n, f, nh, nw, nl)
    fr = F.grid_sample(right, norm(r_pts), mode="bilinear", padding_mode="zeros",
                       align_corners=False).reshape(n, f, nh, nw, nl)
    return torch.cat([fl, fr], dim=1)


# ----------------------------------------------------------------------------- local trunk
class VernierTrunk(nn.Module):
    """The BEV_type3 network of VernierScale without the HRNet backbone.

    Attribute names equal the reference's so that state-dict keys coincide
    (vernier.py:249-313); ``predict_3d_heatmaps`` follows vernier.py:414-458.
    """

    def __init__(self, dim=32, grid=(32, 128, 192), gn=False, num_parts=9, part_reg_head=False,
                 x_range=(-1.0, 1.0), z_range=(-1.0, 1.0), heads=True, vernier_type="BEV_type3"):
        """``heads=False``: only the 3D trunk's layers (vernier.py:249-289) are built -- for grids whose BEV neck the reference
        cannot instantiate (nh not in {16, 32}: BASELINE configs[2] 96^3 crops, configs[4] 80x160x160; vernier.py:290-295 raises);
        ``trunk_3d`` is the same code either way, and the default construction is what tests/golden pins."""
        super().__init__()
        nh, nw, nl = grid
        self.grid = grid
        self.dim = dim
        self.small = nw <= 16
        self.vimg_feat = nn.Sequential(convbn_3d(2 * dim, dim, 1, 1, 0, gn=gn), nn.ReLU(inplace=True))
        self.conv1 = nn.Sequential(convbn_3d(2 * dim, dim, 7, 1, 3, gn=gn), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(convbn_3d(dim, dim, 5, 1, 2, gn=gn), nn.ReLU(inplace=True))
        self.conv3 = nn.Sequential(convbn_3d(dim, dim, 5, 1, 4, dilation=2, gn=gn), nn.ReLU(inplace=True))
        self.conv4 = nn.Sequential(convbn_3d(2 * dim, dim, 3, 1, 1, gn=gn), nn.ReLU(inplace=True))
        self.hg_conv3d = hourglass(dim, gn=gn) if self.small else hourglass_downsample_16(dim, gn=gn)
        self.fg_cls_head = nn.Sequential(convbn_3d(dim, dim, 3, 1, 1, gn=gn), nn.ReLU(inplace=True),
                                         nn.Conv3d(dim, 1, 3, 1, 1, bias=False), nn.Sigmoid())
        if part_reg_head:
            self.part_reg_head = nn.Sequential(convbn_3d(dim, dim, 3, 1, 1, gn=gn), nn.ReLU(inplace=True),
                                               nn.Conv3d(dim, 27, 1, 1, 0, bias=False))
        self.pool_3d = nn.AvgPool3d((4, 1, 1), stride=(4, 1, 1))
        if not heads:
            self._init_weights()
            return
        self.vernier_type = vernier_type
        if vernier_type == "BEV_type2":      # vernier.py:191-248: the same 3D trunk, conv5 over dim * 8 channels, no coordinate head
            dim_height = dim * 8
        elif nh == 32:
            dim_height = 256
        elif nh == 16:
            dim_height = 128
        else:
            raise NotImplementedError
        self.conv5 = nn.Sequential(convbn(dim_height, 64, 3, 1, 1, 1, gn=gn), nn.ReLU(inplace=True))
        self.hm1 = hourglass2d(64, gn=gn) if self.small else hourglass2d_downsample_16(64, gn=gn)
        self.hm2 = nn.Conv2d(64, num_parts, 3, 1, 1, bias=False)
        if vernier_type == "BEV_type2":
            self._init_weights()
            return
        # coord head, vernier.py:68-93
        mods = [BasicBlock2d(num_parts + 2, num_parts * 2, stride=2,
                             downsample=basicdownsample(num_parts + 2, num_parts * 2))]
        for _ in range(int(4 - np.log2(192 / nl))):
            mods.append(BasicBlock2d(num_parts * 2, num_parts * 2, stride=2,
                                     downsample=basicdownsample(num_parts * 2, num_parts * 2)))
        mods.append(nn.Conv2d(num_parts * 2, num_parts * 2, kernel_size=(6, 4)))
        mods.append(nn.Sigmoid())
        self.coord_head = nn.Sequential(*mods)
        # coordinate maps, vernier.py:99-114
        mh, mw = nl, nw
        x_map = np.tile(np.linspace(0, 1, mw), (mh, 1)).reshape(1, 1, mh, mw)
        z_map = np.tile(np.linspace(0, 1, mh).reshape(mh, 1), (1, mw)).reshape(1, 1, mh, mw)
        self.coor_maps = torch.from_numpy(np.concatenate([x_map, z_map], axis=1).astype(np.float32))
        self._init_weights()

    def _init_weights(self):
        # init, vernier.py:38-54
        for m in self.modules():
            if isinstance(m, (nn.Conv3d, nn.Conv2d)):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm3d, nn.BatchNorm2d)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def trunk_3d(self, voxel):
        """vernier.py:415-438 -> (bev [N, F*nh/4, nw, nl], occupancy [N,1,nh,nw,nl], offset)."""
        f = self.vimg_feat(voxel)
        v = self.conv1(voxel)
        v = self.conv2(v) + v
        v = self.conv3(v) + v
        v = (self.hg_conv3d(v, None, None)[0] if self.small else self.hg_conv3d(v)) + v
        occ = self.fg_cls_head(v)
        offset = self.part_reg_head(v) if hasattr(self, "part_reg_head") else None
        v = torch.cat([v, f * occ], dim=1)
        v = self.pool_3d(self.conv4(v))
        n, _, _, w, l = v.shape
        return v.reshape(n, -1, w, l), occ, offset

    def heads_2d(self, bev):
        """vernier.py:440-450"""
        bev = self.conv5(bev)
        feats = (self.hm1(bev, None, None)[0] if self.small else self.hm1(bev)).permute(0, 1, 3, 2)
        heat = self.hm2(feats)
        if self.vernier_type == "BEV_type2":      # vernier.py:409-410: heat maps only
            return heat, None
        n = len(heat)
        aug = torch.cat([heat, self.coor_maps.repeat(n, 1, 1, 1).to(heat.device)], dim=1)
        return heat, self.coord_head(aug).view(n, -1, 2)

    def predict_3d_heatmaps(self, voxel):
        bev, occ, offset = self.trunk_3d(voxel)
        heat, coords = self.heads_2d(bev)
        return heat, occ.squeeze(1), offset, coords, None


# ----------------------------------------------------------------------------- global stack
class GlobalStack(nn.Module):
    """cfg-1 / cfg-2 stack of SURVEY.md section 8(d): the reference's '3D' branch pattern
    (vernier.py:128-142 ctor, :366-371 forward) on a concat cost volume, 2C -> C channels.

    conv1 = convbn_3d(2C, C, 3,1,1)+ReLU; conv2 = convbn_3d(C, C, 3,1,1)+ReLU;
    hg_conv3d = hourglass(C); voxel = voxel + hg(voxel)[0]; classifier = Conv3d(C,1,k1).
    (The upstream branch defines ``hg_conv`` but calls ``hg_conv3d`` -- dead code there; the
    attribute is named ``hg_conv3d`` here so the forward runs.)
    """

    def __init__(self, c=32, gn=False):
        super().__init__()
        self.conv1 = nn.Sequential(convbn_3d(2 * c, c, 3, 1, 1, gn=gn), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(convbn_3d(c, c, 3, 1, 1, gn=gn), nn.ReLU(inplace=True))
        self.hg_conv3d = hourglass(c, gn=gn)
        self.classifier = nn.Conv3d(c, 1, kernel_size=1, padding=0, stride=1, bias=False)
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, volume):
        v = self.conv2(self.conv1(volume))
        v = v + self.hg_conv3d(v, None, None)[0]
        return self.classifier(v)


# ----------------------------------------------------------------------------- seeding helpers
def seeded_state_dict(module: nn.Module, seed: int):
    """Deterministic, non-trivial parameters/buffers drawn from numpy's default_rng in
    state_dict() key order (SURVEY.md section 8c golden-vector plan): conv weights
    ~ N(0, sqrt(2/fan_in)) so activations stay O(1) through ~20 layers; norm gamma in [0.5,1.5],
    beta in [-0.2,0.2], running_mean in [-0.2,0.2], running_var in [0.5,1.5]."""
    rng = np.random.default_rng(seed)
    sd = {}
    for k, v in module.state_dict().items():
        shp = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(7, dtype=v.dtype)
        elif k.endswith("running_mean"):
            sd[k] = torch.from_numpy(rng.uniform(-0.2, 0.2, shp).astype(np.float32))
        elif k.endswith("running_var"):
            sd[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shp).astype(np.float32))
        elif v.dim() == 1 and k.endswith("weight"):
            sd[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shp).astype(np.float32))
        elif v.dim() == 1 and k.endswith("bias"):
            sd[k] = torch.from_numpy(rng.uniform(-0.2, 0.2, shp).astype(np.float32))
        else:
            fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else shp[0]
            std = math.sqrt(2.0 / max(fan_in, 1))
            sd[k] = torch.from_numpy((rng.standard_normal(shp) * std).astype(np.float32))
    return sd
