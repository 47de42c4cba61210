"""Drop-in for the 3D trunk of ``snvc.models.vernier.VernierScale`` (``BEV_type3``; r5: ``BEV_type2``, the same trunk without the coordinate head).

Module tree, attribute names and state-dict keys equal the reference's (vernier.py:249-313), so a
reference checkpoint loads with ``strict=True``.  What changes is how ``forward`` executes:

  reference (vernier.py)                              here
  --------------------------------------------------- -----------------------------------------
  _sample_2d_feat: 4 normalisation passes,            ONE gather kernel writing the concatenated
    2 grid_sample, 1 torch.cat           :323-349       [N,2F,nh,nw,nl] volume
  predict_3d_heatmaps 3D part: ~45 cuDNN / elementwise 22 fused conv launches + 3 small kernels;
    kernels                              :414-438       residual adds, ReLU, Sigmoid and the
                                                        torch.cat of :433 live in conv epilogues
  2D neck + heads (conv5, hm1, hm2, coord_head)        the same conv kernels in their depth-1 form
                                         :440-450       (SURVEY 8f N1), fused norm/bias/res/act

The HRNet backbone is outside the path.  ``get_feat_extraction`` is the
same hook the reference uses (vernier.py:837-839): assign a factory to it or
pass ``feat_net=`` to the constructor.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .submodule import (BasicBlock2d, ConvBNReLU3d, HipConv3d, basicdownsample, convbn, convbn_3d, hourglass,
                        hourglass2d, hourglass2d_downsample_16, hourglass_downsample_16)


class _VoxelGatherFn(torch.autograd.Function):
    """Differentiable feature->voxel gather (gradients w.r.t. the two feature maps; the projected
    coordinates get none, as in the reference where they come from the data loader)."""

    @staticmethod
    def forward(ctx, left, right, l_pts, r_pts, resolution):
        ctx.save_for_backward(l_pts, r_pts)
        ctx.feat_shape, ctx.resolution = tuple(left.shape), tuple(resolution)
        return ops.voxel_gather_forward(left, right, l_pts, r_pts, resolution)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad):
        l_pts, r_pts = ctx.saved_tensors
        gl, gr = ops.voxel_gather_backward(grad.contiguous(), l_pts, r_pts, ctx.feat_shape, ctx.resolution)
        return gl, gr, None, None, None


def get_feat_extraction(cfg, is_train=False, **kwargs):
    """Factory hook for the 2D backbone (reference vernier.py:835-839 builds HRNet here)."""
    if getattr(cfg, "name", None) == "identity":
        return nn.Identity()
    raise NotImplementedError(
        "the 2D backbone (HRNet) is outside the MI355X hot path; set "
        "snvc_amd.models.vernier.get_feat_extraction to the reference's factory, or pass feat_net=...")


class VernierScale(nn.Module):
    def __init__(self, cfg, is_train=False, feat_net=None):
        super().__init__()
        self.cfg = cfg
        self.is_train = is_train
        if cfg.vernier_type not in ("BEV_type3", "BEV_type2"):
            # '3D' calls an attribute its constructor never makes (vernier.py:134 hg_conv, :369 hg_conv3d) and 'BEV' is a 2D model
            raise NotImplementedError("vernier_type 'BEV_type3' (the released V-A model) and 'BEV_type2' are on the path")
        self._init_3d_net()
        self._init_grid()
        if cfg.vernier_type == "BEV_type3":       # reference vernier.py:33-36
            self._init_coord_head()
        if getattr(self.cfg, "use_bbox_head", False):
            raise NotImplementedError("use_bbox_head (FCmodel) is off by default and outside the path")
        for m in self.modules():  # reference vernier.py:38-54
            if isinstance(m, (nn.Conv3d, nn.Conv2d)):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm3d, nn.BatchNorm2d)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self.feat_net = feat_net if feat_net is not None else get_feat_extraction(
            cfg=getattr(self.cfg, self.cfg.backbone), is_train=self.is_train)

    # ------------------------------------------------------------------ construction
    def _init_coord_head(self):
        """reference vernier.py:68-93"""
        num_chan = self.cfg.num_parts
        modules = [BasicBlock2d(num_chan + 2, num_chan * 2, stride=2,
                                downsample=basicdownsample(num_chan + 2, num_chan * 2))]
        num_ds = int(4 - np.log2(192 / self.cfg.grid_resolution[2]))
        for _ in range(num_ds):
            modules.append(BasicBlock2d(num_chan * 2, num_chan * 2, stride=2,
                                        downsample=basicdownsample(num_chan * 2, num_chan * 2)))
        modules.append(nn.Conv2d(num_chan * 2, num_chan * 2, kernel_size=(6, 4)))
        modules.append(nn.Sigmoid())
        self.coord_head = nn.Sequential(*modules)

    def _init_grid(self):
        """reference vernier.py:99-114"""
        map_height, map_width = self.cfg.grid_resolution[2], self.cfg.grid_resolution[1]
        x_map = np.tile(np.linspace(0, 1, map_width), (map_height, 1)).reshape(1, 1, map_height, map_width)
        z_map = np.tile(np.linspace(0, 1, map_height).reshape(map_height, 1), (1, map_width))
        z_map = z_map.reshape(1, 1, map_height, map_width)
        self.coor_maps = torch.from_numpy(np.concatenate([x_map, z_map], axis=1).astype(np.float32))
        self.xrange = self.cfg.x_range[1] - self.cfg.x_range[0]
        self.zrange = self.cfg.z_range[1] - self.cfg.z_range[0]

    def _init_3d_net(self):
        """reference vernier.py:249-313 (BEV_type3 branch)"""
        dim = self.cfg.hrfeat.output_channel
        gn = self.cfg.gn
        num_parts = getattr(self.cfg, "num_parts", 9)
        relu = lambda: nn.ReLU(inplace=True)  # noqa: E731
        self.vimg_feat = ConvBNReLU3d(convbn_3d(2 * dim, dim, 1, 1, 0, gn=gn), relu())
        self.conv1 = ConvBNReLU3d(convbn_3d(2 * dim, dim, 7, 1, 3, gn=gn), relu())
        self.conv2 = ConvBNReLU3d(convbn_3d(dim, dim, 5, 1, 2, gn=gn), relu())
        self.conv3 = ConvBNReLU3d(convbn_3d(dim, dim, 5, 1, 4, dilation=2, gn=gn), relu())
        self.conv4 = ConvBNReLU3d(convbn_3d(2 * dim, dim, 3, 1, 1, gn=gn), relu())
        self.small = self.cfg.n_sample_w <= 16
        self.hg_conv3d = hourglass(dim, gn=gn) if self.small else hourglass_downsample_16(dim, gn=gn)
        self.fg_cls_head = nn.Sequential(convbn_3d(dim, dim, 3, 1, 1, gn=gn), relu(),
                                         HipConv3d(dim, 1, 3, 1, 1, bias=False), nn.Sigmoid())
        if getattr(self.cfg, "use_part_reg_head", False):
            self.part_reg_head = nn.Sequential(convbn_3d(dim, dim, 3, 1, 1, gn=gn), relu(),
                                               HipConv3d(dim, 27, 1, 1, 0, bias=False))
        self.pool_3d = nn.AvgPool3d((4, 1, 1), stride=(4, 1, 1))
        if self.cfg.vernier_type == "BEV_type2":      # reference vernier.py:231: convbn(dim*8, 64, ...)
            dim_height = dim * 8
        elif self.cfg.grid_resolution[0] == 32:
            dim_height = 256
        elif self.cfg.grid_resolution[0] == 16:
            dim_height = 128
        else:
            raise NotImplementedError
        self.conv5 = nn.Sequential(convbn(dim_height, 64, 3, 1, 1, 1, gn=gn), relu())
        self.hm1 = hourglass2d(64, gn=gn) if self.small else hourglass2d_downsample_16(64, gn=gn)
        self.hm2 = nn.Conv2d(64, num_parts, 3, 1, 1, bias=False)

    # ------------------------------------------------------------------ a3: feature -> voxel
    def _sample_2d_feat(self, left, right, l_pts, r_pts, aggregate="concat"):
        """reference vernier.py:323-349.  The reference normalises ``l_pts`` / ``r_pts`` IN PLACE
        through a view (:335-338); this implementation leaves the caller's tensors untouched."""
        if aggregate not in ("concat", "concat-atten"):
            raise NotImplementedError
        if aggregate == "concat-atten" and torch.is_grad_enabled() and (left.requires_grad or right.requires_grad):
            raise NotImplementedError('aggregate="concat-atten" is built for inference (construct_voxel uses "concat")')
        nh, nw, nl = self.cfg.n_sample_h, self.cfg.n_sample_w, self.cfg.n_sample_l
        if l_pts.size(2) != nh * nw * nl:
            raise RuntimeError("grid projection does not have nh*nw*nl points")
        if torch.is_grad_enabled() and (left.requires_grad or right.requires_grad):
            vox = _VoxelGatherFn.apply(left, right, l_pts, r_pts, tuple(self.cfg.resolution))
        else:
            vox = ops.voxel_gather_forward(left, right, l_pts, r_pts, self.cfg.resolution)
        if aggregate == "concat-atten":                       # :341-344
            ops.voxel_atten_scale_(vox)
        return vox.view(left.size(0), 2 * left.size(1), nh, nw, nl)

    def construct_voxel(self, left, right, grid_proj_left, grid_proj_right):
        """reference vernier.py:351-360"""
        return self._sample_2d_feat(left, right, grid_proj_left, grid_proj_right)

    # ------------------------------------------------------------------ a7: 3D trunk
    # precision: "auto" (default: the 3D trunk in split mode at inference when it qualifies), "f32" (fp32-MFMA kernels only),
    # "x3" (split mode or an error), "f16" (the fp16-STORAGE mode of BASELINE configs[4]: forward() gathers into C8 halves)
    precision = "auto"

    def construct_voxel_x3(self, left, right, grid_proj_left, grid_proj_right):
        """``construct_voxel`` written directly as the split C8 pair the split-mode trunk starts from (r4): the scale comes from
        the two feature maps' own maximum (a bilinear sample is a convex combination of feature values), so the fp32 voxel tensor,
        the pass that looked for its maximum and the layout pass are gone -- same bits as ``to_split(construct_voxel(...))`` with
        that scale.  Returns a ``SplitT``, or None when this call does not run in split mode (``trunk_3d`` decides the same way)."""
        from .submodule import SplitT
        nh, nw, nl = self.cfg.n_sample_h, self.cfg.n_sample_w, self.cfg.n_sample_l
        if (torch.is_grad_enabled() or not left.is_cuda or left.dtype != torch.float32 or grid_proj_left.size(2) != nh * nw * nl
                or self._x3_local(None, left.device, 2 * left.size(1)) is None):
            return None
        mul = ops.split_scale_for(left, right)
        try:
            vox = ops.voxel_gather_forward_split(left, right, grid_proj_left, grid_proj_right, self.cfg.resolution, mul)
        except ops.Unsupported:
            return None
        return SplitT(vox.view(left.size(0), 2, 2 * left.size(1) // 8, nh, nw, nl, 8), 0, None, mul)

    def _x3_local(self, voxel, device=None, channels=None):
        """The split-mode bookkeeping of this model (overflow flag + its pinned host copy) if this call runs the trunk in split
        mode, else None: inference only, every norm a frozen BatchNorm3d or (r5) a GroupNorm, channels a multiple of 32, no part_reg_head.
        ``voxel``: the fp32 gather result, an already split ``SplitT`` (``construct_voxel_x3``), or None with ``device`` /
        ``channels`` given (the question asked before the gather runs)."""
        from .submodule import SplitT, x3_ok
        mode = getattr(self, "precision", "auto")
        want = mode == "x3"
        if mode in ("f32", "f16"):
            return None
        if self.__dict__.get("_snvc_x3_off"):
            if want:
                raise RuntimeError("precision='x3': split mode was switched off by an earlier overflow (reset_split_mode())")
            return None
        if isinstance(voxel, SplitT):
            device, channels, shape_ok = voxel.t.device, 8 * voxel.t.size(2), voxel.t.dim() == 7
        elif voxel is not None:
            device, channels = voxel.device, voxel.size(1)
            shape_ok = voxel.is_cuda and voxel.dtype == torch.float32 and voxel.dim() == 5
        else:
            shape_ok = device.type == "cuda"
        ok = (shape_ok and not hasattr(self, "part_reg_head")
              and channels % 64 == 0 and not self.training
              and x3_ok(self.vimg_feat, self.conv1, self.conv2, self.conv3, self.conv4, self.hg_conv3d, self.fg_cls_head))
        if not ok:
            if want:
                raise RuntimeError("precision='x3': the trunk does not qualify (inference, eval-mode BatchNorm3d or GroupNorm, 2F % 64 == 0)")
            return None
        from .submodule import overflow_guard
        guard = overflow_guard(self, device)
        if guard.event is not None and self._x3_overflowed(guard, "an earlier call's result clamped it"):
            return None                               # overflow_check = "deferred": the previous call's flag
        return guard

    # "call" (default, r5): the overflow flag is read before the trunk's results leave the call, and a flagged call is redone on
    # the fp32-MFMA kernels; "deferred": the flag is only posted (check_overflow() or the next call looks) -- see GlobalStack
    overflow_check = "call"

    def _x3_overflowed(self, guard, what):
        if not guard.wait():
            return False
        return self._leave_split_mode(what)

    def _leave_split_mode(self, what):
        import warnings
        from .submodule import X3_SIGMAS
        self.__dict__["_snvc_x3_off"] = True
        msg = ("snvc_amd: split-mode (f16x3) overflow in the local trunk -- an activation exceeded |beta| + %g |gamma| of its "
               "BatchNorm; %s.  This model now runs on the fp32-MFMA kernels (reset_split_mode() turns split mode back on)."
               % (X3_SIGMAS, what))
        if getattr(self, "precision", "auto") == "x3":
            raise RuntimeError("precision='x3': " + msg)
        warnings.warn(msg)
        return True

    def check_overflow(self) -> bool:
        """With ``overflow_check = "deferred"``: wait for the last split-mode call's flag; True if its result was clamped."""
        hit = False
        for guard in self.__dict__.get("_snvc_x3_guard", {}).values():
            hit |= self._x3_overflowed(guard, "the last result clamped it")
        return hit

    def reset_split_mode(self):
        self.__dict__.pop("_snvc_x3_off", None)

    def trunk_3d_x3(self, voxel, st):
        """``trunk_3d`` (reference vernier.py:415-438) in split mode: the same fp32 layers, every product three
        half-precision MFMAs on (hi, lo) pairs with fp32 accumulation.  ``voxel`` is the fp32 gather result; it is scaled by a
        power of two derived from its own maximum on the device (no host round trip) and split once; every later tensor's
        exponent comes from its folded BatchNorm.  Returns the same float32 tensors as ``trunk_3d``."""
        from .submodule import SplitT, x3_exponent, x3_norm_bound, _Plan
        guard, flag = st, st.flag
        if isinstance(voxel, SplitT):            # construct_voxel_x3: the gather wrote the pair itself
            vs = voxel
            voxel = vs.t
            n, c2 = voxel.size(0), 8 * voxel.size(2)
        else:
            n, c2 = voxel.size(0), voxel.size(1)
            mul = ops.split_scale_for(voxel)
            vs = SplitT(ops.to_split(voxel, mul_dev=mul), 0, None, mul)
        g = c2 // 16                                                             # channel groups of F channels

        def nb(seq):        # the bound of a ConvBN3d's result
            conv, norm = seq[0], seq[1]
            return x3_norm_bound(norm, conv.__dict__.setdefault("_snvc_plans_x3", {}).setdefault(voxel.device, _Plan()))
        # the two halves of the concat of :433 share one exponent: both bounds are known from the parameters alone
        b_v3 = nb(self.conv1[0]) + nb(self.conv2[0]) + nb(self.conv3[0])
        last = self.hg_conv3d.conv6 if self.small else self.hg_conv3d.conv12
        e_cat = x3_exponent(max(nb(last) + b_v3, nb(self.vimg_feat[0])))
        img = self.vimg_feat.fused_x3(vs, out_exp=e_cat, flag=flag)             # :415
        v = self.conv1.fused_x3(vs, flag=flag)                                   # :417
        v = self.conv2.fused_x3(v, residual=v, residual_after_act=True, flag=flag)      # conv2(v) + v   :418
        v = self.conv3.fused_x3(v, residual=v, residual_after_act=True, flag=flag)      # conv3(v) + v   :419
        cat = torch.empty((n, 2, 2 * g) + tuple(v.t.shape[3:]), dtype=torch.float16, device=voxel.device)
        dst = cat[:, :, :g]
        if self.small:                                                           # :420-423
            vh, _, _ = self.hg_conv3d.forward_x3(v, residual=v, out=dst, out_exp=e_cat, flag=flag)
        else:
            vh = self.hg_conv3d.forward_x3(v, residual=v, out=dst, out_exp=e_cat, flag=flag)
        t = self.fg_cls_head[0].fused_x3(vh, relu=True, flag=flag)               # :427
        guard.post()        # the last layer that can clamp (the ones below write float32 / multiply by occ in [0, 1])
        occ = self.fg_cls_head[2].fused_x3(t, sigmoid=True)                      # float32 [N,1,nh,nw,nl]
        ops.mul_broadcast_split(img.t, occ, out=cat[:, :, g:])                   # cat([v, img * occ])  :433
        v = self.conv4.fused_x3(SplitT(cat, e_cat, max(vh.bound, img.bound)), to_f32=True)      # :435, float32 NCDHW
        v = ops.avgpool_depth4(v)                                                # :436
        from .submodule import _ROUTES, SplitOverflow
        if self.overflow_check == "call" and guard.wait():      # waited for while the last three layers still run
            raise SplitOverflow()
        _ROUTES["x3_local_trunk"] += 1
        return v.reshape(n, -1, v.size(3), v.size(4)), occ, None                 # :437-438

    def trunk_3d(self, voxel):
        """reference vernier.py:415-438 -> (voxel_BEV [N, F*nh/4, nw, nl], occupancy [N,1,nh,nw,nl], offset)."""
        from .submodule import SplitT
        if not torch.is_grad_enabled():
            st = self._x3_local(voxel)
            if st is not None:
                from .submodule import SplitOverflow, _ROUTES
                try:
                    return self.trunk_3d_x3(voxel, st)
                except SplitOverflow:       # this call's clamped result is dropped; the fp32 layers below redo it
                    self._leave_split_mode("this call was redone in fp32")
                    _ROUTES["x3_overflow_redo"] += 1
        if isinstance(voxel, SplitT):            # split by construct_voxel_x3, but the trunk has left split mode since (overflow flag)
            voxel = ops.from_split(voxel.t) / voxel.mul_dev
        n, c2 = voxel.size(0), voxel.size(1)
        f = c2 // 2
        training_graph = torch.is_grad_enabled() and (voxel.requires_grad or any(p.requires_grad for p in self.parameters()))
        img = self.vimg_feat(voxel)                                             # :415
        v = self.conv1(voxel)                                                   # :417
        v = self.conv2.fused(v, residual=v, residual_after_act=True)            # conv2(v) + v   :418
        v = self.conv3.fused(v, residual=v, residual_after_act=True)            # conv3(v) + v   :419
        # inference: the torch.cat of :433 is built in place (producers write channel slices);
        # under autograd the slices become ordinary tensors and the glue ops are torch's own
        cat = None if training_graph else torch.empty((n, 2 * f) + tuple(v.shape[2:]), dtype=v.dtype, device=v.device)
        dst = None if training_graph else cat[:, :f]
        if self.small:                                                          # :420-423
            v, _, _ = self.hg_conv3d(v, None, None, residual=v, out=dst)
        else:
            v = self.hg_conv3d(v, residual=v, out=dst)
        t = self.fg_cls_head[0].fused(v, relu=True)                             # :427
        occ = self.fg_cls_head[2].fused(t, sigmoid=True)
        offset = None
        if hasattr(self, "part_reg_head"):                                      # :428-431
            offset = self.part_reg_head[2](self.part_reg_head[0].fused(v, relu=True))
        if training_graph:
            cat = torch.cat([v, img * occ], dim=1)                              # :433
            v = self.conv4(cat)                                                 # :435
            v = ops.AvgPoolDepth4Fn.apply(v) if v.is_cuda and v.dtype == torch.float32 else torch.nn.functional.avg_pool3d(v, (4, 1, 1), (4, 1, 1))   # :436
        else:
            ops.mul_broadcast(img, occ, out=cat[:, f:])                         # cat([v, img*occ])  :433
            from .submodule import fused_conv3d_avgpool_d4
            v = fused_conv3d_avgpool_d4(self.conv4[0][0], self.conv4[0][1], cat, relu=True)    # :435-436, one launch
        return v.reshape(n, -1, v.size(3), v.size(4)), occ, offset              # :437-438

    # ------------------------------------------------------------------ fp16-storage mode (BASELINE configs[4])
    def construct_voxel_f16(self, left, right, grid_proj_left, grid_proj_right):
        """``construct_voxel`` with a C8 half result ``[N, 2F/8, nh, nw, nl, 8]`` (``ops.to_c8`` layout): the fp32
        feature maps are sampled exactly as in the fp32 path and rounded to half once."""
        nh, nw, nl = self.cfg.n_sample_h, self.cfg.n_sample_w, self.cfg.n_sample_l
        if grid_proj_left.size(2) != nh * nw * nl:
            raise RuntimeError("grid projection does not have nh*nw*nl points")
        vox = ops.voxel_gather_forward_f16(left, right, grid_proj_left, grid_proj_right, self.cfg.resolution)
        return vox.view(left.size(0), 2 * left.size(1) // 8, nh, nw, nl, 8)

    def trunk_3d_f16(self, voxel):
        """``trunk_3d`` (reference vernier.py:415-438) with half storage: ``voxel`` is a C8 tensor; activations and
        weights are half in HBM, accumulation and epilogues fp32.  Returns the same float32 tensors as ``trunk_3d``
        (voxel_BEV, occupancy, offset)."""
        if hasattr(self, "part_reg_head"):
            raise NotImplementedError("part_reg_head is not built for the fp16-storage mode")
        n, g2 = voxel.size(0), voxel.size(1)
        g = g2 // 2
        img = self.vimg_feat.fused_f16(voxel)                                   # :415
        v = self.conv1.fused_f16(voxel)                                         # :417
        v = self.conv2.fused_f16(v, residual=v, residual_after_act=True)        # :418
        v = self.conv3.fused_f16(v, residual=v, residual_after_act=True)        # :419
        cat = torch.empty((n, g2) + tuple(v.shape[2:]), dtype=v.dtype, device=v.device)   # the cat of :433, in place
        if self.small:                                                          # :420-423
            v, _, _ = self.hg_conv3d.forward_f16(v, residual=v, out=cat[:, :g])
        else:
            v = self.hg_conv3d.forward_f16(v, residual=v, out=cat[:, :g])
        t = self.fg_cls_head[0].fused_f16(v, relu=True)                         # :427
        occ = self.fg_cls_head[2].fused_f16(t, sigmoid=True)                    # float32 [N,1,nh,nw,nl]
        ops.mul_broadcast_c8(img, occ, out=cat[:, g:])                          # :433
        v = self.conv4.fused_f16(cat)                                           # :435
        v = ops.avgpool_depth4_c8(v)                                            # :436 -> float32 [N,F,nh/4,nw,nl]
        return v.reshape(n, -1, v.size(3), v.size(4)), occ, None                # :437-438

    def heads_2d(self, voxel_BEV):
        """reference vernier.py:440-450.  Inference with eval-mode BatchNorm runs every convolution of the neck
        (conv5, hm1, hm2, the coordinate head's blocks and its last full-extent layer) on the depth-1 HIP kernels
        with fused norm / bias / residual / ReLU / Sigmoid epilogues; otherwise the modules' torch forward."""
        from .submodule import _hip_2d_ok, _cbr2d, fused_conv2d
        type2 = self.cfg.vernier_type == "BEV_type2"
        if type2 and self.small:      # reference vernier.py:394: `self.hg_conv3d(voxel) + voxel` -- the plain hourglass returns a tuple there
            raise NotImplementedError("vernier_type='BEV_type2' with n_sample_w <= 16 fails in the reference too (vernier.py:394, :409)")
        hip = _hip_2d_ok(voxel_BEV, self.conv5, self.hm1, self.hm2, *(() if type2 else (self.coord_head,)))
        voxel_BEV = _cbr2d(self.conv5, voxel_BEV) if hip else self.conv5(voxel_BEV)
        feats = self.hm1(voxel_BEV, None, None)[0] if self.small else self.hm1(voxel_BEV)
        if hip:      # hm2(feats.permute(0, 1, 3, 2)) without copying the 64-channel tensor: swapped kernel, transposed view out
            heatmaps = fused_conv2d(self.hm2, None, feats, transposed_input=True)
        else:
            heatmaps = self.hm2(feats.permute(0, 1, 3, 2))
        if type2:                     # reference vernier.py:396-410: heat maps and occupancy only
            return heatmaps, None
        num_sample = len(heatmaps)
        # the coordinate maps are a plain attribute in the reference (not in the state dict): one device copy is kept,
        # so that no host -> device copy sits in the middle of the neck (and of a captured graph)
        cm = self.__dict__.get("_snvc_coor_maps")
        if cm is None or cm.device != heatmaps.device:
            cm = self.__dict__["_snvc_coor_maps"] = self.coor_maps.to(heatmaps.device)
        coor_maps = cm.expand(num_sample, -1, -1, -1) if cm.size(0) == 1 else cm.repeat(num_sample, 1, 1, 1)
        augmented_maps = torch.cat([heatmaps, coor_maps], dim=1)
        last = self.coord_head[-2]
        if hip and tuple(last.kernel_size) != (1, 1):
            t = augmented_maps
            for blk in self.coord_head[:-2]:
                t = blk(t)                                                          # BasicBlock2d: HIP forward
            if tuple(t.shape[2:]) == tuple(last.kernel_size):
                coordinates = fused_conv2d(last, None, t.contiguous(), sigmoid=True)   # Conv2d((6,4)) + Sigmoid  :87-88
            else:
                coordinates = self.coord_head[-1](last(t))
            coordinates = coordinates.reshape(num_sample, -1, 2)
        else:
            coordinates = self.coord_head(augmented_maps).view(num_sample, -1, 2)
        return heatmaps, coordinates

    def predict_3d_heatmaps(self, voxel, depth=None):
        """reference vernier.py:362-458 -> (heatmaps, occupancy, offset, coordinates, bbox)"""
        if depth is not None:
            raise NotImplementedError
        from .submodule import SplitT
        if not isinstance(voxel, SplitT) and voxel.dtype == torch.float16:        # a C8 tensor from construct_voxel_f16: fp16-storage mode
            voxel_BEV, occupancy, offset = self.trunk_3d_f16(voxel)
        else:
            voxel_BEV, occupancy, offset = self.trunk_3d(voxel)
        heatmaps, coordinates = self.heads_2d(voxel_BEV)
        return heatmaps, occupancy.squeeze(1), offset, coordinates, None

    def forward(self, left_roi, right_roi, grid_proj_left, grid_proj_right, meta_data=None, test=False):
        """reference vernier.py:460-555"""
        left_feat = self.feat_net(left_roi)
        right_feat = self.feat_net(right_roi)
        if test:
            self.last_self_check = self._aggregation_self_check(left_feat, right_feat, grid_proj_left, grid_proj_right, meta_data)
        if getattr(self, "precision", "f32") == "f16" and not torch.is_grad_enabled():
            voxels = self.construct_voxel_f16(left_feat, right_feat, grid_proj_left, grid_proj_right)
        else:
            voxels = self.construct_voxel_x3(left_feat, right_feat, grid_proj_left, grid_proj_right)    # split mode: a (hi, lo) pair
            if voxels is None:
                voxels = self.construct_voxel(left_feat, right_feat, grid_proj_left, grid_proj_right)
        ncf, occupancy, part_offsets, coordinates, bboxes = self.predict_3d_heatmaps(voxels)
        return {"ncf": ncf, "occupancy": occupancy, "coordinates": coordinates}

    def _aggregation_self_check(self, left_feat, right_feat, grid_proj_left, grid_proj_right, meta_data, voxel=None):
        """The numeric half of the reference's ``forward(test=True)`` (vernier.py:479-519; its matplotlib half is not
        reproduced): one voxel (i, j, k) of sample 0 is re-projected on the host from ``meta_data['grid_3d']`` through
        the sample's calibration objects and crop transforms, compared with the projected grid that was fed in, and the
        voxel's aggregated feature is compared with the two feature maps at the ROUNDED pixel (nearest neighbour against
        the kernel's bilinear sample, as in the reference: an eyeball check).  Returns / stores the numbers it prints."""
        import numpy as np
        nh, nw, nl = self.cfg.n_sample_h, self.cfg.n_sample_w, self.cfg.n_sample_l
        i, j, k = voxel if voxel is not None else (np.random.randint(0, nh), np.random.randint(0, nw), np.random.randint(0, nl))
        n, down = len(left_feat), 4.0
        p3 = np.asarray(meta_data["grid_3d"][0]).reshape(nh, nw, nl, 3)[i, j, k].reshape(1, 3)

        def to_crop(calib, trans):            # img_proc.affine_transform (:71-74) of the projected point
            p2 = np.asarray(calib.project_rect_to_image(p3), dtype=np.float64).reshape(1, 2)
            return (np.asarray(trans, dtype=np.float64) @ np.concatenate([p2, np.ones((1, 1))], axis=1).T).astype(np.float32) / down

        cl = to_crop(meta_data["calib_left"][0], meta_data["trans_l"][0])
        cr = to_crop(meta_data["calib_right"][0], meta_data["trans_r"][0])
        gpl = meta_data["grid_proj_left"].reshape(n, 2, nh, nw, nl)[0, :, i, j, k].detach().cpu().numpy()
        gpr = meta_data["grid_proj_right"].reshape(n, 2, nh, nw, nl)[0, :, i, j, k].detach().cpu().numpy()
        voxels = self.construct_voxel(left_feat[:1], right_feat[:1], grid_proj_left[:1], grid_proj_right[:1])
        xl, yl = int(np.round(cl[0, 0])), int(np.round(cl[1, 0]))
        xr, yr = int(np.round(cr[0, 0])), int(np.round(cr[1, 0]))
        f3d = voxels[0, :, i, j, k]
        dif = torch.abs(f3d - torch.cat((left_feat[0, :, yl, xl], right_feat[0, :, yr, xr])))
        out = {"voxel": (i, j, k), "projection_error_left": cl[:, 0] * down - gpl, "projection_error_right": cr[:, 0] * down - gpr,
               "feature_abs_diff": dif.detach().cpu(), "voxel_feature": f3d.detach().cpu()}
        print(out["projection_error_left"]); print(out["projection_error_right"]); print(dif); print(f3d)
        return out

    # ------------------------------------------------------------------ a12: index extraction
    def ncf_argmax(self, ncf):
        """np.argmax(ncf.reshape(N, parts, -1), axis=2) of ncf_to_update_2d / ncf_to_offset
        (reference vernier.py:570-572,693) on the device; returns (int64 indices, confidences)."""
        n, p = ncf.shape[0], ncf.shape[1]
        idx, val = ops.argmax_rows(ncf.reshape(n * p, -1))
        return idx.view(n, p), val.view(n, p)


    # ------------------------------------------------------------------ N4: decode
    def ncf_to_update_2d(self, ncf, samples, grid, filter_3d, arg_max="hard", coordinates=None):
        """reference vernier.py:665-738 (see snvc_amd.decode.ncf_to_update_2d: max / argmax / range test on the device,
        the 9-point pose fit on the host in float64 like the reference)."""
        from .. import decode
        return decode.ncf_to_update_2d(self.cfg, ncf, samples, grid, filter_3d, arg_max=arg_max, coordinates=coordinates)


def get_model(cfgs, is_train=False):
    """reference vernier.py:841-842"""
    return VernierScale(cfgs, is_train)
