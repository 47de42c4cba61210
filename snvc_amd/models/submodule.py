"""Drop-in for the 3D building blocks of ``snvc.models.submodule``.

Same constructors, same module tree and therefore the same ``state_dict`` keys as the
reference (``load_state_dict(strict=True)`` of a reference checkpoint works,
tools/inference_agnostic.py:452), but ``forward`` runs hand-written HIP kernels:

  reference (submodule.py)                         here
  ------------------------------------------------ ------------------------------------------
  convbn_3d                     :32-50             ConvBN3d(nn.Sequential): ONE fused launch
                                                   conv + folded BatchNorm (+ReLU/+residual)
  hourglass                     :85-168            6 fused launches (4 conv, 2 deconv)
  get_hg_down_sample            :170-181           Sequential(ConvBN3d, ReLU) -> 1 launch
  get_hg_up_sample              :197-208           ConvBN3d over ConvTranspose3d -> 1 launch
  hourglass_downsample_16       :223-268           12 fused launches, skips added in the epilogue
  disparityregression           :76-83             1 reduction kernel

The nn.Conv3d / nn.BatchNorm3d / nn.GroupNorm children only HOLD parameters; their own
``forward`` is never called.  GroupNorm and train-mode BatchNorm need statistics of the conv
output, so they take three launches (conv, statistics, normalise+activation).

The 2D helpers at the bottom (convbn, hourglass2d, ...) keep torch.nn's
module tree; their inference forward runs on the depth-1 form of the same HIP conv kernels
(``fused_conv2d`` / ``fused_deconv2d``); under autograd / with train-mode BatchNorm the same kernels sit behind
``_Conv2dNormActFn`` (r4: HIP backward -- epilogue reductions, data gradient as a twin layer, deterministic weight gradient).
"""
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..ops import EPI_ADD_POST, EPI_ADD_PRE, EPI_RELU, EPI_SIGMOID


def _first_arg(k):
    return k[0] if isinstance(k, (tuple, list)) else k


def _cubic(v, what):
    if isinstance(v, (tuple, list)):
        if len(set(v)) != 1:
            raise NotImplementedError(f"non-cubic {what} {tuple(v)} is not on the hot path")
        return int(v[0])
    return int(v)


_GENERATION = [0]   # bumped by invalidate_plans(): part of every cache key

# How often each fused route was taken (tests assert on it: a silent regression to a slower or to a torch route shows)
import collections
import math
_ROUTES = collections.Counter()
TRAIN_EXACT_K57 = [False]     # True: k5 / k7 layers under autograd on the direct kernels (exact fp32 FMA chains) instead of Winograd


def invalidate_plans(module: Optional[nn.Module] = None) -> None:
    """Drop the packed-weight / folded-BatchNorm / dgrad / factored-conv caches.

    The caches are keyed on ``(data_ptr, tensor._version, device)``.  Every in-place update that goes
    through autograd-visible tensors (``optimizer.step()``, ``load_state_dict``, ``p.copy_()`` under
    ``no_grad``, ``p.detach().mul_()``) bumps ``_version`` and invalidates them by itself.  Writes through
    ``.data`` (``p.data.mul_(2)``, an EMA swap that assigns ``.data`` storage in place) do NOT bump it: after
    such a write call this function -- for one module tree, or with no argument for every model of the
    process (a generation counter in the keys)."""
    if module is None:
        _GENERATION[0] += 1
        return
    for m in module.modules():
        for name in CACHE_ATTRS:
            m.__dict__.pop(name, None)


# Every per-module cache this package hangs on a module's __dict__ (packed weights, folded norms, the factored first
# convolution's split weights for inference and training, the persistent inference workspace, the device copy of the
# coordinate maps).  invalidate_plans() drops exactly these; tests/test_host_cpu.py checks that no other `_snvc_*`
# name is written anywhere in the package.
CACHE_ATTRS = ("_snvc_plans", "_snvc_plans_f16", "_snvc_plans_x3", "_snvc_plans2d", "_snvc_plans2d_t", "_snvc_factored", "_snvc_factored_train", "_snvc_ws",
               "_snvc_coor_maps", "_snvc_x3", "_snvc_x3_off", "_snvc_x3_guard", "_snvc_last_v1", "_snvc_streams", "_snvc_lazy_warned", "_snvc_prep_ws", "_snvc_prep_epoch")


class _Plan:
    """Packed weights + folded affine for one conv(+norm) pair, rebuilt when parameters change."""

    def __init__(self):
        self.key = None
        self.layer: Optional[ops.Conv3dLayer] = None
        self.scale = None
        self.bias = None


def _conv_geometry(conv: nn.Module):
    transposed = isinstance(conv, nn.ConvTranspose3d)
    k = _cubic(conv.kernel_size, "kernel")
    s = _cubic(conv.stride, "stride")
    p = _cubic(conv.padding, "padding")
    d = _cubic(conv.dilation, "dilation")
    if conv.bias is not None:
        raise NotImplementedError("conv bias is not on the path (all 3D convs use bias=False)")
    if conv.groups != 1:
        raise NotImplementedError("grouped 3D convolutions are not on the path")
    if transposed and _cubic(conv.output_padding, "output_padding") != 1:
        raise NotImplementedError("ConvTranspose3d on the path always has output_padding=1")
    return k, s, p, d, transposed


def _get_layer(conv: nn.Module, plan: _Plan) -> ops.Conv3dLayer:
    w = conv.weight
    key = (w.data_ptr(), w._version, w.device, _GENERATION[0])
    if plan.layer is None or plan.key != key:
        k, s, p, d, transposed = _conv_geometry(conv)
        plan.layer = ops.Conv3dLayer(w.detach(), k, s, p, d, transposed)
        plan.key = key
    return plan.layer


def _folded_bn(bn: nn.BatchNorm3d, plan: _Plan):
    """Eval-mode BatchNorm3d as y = x*scale + bias (fp64 fold, cast once)."""
    key = (bn.weight._version if bn.weight is not None else -1, bn.bias._version if bn.bias is not None else -1,
           bn.running_mean._version, bn.running_var._version, bn.running_mean.data_ptr(), bn.running_mean.device,
           _GENERATION[0])
    if plan.scale is None or getattr(plan, "bn_key", None) != key:
        var = bn.running_var.detach().double()
        mean = bn.running_mean.detach().double()
        g = bn.weight.detach().double() if bn.weight is not None else torch.ones_like(var)
        b = bn.bias.detach().double() if bn.bias is not None else torch.zeros_like(var)
        sc = g / torch.sqrt(var + bn.eps)
        plan.scale = sc.float().contiguous()
        plan.bias = (b - mean * sc).float().contiguous()
        plan.bn_key = key
    return plan.scale, plan.bias


_BatchNormNd = (nn.BatchNorm3d, nn.BatchNorm2d)      # the 2D neck's layers run as depth-1 3D layers (raw is 5-D either way)


def _norm_from_raw(raw, norm, plan, residual, flags, out=None):
    """norm + residual + activation of an already computed conv output `raw` (kept for the backward pass):
    returns (y, scale, shift, mean, var, per_sample)."""
    if norm is None:
        if flags or residual is not None:
            return ops.affine_act(raw, None, None, residual, flags, out=out), None, None, None, None, False
        return raw, None, None, None, None, False
    if isinstance(norm, _BatchNormNd) and not (norm.training or norm.running_mean is None):
        scale, bias = _folded_bn(norm, plan)
        return ops.affine_act(raw, scale, bias, residual, flags, out=out), scale, bias, None, None, False
    c = raw.size(1)
    if isinstance(norm, nn.GroupNorm):
        scale, shift, mean, var = ops.norm_stats(raw, norm.weight, norm.bias, norm.num_groups, True, norm.eps)
        return ops.affine_act(raw, scale, shift, residual, flags, per_sample=True, out=out), scale, shift, mean, var, True
    if isinstance(norm, _BatchNormNd):
        scale, shift, mean, var = ops.norm_stats(raw, norm.weight, norm.bias, c, False, norm.eps)
        _bn_track(norm, mean, var, raw.numel() / c)      # nn.BatchNorm3d bookkeeping: momentum update with the unbiased variance
        return ops.affine_act(raw, scale, shift, residual, flags, per_sample=False, out=out), scale, shift, mean, var, False
    raise NotImplementedError(f"norm layer {type(norm).__name__} is not on the path")


def _norm_forward(layer, norm, plan, x, residual, flags, out, keep_raw, exact=False, twin=None):
    """Shared forward: returns (y, raw, scale, shift, mean, var, per_sample).  `raw` is the conv
    output before the affine/activation (None when the single fused launch was used).
    ``keep_raw`` (the autograd functions): the pass that writes y also leaves max|y| in a device word tagged onto y (ops.tag_amax):
    the next layer's split-operand weight gradient scales its x operand by it (r6).  ``twin`` (r6, see _twin_mul_forward): that pass
    also writes y's split C8 twin for a consumer that runs on the split kernels."""
    am = ops.amax_word(x.device) if (keep_raw and x.is_cuda) else None
    res = _norm_forward_impl(layer, norm, plan, x, residual, flags, out, keep_raw, exact, am, twin)
    if am is not None and res[1] is not None and res[0] is not res[1]:      # y came out of affine_act (not the raw tensor itself)
        ops.tag_amax(res[0], am)
    return res


def _twin_mul_forward(twin, scale, shift, residual, c, device):
    """The scale of y's split twin when the layer's consumer asked for one (``twin`` = (L1 norms of the filters per output channel,
    the words of max|x|)): from the bound  max_c(|scale_c| * L1_c * max|x| + |shift_c|) + max|residual|  -- |raw_c| <= L1_c * max|x| --
    in one small launch, or None when a maximum is not known on the device."""
    if twin is None or twin[1] is None:
        return None
    ar = None
    if residual is not None:
        ar = ops.amax_of(residual)
        if ar is None:
            return None
    rows = scale.numel() if scale is not None else c
    return ops.split_scale_bound(rows, c, device, b=scale, l1=twin[0], amax_x=twin[1], cc=shift, amax_r=ar)


def _norm_forward_impl(layer, norm, plan, x, residual, flags, out, keep_raw, exact, am, twin=None):
    _affine_act = ops.affine_act
    if twin is not None:
        def _affine_act(raw, scale, shift, residual=None, flags=0, per_sample=False, out=None, amax=None):
            ok = ops.twin_ok(raw) and not (flags & EPI_SIGMOID)           # a sigmoid's result is not bounded by its argument's bound
            tm = _twin_mul_forward(twin, scale, shift, residual, raw.size(1), raw.device) if ok else None
            return ops.affine_act(raw, scale, shift, residual, flags, per_sample=per_sample, out=out, amax=amax, twin_mul=tm)
    if norm is None:
        if keep_raw and (flags or residual is not None):
            raw = layer(x, None, None, None, 0, None, exact=exact)
            return _affine_act(raw, None, None, residual, flags, out=out, amax=am), raw, None, None, None, None, False
        y = layer(x, None, None, residual, flags, out, exact=exact)
        return y, (y if keep_raw else None), None, None, None, None, False
    if isinstance(norm, nn.BatchNorm3d) and not (norm.training or norm.running_mean is None):
        scale, bias = _folded_bn(norm, plan)
        if keep_raw:
            raw = layer(x, None, None, None, 0, None, exact=exact)
            return _affine_act(raw, scale, bias, residual, flags, out=out, amax=am), raw, scale, bias, None, None, False
        return layer(x, scale, bias, residual, flags, out, exact=exact), None, scale, bias, None, None, False
    # statistics of the conv output are needed first: conv -> stats -> normalise (+res, +act)
    if isinstance(norm, nn.BatchNorm3d) and not exact:
        # the 3x3x3 Winograd kernels take the batch statistics in their own epilogue (one read of the tensor less)
        got = layer.forward_stats(x, norm.weight.detach() if norm.weight is not None else None,
                                  norm.bias.detach() if norm.bias is not None else None, norm.eps)
        if got is not None:
            raw, scale, shift, mean, var = got
            _ROUTES["conv_stats_epilogue"] += 1
            _bn_track(norm, mean, var, raw.numel() / raw.size(1))
            dst = out if out is not None else (None if keep_raw else raw)
            return _affine_act(raw, scale, shift, residual, flags, per_sample=False, out=dst, amax=am), raw, scale, shift, mean, var, False
    raw = layer(x, None, None, None, 0, None, exact=exact)
    c = raw.size(1)
    dst = out if out is not None else (None if keep_raw else raw)
    if isinstance(norm, nn.GroupNorm):
        scale, shift, mean, var = ops.norm_stats(raw, norm.weight, norm.bias, norm.num_groups, True, norm.eps)
        return _affine_act(raw, scale, shift, residual, flags, per_sample=True, out=dst, amax=am), raw, scale, shift, mean, var, True
    if isinstance(norm, nn.BatchNorm3d):
        scale, shift, mean, var = ops.norm_stats(raw, norm.weight, norm.bias, c, False, norm.eps)
        _bn_track(norm, mean, var, raw.numel() / c)      # nn.BatchNorm3d bookkeeping: momentum update with the unbiased variance
        return _affine_act(raw, scale, shift, residual, flags, per_sample=False, out=dst, amax=am), raw, scale, shift, mean, var, False
    raise NotImplementedError(f"norm layer {type(norm).__name__} is not on the path")


def _dgrad_layer(conv: nn.Module, plan: _Plan) -> ops.Conv3dLayer:
    """The data gradient of a layer is another layer of the same kernel family:
    Conv3d stride 1  -> Conv3d with taps flipped and channels transposed;
    Conv3d(k3,s2,p1) -> ConvTranspose3d(k3,s2,p1,op1) over the SAME weight memory;
    ConvTranspose3d  -> Conv3d(k3,s2,p1) over the SAME weight memory."""
    w = conv.weight
    key = (w.data_ptr(), w._version, w.device, _GENERATION[0])
    if getattr(plan, "dgrad", None) is None or plan.dgrad_key != key:
        k, s, p, d, transposed = _conv_geometry(conv)
        wd = w.detach()
        if transposed:
            plan.dgrad = ops.Conv3dLayer(wd, 3, 2, 1, 1, False)
        elif s == 1:
            plan.dgrad = ops.Conv3dLayer(wd.transpose(0, 1).flip(2, 3, 4).contiguous(), k, 1, p, d, False)
        elif s == 2 and k == 3 and p == 1 and d == 1:
            plan.dgrad = ops.Conv3dLayer(wd, 3, 2, 1, 1, True)
        else:
            raise NotImplementedError("dgrad of this strided convolution is not on the path")
        plan.dgrad_key = key
    return plan.dgrad


# ---------------------------------------------------------------------------------------------------------------------------------
# r6: the training step's forward and data-gradient convolutions on the split kernels.
# A convolution's MFMA operand is 8 channels of one voxel (split C8 pairs); training keeps float32 NCDHW for the weight gradients, the
# statistics and autograd.  Both exist side by side: the pass that writes a tensor (affine_act forward, act_backward_apply backward)
# also writes its split TWIN when the consumer runs on the split kernels (+4 bytes per element on a pass that moves 8-12, against a
# layout pass of 8), scaled by a power of two from an upper bound of the tensor's maximum that is known before the pass runs
# (_twin_mul_forward / _epilogue_backward); the split kernels write float32 NCDHW (y_f32).  Measured per layer in
# profiles/r6/kernel_experiments_r6.txt.  Which layers: every 3x3x3 layer with whole 32-channel blocks on both sides (conv2 and the
# hourglass); the first layer has its own sheared route, the classifier (one output channel) stays on the fp32 kernels.
X3_TRAIN = [True]          # tools / tests flip this
X3_TRAIN_MIN_CC = [1024]   # Cin * Cout from which a layer takes the route (measured: 2048 -> 1024 = conv2 as well: 15.70 -> 15.25 ms)


def _x3_train_route(conv: nn.Module, x: torch.Tensor) -> bool:
    if not (X3_TRAIN[0] and x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and x.numel() > 0):
        return False
    if conv.bias is not None or conv.groups != 1:
        return False
    k, st, p, d, transposed = _conv_geometry(conv)
    cin, cout = conv.in_channels, conv.out_channels
    if k != 3 or p != 1 or d != 1 or st not in (1, 2) or (transposed and st != 2):
        return False
    if cin % 32 or cout % 32 or cin * cout < X3_TRAIN_MIN_CC[0]:
        return False
    if st == 2 and not transposed and any(int(e) % 2 for e in x.shape[2:]):
        return False                                          # odd extents: the fp32 route crops / pads
    return True


def _l1_out(conv: nn.Module, plan: _Plan) -> torch.Tensor:
    """sum |w| per OUTPUT channel (cached per weight version): |conv(x)[c]| <= l1[c] * max|x|."""
    w = conv.weight
    key = (w.data_ptr(), w._version, w.device, _GENERATION[0])
    if getattr(plan, "l1", None) is None or plan.l1_key != key:
        dims = (0, 2, 3, 4) if isinstance(conv, nn.ConvTranspose3d) else (1, 2, 3, 4)
        plan.l1 = w.detach().abs().sum(dims).float().contiguous()
        plan.l1_key = key
    return plan.l1


def _l1_in(conv: nn.Module, plan: _Plan) -> torch.Tensor:
    """sum |w| per INPUT channel: the data gradient's filters (|dgrad(g)[c]| <= l1_in[c] * max|g|)."""
    w = conv.weight
    key = (w.data_ptr(), w._version, w.device, _GENERATION[0])
    if getattr(plan, "l1_in", None) is None or plan.l1_in_key != key:
        dims = (1, 2, 3, 4) if isinstance(conv, nn.ConvTranspose3d) else (0, 2, 3, 4)
        plan.l1_in = w.detach().abs().sum(dims).float().contiguous()
        plan.l1_in_key = key
    return plan.l1_in


def _x3_train_layers(conv: nn.Module, plan: _Plan):
    """(forward layer, data-gradient layer) on the split kernels, weights scaled on the device; rebuilt when the weights change."""
    w = conv.weight
    key = (w.data_ptr(), w._version, w.device, _GENERATION[0])
    if getattr(plan, "x3t", None) is None or plan.x3t_key != key:
        k, s, p, d, transposed = _conv_geometry(conv)
        wd = w.detach()
        wm = ops.split_scale_of(wd.contiguous())
        fwd = ops.Conv3dLayerX3(wd, 3, s, 1, 1, transposed, w_mul_dev=wm)
        if transposed:
            dg = ops.Conv3dLayerX3(wd, 3, 2, 1, 1, False, w_mul_dev=wm)
        elif s == 1:
            dg = ops.Conv3dLayerX3(_flip3d(wd), 3, 1, 1, 1, False, w_mul_dev=wm)
        else:
            dg = ops.Conv3dLayerX3(wd, 3, 2, 1, 1, True, w_mul_dev=wm)
        plan.x3t, plan.x3t_key = (fwd, dg), key
    return plan.x3t


def _split_operand(t: torch.Tensor, amax: Optional[torch.Tensor]):
    """(pair, mul) of a float32 tensor for the split kernels: the twin its producer wrote, else a layout pass of its own (and a note
    to the producer, which writes the twin from the next step on)."""
    tw = ops.twin_of(t)
    if tw is not None:
        _ROUTES["x3_train_twin"] += 1
        return tw
    src = getattr(t, "snvc_twin_src", None)
    if src is not None:
        src.want_twin = True
    _ROUTES["x3_train_layout_pass"] += 1
    tc = t if (ops._dense_inner(t) and t.data_ptr() % 16 == 0) else t.contiguous()
    mul = ops.split_scale_bound(1, 1, t.device, amax_x=amax) if amax is not None else ops.split_scale_of(tc.contiguous())
    return ops.to_split(tc, mul_dev=mul), mul


class _X3TrainLayer:
    """The plain convolution of a layer on the split kernels behind Conv3dLayer's calling convention (what _norm_forward_impl calls
    with keep_raw): float32 NCDHW in (through its twin), float32 NCDHW out."""
    ksize = 3

    def __init__(self, layer: "ops.Conv3dLayerX3"):
        self.layer = layer

    def __call__(self, x, scale=None, bias=None, residual=None, flags=0, out=None, exact=False):
        if scale is not None or bias is not None or residual is not None or flags or out is not None:
            raise RuntimeError("_X3TrainLayer: the plain convolution only")
        pair, mul = _split_operand(x, ops.amax_of(x))
        return self.layer.forward_f32(pair, mul)

    def forward_stats(self, x, gamma, beta, eps):
        pair, mul = _split_operand(x, ops.amax_of(x))
        return self.layer.forward_stats(pair, mul, gamma, beta, eps)


class _GradBox:
    """Where a skip connection's gradient waits for the layer whose data gradient it is added to (see _SkipTap)."""
    __slots__ = ("grad", "consumed", "took")

    def __init__(self):
        self.grad, self.consumed, self.took = None, False, False


class _SkipTap(torch.autograd.Function):
    """``x`` used twice -- by a convolution and, further down, as a skip connection -- makes autograd add two gradients of x's
    size (0.38 ms for the 736 MB tensor of cfg4).  The skip branch goes through this identity instead: its gradient is parked in
    ``box`` and the convolution's own backward adds it in the epilogue of its data-gradient kernel (_ConvNormActFn, ``grad_box``).
    The skip's consumer lies downstream of that convolution, so its backward always runs first; if it ever did not, the
    tap raises instead of losing a gradient."""

    @staticmethod
    def forward(ctx, x, box):
        ctx.box = box
        out = x.view_as(x)
        am = ops.amax_of(x)               # the view is the same data: its consumer (a residual) may need max|x| for a twin's bound
        if am is not None:
            ops.tag_amax(out, am)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        if ctx.box.consumed:
            # the convolution's backward has run since the last time this tap fired.  If it TOOK a parked gradient then, that
            # was the previous pass over a retained graph (retain_graph=True, torch.autograd.grad twice): start a new pass.
            # If it took nothing, it ran before this tap in the same pass and this gradient would be lost.
            if not ctx.box.took:
                raise RuntimeError("_SkipTap: the convolution's backward ran before the skip connection's")
            ctx.box.consumed = ctx.box.took = False
        ctx.box.grad = g if ctx.box.grad is None else ctx.box.grad + g
        return None, None


def skip_box(x: torch.Tensor) -> Optional[_GradBox]:
    """A box for the gradient of x's skip-connection use, or None outside HIP training.  Order of use: make the box, run the
    convolution with ``grad_box=box``, THEN route the skip through ``_SkipTap.apply(x, box)``: autograd runs ready nodes
    newest first, so the tap (made after the convolution's node) hands its gradient over before the convolution's backward."""
    if torch.is_grad_enabled() and x.requires_grad and x.is_cuda and x.dtype == torch.float32:
        return _GradBox()
    return None


class _ConvNormActFn(torch.autograd.Function):
    """Differentiable fused layer: forward = conv (+norm) (+residual) (+activation) on the HIP
    kernels, backward = HIP epilogue-backward reductions, dgrad (forward kernels) and wgrad."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, conv, norm, flags, plan, grad_box=None):
        ctx.grad_box = grad_box
        ctx.x3 = _x3_train_route(conv, x)
        layer = _X3TrainLayer(_x3_train_layers(conv, plan)[0]) if ctx.x3 else _get_layer(conv, plan)
        # k5 / k7 layers under autograd: Winograd F(4,5) / F(4,7) forward and data gradient like at inference (r4; the local
        # trunk's conv1 3.7 instead of 7.5 ms) unless TRAIN_EXACT_K57 asks for the direct kernels' exact fp32 FMA chain -- the
        # F(4,7) forward is 1e-4 of the range off, inside the 1e-3 contract, but it moves a few ReLU masks, which gradient
        # comparisons against another implementation see as isolated differences
        ctx.x_amax = ops.amax_of(x)          # max|x| left by the pass that wrote x (None: the weight gradient finds it itself)
        twin = (_l1_out(conv, plan), ctx.x_amax) if (getattr(plan, "want_twin", False) and X3_TRAIN[0] and x.is_cuda) else None
        y, raw, scale, shift, mean, var, per_sample = _norm_forward(layer, norm, plan, x, residual, flags, None, True,
                                                                    exact=layer.ksize >= 5 and TRAIN_EXACT_K57[0], twin=twin)
        if y is not raw:
            y.snvc_twin_src = plan            # a consumer on the split kernels asks for the twin here (_split_operand)
        ctx.conv, ctx.norm, ctx.flags, ctx.plan, ctx.per_sample = conv, norm, flags, plan, per_sample
        ctx.has_res = residual is not None
        ctx.train_stats = mean is not None
        res_saved = residual if (residual is not None and (flags & EPI_ADD_PRE)) else None
        ctx.save_for_backward(x, raw, scale, shift, mean, var, res_saved)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, raw, scale, shift, mean, var, res = ctx.saved_tensors
        conv, norm, flags, plan = ctx.conv, ctx.norm, ctx.flags, ctx.plan
        needs = ctx.needs_input_grad
        x3 = ctx.x3 and needs[0]
        g_amax = ops.amax_word(raw.device) if ((needs[1] or x3) and raw.is_cuda) else None
        twin = (_l1_out(conv, plan), ctx.x_amax) if (x3 and ctx.x_amax is not None) else None
        draw, gres, dg, db = _epilogue_backward(raw, gy, res, scale, shift, mean, var, norm, flags, ctx.per_sample, ctx.train_stats,
                                                ctx.has_res and needs[4], needs[2], needs[3], amax_out=g_amax, twin=twin)
        if draw is gy:                       # no epilogue pass ran: nothing wrote the word
            g_amax = None
        # data and weight gradients
        k, st, p, d, transposed = _conv_geometry(conv)
        dl = _dgrad_layer(conv, plan)
        extra = None
        if ctx.grad_box is not None:        # the gradient of x's other use (a skip connection): added by the dgrad kernel's epilogue
            extra, ctx.grad_box.grad, ctx.grad_box.consumed = ctx.grad_box.grad, None, True
            ctx.grad_box.took = extra is not None
            if extra is not None and not needs[0]:
                raise RuntimeError("a skip connection's gradient was parked for a layer whose input needs no gradient")
            if extra is not None:
                extra = extra.contiguous()
        # Conv3d(k3,s2,p1) takes any extent (reference submodule.py:170-181); its data gradient runs on the
        # ConvTranspose3d(k3,s2,p1,op1) kernel, whose result is exactly twice the size of draw: for an odd input extent that is
        # one plane / row / column more than x has -- the gradient of the zero padding -- which is cropped off; the weight
        # gradient sees x with that padding position made explicit (a zero plane contributes nothing)
        odd = (st == 2 and not transposed) and any(int(e) % 2 for e in x.shape[2:])
        gx = None
        if x3:                               # r6: the data gradient on the split kernels, draw through its twin, the skip's gradient in the epilogue
            pair, mul = _split_operand(draw, g_amax)
            gx = _x3_train_layers(conv, plan)[1].forward_f32(pair, mul, residual_f32=extra)
            _ROUTES["x3_train_dgrad"] += 1
        elif needs[0]:
            if odd:
                gx = dl(draw, None, None, None, 0, None)[:, :, :x.size(2), :x.size(3), :x.size(4)].contiguous()
                if extra is not None:
                    gx = gx + extra
            else:
                gx = dl(draw, None, None, extra, EPI_ADD_POST if extra is not None else 0, None, exact=dl.ksize >= 5 and TRAIN_EXACT_K57[0])
        gw = None
        if needs[1]:
            if transposed:   # roles swapped, see snvc_conv3d_wgrad
                gw = ops.conv3d_wgrad(draw, x, 3, 2, 1, 1, amax_x=g_amax, amax_g=ctx.x_amax)
            elif odd:
                gw = ops.conv3d_wgrad(F.pad(x, (0, x.size(4) % 2, 0, x.size(3) % 2, 0, x.size(2) % 2)), draw, k, st, p, d)
            else:
                gw = ops.conv3d_wgrad(x, draw, k, st, p, d, amax_x=ctx.x_amax, amax_g=g_amax)
        return gx, gw, dg, db, gres, None, None, None, None, None


def _epilogue_backward(raw, gy, res, scale, shift, mean, var, norm, flags, per_sample, train_stats, want_res, want_gamma, want_beta,
                       amax_out=None, twin=None):
    """Backward of  y = act(norm(raw) [+ res]) [+ res]  given gy: returns (draw, gres, dgamma, dbeta) on the HIP
    reduction / apply kernels (BatchNorm / GroupNorm backward coefficients in fp64).  ``twin`` (r6) = (L1 norms of the layer's
    filters per output channel, the words of max|x|): the apply pass also writes draw's split twin, scaled from the bound
    max_c(|A_c| * max|gy| + |B_c| * L1_c * max|x| + |C_c|) with max|gy| taken by the reduction pass."""
    gy = gy.contiguous()
    am_gy = ops.amax_word(raw.device) if (twin is not None and norm is not None and ops.twin_ok(raw) and not (flags & EPI_SIGMOID)) else None

    def twin_mul(coef_g, coef_raw, coef_const):
        if am_gy is None:
            return None
        if coef_raw is None:               # frozen statistics: draw = A * g
            return ops.split_scale_bound(coef_g.numel(), raw.size(1), raw.device, a=coef_g, amax_p=am_gy, b=torch.zeros_like(coef_g))
        return ops.split_scale_bound(coef_g.numel(), raw.size(1), raw.device, a=coef_g, amax_p=am_gy, b=coef_raw, l1=twin[0], amax_x=twin[1],
                                     cc=coef_const)

    n, c = raw.shape[0], raw.shape[1]
    s = raw[0, 0].numel()
    dev = raw.device
    act_flags = flags & (EPI_RELU | EPI_SIGMOID | EPI_ADD_PRE)
    dgamma = dbeta = None
    if norm is None:
        coef_g = torch.ones(c, device=dev)
        coef_raw = coef_const = None
        per_sample = False
    else:
        sums = ops.act_backward_reduce(raw, gy, res, scale, shift, act_flags, per_sample, amax_gy=am_gy)   # [n, c, 2] fp64
        gam = norm.weight.detach().double() if norm.weight is not None else torch.ones(c, device=dev, dtype=torch.float64)
        if isinstance(norm, nn.GroupNorm):
            groups = norm.num_groups
            cpg = c // groups
            mu = mean.double().repeat_interleave(cpg, dim=1)                       # [n, c]
            rstd = torch.rsqrt(var.double() + norm.eps).repeat_interleave(cpg, dim=1)
            sg, sgr = sums[..., 0], sums[..., 1]
            sgx = rstd * (sgr - mu * sg)                                          # sum g * xhat per (n, c)
            p1 = (gam * sg).view(n, groups, cpg).sum(2).repeat_interleave(cpg, dim=1)
            p2 = (gam * sgx).view(n, groups, cpg).sum(2).repeat_interleave(cpg, dim=1)
            m = float(cpg * s)
            a_ = rstd * gam
            b_ = -rstd * rstd * p2 / m
            c_ = -rstd * p1 / m - b_ * mu
            dgamma, dbeta = sgx.sum(0), sg.sum(0)
        else:
            if train_stats and mean.dtype == torch.float32 and var.dtype == torch.float32:
                # batch statistics: every coefficient in one launch (fp64 inside), instead of ~15 per-channel tensor ops
                gw = norm.weight.detach().float().contiguous() if norm.weight is not None else None
                coef_g, coef_raw, coef_const, dgamma, dbeta = ops.bn_backward_coefs(
                    sums, mean[0].contiguous(), var[0].contiguous(), gw, float(n * s), float(norm.eps))
                if norm.weight is None:
                    dgamma = dbeta = None
                # the residual's gradient is g = gy * act'(v): with no activation (conv6: bn(conv) + x) it IS gy -- nothing to write
                # (r6: the pass wrote a 736 MB copy of gy at cfg4)
                want_g = want_res and bool(flags & EPI_ADD_PRE) and bool(flags & (EPI_RELU | EPI_SIGMOID))
                draw, g_out = ops.act_backward_apply(raw, gy, res, scale, shift, coef_g, coef_raw, coef_const, act_flags,
                                                     per_sample, want_g, amax=amax_out, twin_mul=twin_mul(coef_g, coef_raw, coef_const))
                gres = (g_out if want_g else gy) if want_res else None
                return draw, gres, (dgamma if want_gamma else None), (dbeta if want_beta else None)
            sg, sgr = sums[..., 0].sum(0), sums[..., 1].sum(0)                    # [c]
            if train_stats:
                mu, rstd = mean[0].double(), torch.rsqrt(var[0].double() + norm.eps)
                sgx = rstd * (sgr - mu * sg)
                m = float(n * s)
                a_ = gam * rstd
                b_ = -gam * rstd * rstd * sgx / m
                c_ = -gam * rstd * sg / m - b_ * mu
            else:   # frozen statistics: a plain per-channel affine
                mu, rstd = norm.running_mean.double(), torch.rsqrt(norm.running_var.double() + norm.eps)
                sgx = rstd * (sgr - mu * sg)
                a_, b_, c_ = gam * rstd, None, None
            dgamma, dbeta = sgx, sg
        coef_g = a_.float().contiguous()
        coef_raw = b_.float().contiguous() if b_ is not None else None
        coef_const = c_.float().contiguous() if c_ is not None else None
        if norm.weight is None:
            dgamma = dbeta = None
    want_g = want_res and bool(flags & EPI_ADD_PRE) and bool(flags & (EPI_RELU | EPI_SIGMOID))
    if norm is None and not act_flags:
        draw, g_out = gy, gy
    else:
        draw, g_out = ops.act_backward_apply(raw, gy, res, scale, shift, coef_g, coef_raw, coef_const, act_flags,
                                             per_sample, want_g, amax=amax_out, twin_mul=twin_mul(coef_g, coef_raw, coef_const))
    gres = None
    if want_res:
        gres = g_out if want_g else gy
    dg = dgamma.float() if (dgamma is not None and want_gamma) else None
    db = dbeta.float() if (dbeta is not None and want_beta) else None
    return draw, gres, dg, db


def _flip3d(w):
    """Weights of the dgrad of a stride-1 convolution: channel roles swapped, every axis flipped."""
    return w.transpose(0, 1).flip(2, 3, 4).contiguous()


def _first_conv_train_cache(conv, weight, c):
    """Packed layers of the factored first convolution's training functions, rebuilt when the weight changes: ``fl`` / ``bl``
    = the left half's forward / dgrad layers (the general and the sheared function add their right-half layers lazily)."""
    fac = conv.__dict__.setdefault("_snvc_factored_train", {})
    key = (weight.data_ptr(), weight._version, weight.device, _GENERATION[0])
    if fac.get("key") != key:
        wl = weight.detach()[:, :c].contiguous()
        fac.clear()
        fac.update(key=key, fl=ops.Conv3dLayer(wl, 3, 1, 1, 1, False), bl=ops.Conv3dLayer(_flip3d(wl), 3, 1, 1, 1, False))
    return fac


class _FactoredFirstConvFn(torch.autograd.Function):
    """Differentiable first layer of the global stack over a CONCAT cost volume that is never built:
        y = act(norm(conv3d(build_cost_volume(left, right, shift, 1), W)))          (k3, stride 1, 2C -> Cout)
    The left half of the volume repeats the left feature on every disparity plane (BuildCostVolume_cuda.cu:86), so
      forward : raw = conv3d(warped right half, W[:, C:]) + depth-class planes of conv3d(left stacked 3 deep, W[:, :C])
      backward: the right half goes through the ordinary 3D dgrad / wgrad at HALF the channels and the right-only
                cost-volume adjoint; the left half collapses to 2D work on the depth-class sums of the output gradient
                (snvc_depth_class_sums): a 3-plane wgrad / dgrad instead of a D-plane one.
    Same gradients as the materialised path up to summation order (tests: training step vs C oracle + torch autograd)."""

    @staticmethod
    def forward(ctx, left, right, shift, weight, gamma, beta, conv, norm, flags, plan, commuted=True):
        c = left.size(1)
        fac = _first_conv_train_cache(conv, weight, c)
        if "fr" not in fac:
            wr = weight.detach()[:, c:].contiguous()
            fac.update(fr=ops.Conv3dLayer(wr, 3, 1, 1, 1, False), br=ops.Conv3dLayer(_flip3d(wr), 3, 1, 1, 1, False))
        left3 = left.detach().unsqueeze(2).expand(-1, -1, 3, -1, -1).contiguous()
        planes = fac["fl"](left3)
        rd = right.detach()
        if commuted and shift.dtype == torch.float32 and rd.size(3) % 4 == 0 and rd.size(3) <= 2048:
            # forward: warp AFTER the convolution (csrc/sheared_conv.hip, any shift array): three depth-1 convolutions of the
            # right feature + three interpolations per voxel instead of the volume build and the 3D convolution over it
            if "commuted" not in fac:
                w = weight.detach()[:, c:]
                cout = w.size(0)
                wk = w.permute(2, 0, 1, 3, 4).contiguous()
                kq = torch.zeros_like(wk)
                kq[..., 1] = wk[..., 2]
                ke = torch.zeros((3, 3, cout, c, 3, 3), dtype=w.dtype, device=w.device)
                for kw in range(3):
                    ke[:, kw, :, :, :, 1] = wk[..., kw]
                mk = lambda t: ops.Conv3dLayer(t.reshape(-1, c, 3, 3).contiguous(), 3, 1, 1, 1, False, planar=True)   # noqa: E731
                fac["commuted"] = (mk(wk), mk(kq), mk(ke))
            lp, lq, le = fac["commuted"]
            r5 = rd.unsqueeze(2)
            raw = torch.empty((rd.size(0), weight.size(0), shift.size(1)) + tuple(rd.shape[2:]), dtype=torch.float32, device=rd.device)
            ops.warped_expand(lp(r5).squeeze(2), lq(r5).squeeze(2), le(rd[:, :, :, :4].contiguous().unsqueeze(2)).squeeze(2), planes,
                              shift.detach().float().contiguous(), None, None, raw, 0)
            _ROUTES["commuted_first_conv_train"] += 1
            # the backward warps back BEFORE the (transposed) convolution too (snvc_warped_expand_backward, r4) -- it assumes what
            # the reference's wrapper asserts, shift >= 0, which forward_pair has checked on the device by now
            ctx.commuted_bwd = COMMUTED_BACKWARD[0] and rd.size(3) <= 1024 and shift.size(1) >= 2
        else:
            vol_r = ops.cost_volume_forward_right(rd, shift)
            raw = fac["fr"](vol_r, None, None, None, 0, None, depth_planes=planes)
            del vol_r
            ctx.commuted_bwd = False
        y, scale, shf, mean, var, per_sample = _norm_from_raw(raw, norm, plan, None, flags)
        ctx.conv, ctx.norm, ctx.flags, ctx.per_sample, ctx.train_stats = conv, norm, flags, per_sample, mean is not None
        ctx.save_for_backward(left3, right.detach(), shift, raw, scale, shf, mean, var)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        left3, right, shift, raw, scale, shf, mean, var = ctx.saved_tensors
        conv, norm, flags = ctx.conv, ctx.norm, ctx.flags
        needs = ctx.needs_input_grad
        draw, _, dg, db = _epilogue_backward(raw, gy, None, scale, shf, mean, var, norm, flags, ctx.per_sample, ctx.train_stats,
                                             False, needs[4], needs[5])
        fac = conv.__dict__["_snvc_factored_train"]
        g_left = g_right = gw = None
        if ctx.commuted_bwd and (needs[1] or needs[3]):
            # one pass over draw: its nine warped-back, tap-shifted sums a[kd][kw] (and the left half's depth-class sums);
            # the right feature's and the right-half weights' gradients are depth-1 work on those 9 * Cout planes
            a, dplanes = ops.warped_expand_backward(draw, shift.detach().float())
            n, c, cout = right.size(0), right.size(1), draw.size(1)
            a5 = a.view(n, 9 * cout, 1, a.size(4), a.size(5))
            if needs[3]:
                g9 = ops.conv3d_wgrad(right.unsqueeze(2), a5, 3, 1, 1, 1)         # [9 Cout, C, 3, 3, 3]: (kh, centre column) used
                gw_r = g9[:, :, 1, :, 1].reshape(3, 3, cout, c, 3).permute(2, 3, 0, 4, 1)      # [Cout, C, kd, kh, kw]
                gw = torch.cat([ops.conv3d_wgrad(left3, dplanes, 3, 1, 1, 1), gw_r], dim=1)
            if needs[1]:
                if "ba" not in fac:     # dRight[c][y][j] = sum Wt[co][c][kd][kh][kw] a[kd][kw][co][y - kh + 1][j]: a depth-1 k3 layer
                    wr = conv.weight.detach()[:, c:]                              # [Cout, C, kd, kh, kw]
                    wd = torch.zeros((c, 9 * cout, 3, 3), dtype=wr.dtype, device=wr.device)
                    wd[:, :, :, 1] = wr.permute(1, 2, 4, 0, 3).flip(4).reshape(c, 9 * cout, 3)
                    fac["ba"] = ops.Conv3dLayer(wd, 3, 1, 1, 1, False, planar=True)
                g_right = fac["ba"](a5).squeeze(2)
            if needs[0]:
                g_left = fac["bl"](dplanes).sum(dim=2)
            _ROUTES["commuted_first_conv_backward"] += 1
            return g_left, g_right, None, gw, dg, db, None, None, None, None, None
        dplanes = ops.depth_class_sums(draw)                                      # [N,Cout,3,H,W]
        if needs[3]:
            vol_r = ops.cost_volume_forward_right(right, shift)                   # recomputed (0.17 ms) rather than kept (0.74 GB)
            gw_r = ops.conv3d_wgrad(vol_r, draw, 3, 1, 1, 1)
            del vol_r
            gw_l = ops.conv3d_wgrad(left3, dplanes, 3, 1, 1, 1)
            gw = torch.cat([gw_l, gw_r], dim=1)
        if needs[0]:
            g_left = fac["bl"](dplanes).sum(dim=2)                                # the three depth copies are one tensor
        if needs[1]:
            g_right = ops.cost_volume_backward_right(fac["br"](draw), shift)
        return g_left, g_right, None, gw, dg, db, None, None, None, None, None


COMMUTED_BACKWARD = [True]       # False: rounds 1-3's backward of the any-shift first layer (right half built, 3D dgrad + wgrad)


SHEAR_CLASS_KDS = ((0, 1), (-1, 0, 1), (-1, 0))     # kd taps the first plane / the interior planes / the last plane see


def sheared_kernels(wr: torch.Tensor, q: int) -> torch.Tensor:
    """The 3 x 7 kernels of the sheared first convolution (csrc/sheared_conv.hip), folded in fp64 from the right-half weights
    wr [Cout,C,3,3,3]: K[v][cls][co][c][kh][t + 3] = sum over (kd in class cls, kw) with q*kw - kd = t of wr[co,c,kd,kh,kw];
    v = 0: all kw (columns w <= W-2), v = 1: without the kw = +1 taps (the last column)."""
    w = wr.detach().double()
    k = torch.zeros((2, 3, w.shape[0], w.shape[1], 3, 7), dtype=torch.float64, device=w.device)
    for cls, kds in enumerate(SHEAR_CLASS_KDS):
        for kd in kds:
            for kw in (-1, 0, 1):
                t = q * kw - kd
                k[0, cls, :, :, :, t + 3] += w[:, :, kd + 1, :, kw + 1]
                if kw != 1:
                    k[1, cls, :, :, :, t + 3] += w[:, :, kd + 1, :, kw + 1]
    return k


def sheared_geometry(q: int, m0: int, d: int, w: int):
    """(off, wu, off_col, wu_col): where Rq sits on the padded grid G is computed on, and the window of Rq (3 columns of
    context each side) the last output column's G' reads at u = q*(W-1) - d - m0, d = 0 .. D-1."""
    off = 4
    wu = (off + q * (w - 1) + 1 + 3 + 3) // 4 * 4
    u_lo = q * (w - 1) - (d - 1) - m0 - 3
    return off, wu, 4 - u_lo, (d + 6 + 4 + 3) // 4 * 4


def _sheared_train_layers(fac, weight, c, q):
    """(forward layers G | G', their dgrad layers, the [2,3,7,3,3] scatter of the 3 x 7 weight gradient onto the 27 taps) of the
    sheared first convolution, cached with the layer's other training plans."""
    if ("shear", q) not in fac:
        k = sheared_kernels(weight.detach()[:, c:], q)                       # [2,3,Cout,C,3,7]
        cout = k.shape[2]
        planar = lambda t: ops.Conv3dLayer(t.float().contiguous(), 7, 1, 3, 1, False, planar=True, ksize_h=3)   # noqa: E731
        fwd = tuple(planar(k[v].reshape(3 * cout, c, 3, 7)) for v in range(2))
        # dgrad of a stride-1 layer: the same layer with the kernel flipped and the channel roles swapped
        bwd = tuple(planar(k[v].reshape(3 * cout, c, 3, 7).transpose(0, 1).flip(2, 3)) for v in range(2))
        # dW[kd][kw] gathers dK[v][cls][t = q*kw - kd]
        scatter = torch.zeros((2, 3, 7, 3, 3), dtype=torch.float32)
        for cls, kds in enumerate(SHEAR_CLASS_KDS):
            for kd in kds:
                for kw in (-1, 0, 1):
                    scatter[0, cls, q * kw - kd + 3, kd + 1, kw + 1] = 1.0
                    if kw != 1:
                        scatter[1, cls, q * kw - kd + 3, kd + 1, kw + 1] = 1.0
        fac[("shear", q)] = (fwd, bwd, scatter.to(weight.device))
    return fac[("shear", q)]


def _bn_track(norm, mean, var, cnt):
    """nn.BatchNorm3d's bookkeeping in train mode: momentum update of the running statistics with the unbiased variance."""
    if norm.training and norm.track_running_stats and norm.running_mean is not None:
        if ops.bn_track(norm, mean[0] if mean.dim() == 2 else mean, var[0] if var.dim() == 2 else var, cnt):     # r6: one launch instead of four
            return
        with torch.no_grad():
            norm.num_batches_tracked += 1
            m = norm.momentum if norm.momentum is not None else 1.0 / float(norm.num_batches_tracked)
            norm.running_mean.lerp_(mean[0], m)                                   # (1 - m) * running + m * batch, one launch each
            norm.running_var.lerp_(var[0] * (cnt / max(cnt - 1, 1)), m)


class _ShearedFirstConvFn(torch.autograd.Function):
    """``_FactoredFirstConvFn`` for uniformly spaced disparity planes, shift[n][d] = (m0 + d) / q with q in {1, 2}: the warped
    half of the volume is a shear of ONE image Rq and the 3D convolution over it a 2D 3 x 7 convolution G evaluated along the
    shear (csrc/sheared_conv.hip).  Neither the warped volume nor any 3D product over it exists in either direction:
      forward : raw = expand(G, G') + depth-class planes of the left half
      backward: dG / dG' = the sums of draw along the shear lines (snvc_sheared_reduce); the weight gradient is the 3 x 7 one
                of the 2D layer (snvc_sheared_wgrad) scattered back onto the 27 taps; the input gradient is the 2D layer's
                dgrad (flipped kernel) followed by the adjoint of the half-pixel interpolation.  Left half as in
                ``_FactoredFirstConvFn``.
    Same gradients as the general path up to summation order (tests/test_gpu_parity.py)."""

    @staticmethod
    def forward(ctx, left, right, weight, gamma, beta, conv, norm, flags, plan, q, m0, depth):
        c = left.size(1)
        fac = _first_conv_train_cache(conv, weight, c)
        fwd = _sheared_train_layers(fac, weight, c, q)[0]
        n, h, w = left.size(0), left.size(2), left.size(3)
        off, wu, off_col, wu_col = sheared_geometry(q, m0, depth, w)
        left3 = left.detach().unsqueeze(2).expand(-1, -1, 3, -1, -1).contiguous()
        planes = fac["fl"](left3)
        rd = right.detach()
        g = fwd[0](ops.sheared_upsample(rd, q, wu, off).unsqueeze(2)).squeeze(2)
        gcol = fwd[1](ops.sheared_upsample(rd, q, wu_col, off_col).unsqueeze(2)).squeeze(2)
        raw = torch.empty((n, weight.size(0), depth, h, w), dtype=torch.float32, device=left.device)
        ops.sheared_expand(g, gcol, planes, None, None, raw, q, m0, off, off_col, 0)
        del g, gcol
        y, scale, shf, mean, var, per_sample = _norm_from_raw(raw, norm, plan, None, flags)
        ctx.conv, ctx.norm, ctx.flags, ctx.per_sample, ctx.train_stats = conv, norm, flags, per_sample, mean is not None
        ctx.q, ctx.m0 = q, m0
        ctx.save_for_backward(left3, rd, raw, scale, shf, mean, var)
        _ROUTES["sheared_first_conv_train"] += 1
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        left3, right, raw, scale, shf, mean, var = ctx.saved_tensors
        conv, norm, flags, q, m0 = ctx.conv, ctx.norm, ctx.flags, ctx.q, ctx.m0
        needs = ctx.needs_input_grad
        draw, _, dg, db = _epilogue_backward(raw, gy, None, scale, shf, mean, var, norm, flags, ctx.per_sample, ctx.train_stats,
                                             False, needs[3], needs[4])
        fac = conv.__dict__["_snvc_factored_train"]
        depth, w = draw.size(2), draw.size(4)
        off, wu, off_col, wu_col = sheared_geometry(q, m0, depth, w)
        dplanes = ops.depth_class_sums(draw)                                      # [N,Cout,3,H,W]
        d_g = d_gcol = None
        if needs[1] or needs[2]:
            d_g, d_gcol = ops.sheared_reduce(draw, q, m0, wu, off, wu_col, off_col)
        g_left, g_right, gw = _sheared_backward_tail(fac, q, m0, depth, w, left3, right, d_g, d_gcol, dplanes, needs[0], needs[1], needs[2])
        return g_left, g_right, gw, dg, db, None, None, None, None, None, None, None


def _sheared_backward_tail(fac, q, m0, depth, w, left3, right, d_g, d_gcol, dplanes, need_left, need_right, need_weight):
    """From the gradients of G / G' (d_g [N,3C,H,WG], d_gcol [N,3C,H,WG2]) and of the left half's depth-class planes to the
    gradients of the two features and of the layer's 3x3x3 weight."""
    _, bwd, scatter = fac[("shear", q)]
    off, wu, off_col, wu_col = sheared_geometry(q, m0, depth, w)
    g_left = g_right = gw = None
    if need_weight:
        cout = dplanes.size(1)
        dk = torch.stack([ops.sheared_wgrad(ops.sheared_upsample(right, q, wu, off), d_g),
                          ops.sheared_wgrad(ops.sheared_upsample(right, q, wu_col, off_col), d_gcol)])   # [2,3*Cout,C,3,7]
        gw_r = torch.einsum("vsockt,vstdw->ocdkw", dk.view(2, 3, cout, -1, 3, 7), scatter)
        gw_l = ops.conv3d_wgrad(left3, dplanes, 3, 1, 1, 1)
        gw = torch.cat([gw_l, gw_r], dim=1)
    if need_left:
        g_left = fac["bl"](dplanes).sum(dim=2)                                    # the three depth copies are one tensor
    if need_right:
        g_right = ops.sheared_upsample_backward(bwd[0](d_g.unsqueeze(2)).squeeze(2), q, w, off)
        g_right = g_right + ops.sheared_upsample_backward(bwd[1](d_gcol.unsqueeze(2)).squeeze(2), q, w, off_col)
    return g_left, g_right, gw


def _sheared_term_counts(q, m0, depth, w, device):
    """How many elements of the layer's result lie on each shear line / in each last-column slot / depth class (per class):
    the multiplicity of the BatchNorm backward's constant term in the sums of ``_ShearedFirstConvBNFn.backward``."""
    key = (q, m0, depth, w, str(device))
    hit = _SHEAR_COUNTS.get(key)
    if hit is None:
        import numpy as np
        off, wu, off_col, wu_col = sheared_geometry(q, m0, depth, w)
        line, col = np.zeros((3, wu), np.float32), np.zeros((3, wu_col), np.float32)
        d = np.arange(depth)
        cls = np.where(d == 0, 0, np.where(d == depth - 1, 2, 1))
        i = q * np.arange(w - 1)[None, :] - d[:, None] - m0 + off                    # [D, W-1]
        ok = (i >= 0) & (i < wu)
        np.add.at(line, (np.broadcast_to(cls[:, None], i.shape)[ok], i[ok]), 1.0)
        i2 = q * (w - 1) - d - m0 + off_col
        ok2 = (i2 >= 0) & (i2 < wu_col)
        np.add.at(col, (cls[ok2], i2[ok2]), 1.0)
        per_class = np.array([1.0, max(depth - 2, 0), 1.0], np.float32)
        hit = _SHEAR_COUNTS[key] = tuple(torch.from_numpy(a).to(device) for a in (line, col, per_class))
    return hit


_SHEAR_COUNTS = {}


class _ShearedFirstConvBNFn(torch.autograd.Function):
    """``_ShearedFirstConvFn`` for the layer as the reference builds it -- train-mode BatchNorm3d + ReLU -- WITHOUT the layer's
    raw result ever being stored or its gradient formed (two 736 MB tensors on cfg4):
      forward : batch statistics while walking the values the expansion is about to write (snvc_sheared_expand_stats), then
                y = relu(scale * expand(G, G') + planes ... ) written once
      backward: ONE pass over gy (snvc_sheared_backward_reduce) recomputes raw from G, masks gy and leaves per-channel fp64 sums
                plus the sums of the masked gradient and of raw along the shear lines and over the depth classes; the BatchNorm
                backward draw = A*g + B*raw + Cc is linear, so dG, dG' and the planes' gradients are combinations of those sums.
    Gradients equal ``_ShearedFirstConvFn``'s up to summation order (tests/test_gpu_parity.py)."""

    @staticmethod
    def forward(ctx, left, right, weight, gamma, beta, conv, norm, plan, q, m0, depth):
        c = left.size(1)
        fac = _first_conv_train_cache(conv, weight, c)
        fwd = _sheared_train_layers(fac, weight, c, q)[0]
        n, h, w = left.size(0), left.size(2), left.size(3)
        off, wu, off_col, wu_col = sheared_geometry(q, m0, depth, w)
        left3 = left.detach().unsqueeze(2).expand(-1, -1, 3, -1, -1).contiguous()
        planes = fac["fl"](left3)
        rd = right.detach()
        g = fwd[0](ops.sheared_upsample(rd, q, wu, off).unsqueeze(2)).squeeze(2)
        gcol = fwd[1](ops.sheared_upsample(rd, q, wu_col, off_col).unsqueeze(2)).squeeze(2)
        shape = (n, weight.size(0), depth, h, w)
        gam = gamma.detach() if gamma is not None else None
        bet = beta.detach() if beta is not None else None
        scale, shift, mean, var = ops.sheared_expand_stats(g, gcol, planes, gam, bet, shape, q, m0, off, off_col, norm.eps)
        _bn_track(norm, mean, var, float(n * depth * h * w))
        y = torch.empty(shape, dtype=torch.float32, device=left.device)
        # max|y| for the next layer's split operands (r6), left by the expansion itself (first form: an upper bound from max|G|, max|G'|,
        # max|planes| -- sixteen small torch launches per step)
        am = ops.amax_word(y.device)
        ops.sheared_expand(g, gcol, planes, scale, shift, y, q, m0, off, off_col, EPI_RELU, amax=am)
        ops.tag_amax(y, am)
        y.snvc_twin_src = plan
        if getattr(plan, "want_twin", False) and X3_TRAIN[0] and ops.twin_ok(y):
            # the next layer runs on the split kernels (r6): the expansion once more, written as the split pair (0.15 ms at cfg4 against a
            # layout pass over the 736 MB result), scaled by the power of two of the same bound
            mul = ops.split_scale_bound(1, 1, y.device, amax_x=am)
            pair = ops.twin_empty(y)
            ops.sheared_expand_split(g, gcol, planes, (scale * mul).contiguous(), (shift * mul).contiguous(), pair, q, m0, off, off_col, EPI_RELU)
            ops.tag_twin(y, pair, mul)
        ctx.conv, ctx.norm, ctx.q, ctx.m0 = conv, norm, q, m0
        # gamma is saved as the autograd input it is: an in-place update between forward and backward (an interleaved
        # optimizer step, an EMA swap) then trips autograd's version check instead of pairing a new gamma with old scale / shift
        ctx.save_for_backward(left3, rd, g, gcol, planes, scale, shift, mean, var, gamma)
        _ROUTES["sheared_first_conv_train"] += 1
        _ROUTES["sheared_first_conv_train_fused_bn"] += 1
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        left3, right, g, gcol, planes, scale, shift, mean, var, gamma = ctx.saved_tensors
        conv, norm, q, m0 = ctx.conv, ctx.norm, ctx.q, ctx.m0
        needs = ctx.needs_input_grad
        gy = gy.contiguous()
        n, c, depth, h, w = gy.shape
        off, wu, off_col, wu_col = sheared_geometry(q, m0, depth, w)
        line, colsum, lastc, sums = ops.sheared_backward_reduce(g, gcol, planes, scale, shift, gy, q, m0, off, off_col)
        gw_ = gamma.detach().float().contiguous() if gamma is not None else None
        a_, b_, c_, dgamma, dbeta = ops.bn_backward_coefs(sums, mean[0].contiguous(), var[0].contiguous(), gw_,
                                                          float(n * depth * h * w), float(norm.eps))
        if gamma is None:
            dgamma = dbeta = None
        cnt_line, cnt_col, cnt_cls = _sheared_term_counts(q, m0, depth, w, gy.device)
        ch = lambda t: t.view(1, 1, c, 1, 1)                                       # noqa: E731  [N,3,C,H,*] layouts
        combine = lambda t, cnt: (ch(a_) * t[0].view(n, 3, c, h, -1) + ch(b_) * t[1].view(n, 3, c, h, -1)          # noqa: E731
                                  + ch(c_) * cnt.view(1, 3, 1, 1, -1)).view(n, 3 * c, h, -1)
        d_g, d_gcol = combine(line, cnt_line), combine(lastc, cnt_col)
        pc = lambda t: t.view(1, c, 1, 1, 1)                                       # noqa: E731  [N,C,3,H,W] layout
        dplanes = pc(a_) * colsum[0] + pc(b_) * colsum[1] + pc(c_) * cnt_cls.view(1, 1, 3, 1, 1)
        fac = conv.__dict__["_snvc_factored_train"]
        g_left, g_right, gw = _sheared_backward_tail(fac, q, m0, depth, w, left3, right, d_g, d_gcol, dplanes.contiguous(),
                                                     needs[0], needs[1], needs[2])
        return g_left, g_right, gw, (dgamma if needs[3] else None), (dbeta if needs[4] else None), None, None, None, None, None, None


def _is_frozen_norm(norm) -> bool:
    return norm is None or (isinstance(norm, nn.BatchNorm3d) and not (norm.training or norm.running_mean is None))


def _is_channel_head(head, cin) -> bool:
    """A bias-free Conv3d(cin, 1, kernel_size=1): the projection the head fusions are built for."""
    return (isinstance(head, nn.Conv3d) and head.bias is None and tuple(head.weight.shape) == (1, cin, 1, 1, 1)
            and head.weight.dtype == torch.float32)


def _folded_head_layer(conv: nn.Module, norm: Optional[nn.Module], head: nn.Module, plan: _Plan):
    """``head(norm(conv(x)))`` with NO activation in between is one layer to ONE channel:
    W'[ci][tap] = sum_co h[co] * scale[co] * W[co, ci][tap],  b' = sum_co h[co] * shift[co]   (folded in fp64).
    Returns (layer, one, bias) with ``one`` / ``bias`` the [1] scale / shift tensors of that layer's epilogue."""
    w, hw = conv.weight, head.weight
    bn_key = None
    if norm is not None:
        bn_key = (norm.weight._version if norm.weight is not None else -1, norm.bias._version if norm.bias is not None else -1,
                  norm.running_mean._version, norm.running_var._version, norm.running_mean.data_ptr())
    key = (w.data_ptr(), w._version, hw.data_ptr(), hw._version, bn_key, w.device, _GENERATION[0])
    cached = getattr(plan, "folded_head", None)
    if cached is None or cached[0] != key:
        k, s, p, d, transposed = _conv_geometry(conv)
        wf, fb = folded_head_weights(conv, norm, head)
        layer = ops.Conv3dLayer(wf.float().contiguous(), k, s, p, d, transposed)
        one = torch.ones(1, dtype=torch.float32, device=w.device)
        cached = plan.folded_head = (key, layer, one, fb.float().reshape(1))
    return cached[1], cached[2], cached[3]


def folded_head_weights(conv: nn.Module, norm: Optional[nn.Module], head: nn.Module):
    """(W', b') of ``head(norm(conv(x)))`` as one layer to one channel, folded in fp64: W' has conv's weight layout with one output
    channel ([Cin, 1, k, k, k] for a transposed layer, [1, Cin, k, k, k] otherwise), b' is a scalar tensor."""
    w, hw = conv.weight, head.weight
    h = hw.detach().double().reshape(-1)
    if norm is not None:
        var, mean = norm.running_var.detach().double(), norm.running_mean.detach().double()
        g = norm.weight.detach().double() if norm.weight is not None else torch.ones_like(var)
        b = norm.bias.detach().double() if norm.bias is not None else torch.zeros_like(var)
        sc = g / torch.sqrt(var + norm.eps)
        sh = b - mean * sc
    else:
        sc, sh = torch.ones_like(h), torch.zeros_like(h)
    hs = h * sc
    wd = w.detach().double()
    if isinstance(conv, nn.ConvTranspose3d):      # [Cin, Cout, k, k, k] -> [Cin, 1, k, k, k]
        wf = torch.einsum("iodhw,o->idhw", wd, hs).unsqueeze(1)
    else:                                         # [Cout, Cin, k, k, k] -> [1, Cin, k, k, k]
        wf = torch.einsum("oidhw,o->idhw", wd, hs).unsqueeze(0)
    return wf, (h * sh).sum()


def fused_conv3d(conv: nn.Module, norm: Optional[nn.Module], x: torch.Tensor, *, relu=False, sigmoid=False,
                 residual: Optional[torch.Tensor] = None, residual_after_act=False, out=None,
                 plan: Optional[_Plan] = None, head: Optional[nn.Module] = None,
                 head_residual: Optional[torch.Tensor] = None, side_head: Optional[nn.Module] = None,
                 grad_box: Optional[_GradBox] = None):
    """act(norm(conv(x)) [+ residual]) [+ residual] on the HIP kernels.

    ``residual_after_act=False``: relu(norm(conv(x)) + residual)   (hourglass skips, submodule.py:154,162)
    ``residual_after_act=True`` : relu(norm(conv(x))) + residual   (vernier.py:418-419)

    Under autograd (training, BASELINE.json configs[3]) the layer is a differentiable
    ``torch.autograd.Function`` whose backward also runs on the HIP kernels.

    ``head``: a bias-free ``Conv3d(C, 1, kernel_size=1)`` applied to the result (extension).  When the layer is a
    transposed convolution with 32 output channels and frozen statistics, the projection happens inside the
    layer's epilogue and the C-channel tensor is never written (``snvc_conv3d_forward_head``); otherwise the two
    layers run one after the other.  Returns ``head(result)``.
    """
    if plan is None:  # one plan per device (replicas made by nn.DataParallel share __dict__ entries)
        plan = conv.__dict__.setdefault("_snvc_plans", {}).setdefault(x.device, _Plan())
    if conv.weight.device != x.device:
        raise RuntimeError(f"conv3d weight is on {conv.weight.device} but the input is on {x.device}")
    flags = (EPI_RELU if relu else 0) | (EPI_SIGMOID if sigmoid else 0)
    if residual is not None:
        flags |= EPI_ADD_POST if residual_after_act else EPI_ADD_PRE
    tracked = [x, conv.weight] + ([norm.weight, norm.bias] if norm is not None and getattr(norm, "weight", None) is not None else [])
    if residual is not None:
        tracked.append(residual)
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tracked):
        gamma = norm.weight if norm is not None else None
        beta = norm.bias if norm is not None else None
        if out is not None:
            raise NotImplementedError("`out=` (in-place concat slices) is an inference-only fusion")
        y = _ConvNormActFn.apply(x, conv.weight, gamma, beta, residual, conv, norm, flags, plan, grad_box)
        if side_head is not None:
            return y, side_head(y)
        return head(y) if head is not None else y
    if grad_box is not None:
        raise RuntimeError("grad_box was given to a layer that runs outside autograd: the skip connection's gradient would be lost")
    layer = _get_layer(conv, plan)
    if side_head is not None:
        if head is not None:
            raise NotImplementedError("`head` and `side_head` are exclusive")
        if not torch.is_grad_enabled() and _is_frozen_norm(norm) and _is_channel_head(side_head, layer.cout):
            scale, bias = _folded_bn(norm, plan) if norm is not None else (None, None)
            y, hy = layer(x, scale, bias, residual, flags, out, side_head=side_head.weight)
            _ROUTES["side_head" if hy is not None else "side_head_separate"] += 1
        else:
            y, hy = _norm_forward(layer, norm, plan, x, residual, flags, out, False)[0], None
        return y, (hy if hy is not None else side_head(y))
    if head is not None:
        fusable = (out is None and not torch.is_grad_enabled() and _is_channel_head(head, layer.cout) and _is_frozen_norm(norm))
        if fusable and not (flags & (EPI_RELU | EPI_SIGMOID)):
            fl, one, fbias = _folded_head_layer(conv, norm, head, plan)
            hres = None
            if residual is not None or head_residual is not None:      # head_residual alone: the caller only has head(residual)
                hres = head_residual if head_residual is not None else head(residual)
            _ROUTES["folded_head"] += 1
            return fl(x, one, fbias, hres, EPI_ADD_PRE if hres is not None else 0)
        if fusable:
            scale, bias = _folded_bn(norm, plan) if norm is not None else (None, None)
            y = ops.conv3d_forward_head(layer, x, scale, bias, residual, flags, head.weight)
            if y is not None:
                return y
        return head(_norm_forward(layer, norm, plan, x, residual, flags, out, False)[0])
    return _norm_forward(layer, norm, plan, x, residual, flags, out, False)[0]


def fused_conv3d_avgpool_d4(conv: nn.Module, norm: Optional[nn.Module], x: torch.Tensor, *, relu=False) -> torch.Tensor:
    """``avg_pool3d(act(norm(conv(x))), (4,1,1), (4,1,1))`` (reference vernier.py:289,435-436: conv4 and the pool in front of the
    BEV reshape).  With frozen statistics and nothing to differentiate the pool is part of the conv launch's epilogue
    (``SNVC_EPI_AVGPOOL_D4``: the full-resolution tensor is never written); otherwise the layer and the pool run one
    after the other."""
    if not torch.is_grad_enabled() and x.is_cuda and _is_frozen_norm(norm):
        plan = conv.__dict__.setdefault("_snvc_plans", {}).setdefault(x.device, _Plan())
        layer = _get_layer(conv, plan)
        scale, bias = _folded_bn(norm, plan) if norm is not None else (None, None)
        y = ops.conv3d_forward_avgpool_d4(layer, x, scale, bias, EPI_RELU if relu else 0)
        if y is not None:
            _ROUTES["conv_avgpool_fused"] += 1
            return y
    y = fused_conv3d(conv, norm, x, relu=relu)
    if torch.is_grad_enabled() and y.requires_grad:
        return ops.AvgPoolDepth4Fn.apply(y) if (y.is_cuda and y.dtype == torch.float32) else F.avg_pool3d(y, (4, 1, 1), (4, 1, 1))
    _ROUTES["conv_avgpool_separate"] += 1
    return ops.avgpool_depth4(y)


def fused_conv3d_f16(conv: nn.Module, norm: Optional[nn.Module], x: torch.Tensor, *, relu=False, sigmoid=False,
                     residual: Optional[torch.Tensor] = None, residual_after_act=False, out=None) -> torch.Tensor:
    """``fused_conv3d`` in the fp16-storage mode (inference only; BASELINE.json configs[4]): ``x`` / ``residual`` /
    ``out`` are C8 half tensors (``ops.to_c8``), the parameters stay fp32 ``nn.Parameter``s and are rounded to half
    when packed, BatchNorm must be in eval mode (folded into the fp32 epilogue).  A one-output-channel layer
    (the occupancy head) returns its result as a float32 ``[N,1,D,H,W]`` tensor."""
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for m in (conv, norm) if m is not None
                                                           for p in m.parameters())):
        raise NotImplementedError("the fp16-storage mode is inference only: call it under torch.no_grad() "
                                  "(or freeze the parameters); nothing here has a backward pass")
    if norm is not None and not (isinstance(norm, nn.BatchNorm3d) and not norm.training and norm.running_mean is not None):
        raise NotImplementedError("the fp16-storage mode needs eval-mode BatchNorm3d (GroupNorm / batch statistics: fp32 path)")
    if conv.weight.device != x.device:
        raise RuntimeError(f"conv3d weight is on {conv.weight.device} but the input is on {x.device}")
    plan = conv.__dict__.setdefault("_snvc_plans_f16", {}).setdefault(x.device, _Plan())
    w = conv.weight
    key = (w.data_ptr(), w._version, w.device, _GENERATION[0])
    if plan.layer is None or plan.key != key:
        k, s, p, d, transposed = _conv_geometry(conv)
        plan.layer = ops.Conv3dLayerF16(w.detach(), k, s, p, d, transposed)
        plan.key = key
    scale, bias = _folded_bn(norm, plan) if norm is not None else (None, None)
    flags = (EPI_RELU if relu else 0) | (EPI_SIGMOID if sigmoid else 0)
    if residual is not None:
        flags |= EPI_ADD_POST if residual_after_act else EPI_ADD_PRE
    return plan.layer(x, scale, bias, residual, flags, out)


# ------------------------------------------------------------------------------------------
# split mode ("f16x3", r4): inference with frozen statistics on the snvc_f16x3_* kernels -- the fp32 layers at fp32 accuracy on
# the half-precision matrix pipe (csrc/conv3d_f16.hip, F16Cfg::PL).  A tensor travels as SplitT: the (hi, lo) pair,
# the power of two its values are stored times (an int exponent, or a one-element device tensor when the range is only known
# from the data), and the bound |value| is promised to stay below (None for data-scaled tensors: they cannot overflow).
# ------------------------------------------------------------------------------------------
X3_SIGMAS = 64.0     # a BatchNorm output is promised to stay below |beta| + X3_SIGMAS * |gamma|


class SplitT:
    __slots__ = ("t", "exp", "mul_dev", "bound")

    def __init__(self, t, exp=0, bound=None, mul_dev=None):
        self.t, self.exp, self.bound, self.mul_dev = t, exp, bound, mul_dev

    def slice_groups(self, lo: int, hi: int):
        """Channel groups [lo, hi) of the pair (a view: split tensors are [N, 2, C/8, D, H, W, 8])."""
        return SplitT(self.t[:, :, lo:hi], self.exp, self.bound, self.mul_dev)


class SplitOverflow(RuntimeError):
    """Raised INSIDE a split-mode call whose overflow flag came back set: the call's result (an activation clamped to half's
    range) is dropped and the model's public entry point redoes the call on the fp32-MFMA kernels.  Never reaches the caller
    unless split mode was demanded (arithmetic / precision = "x3")."""


class OverflowGuard:
    """The overflow flag of a model's split-mode calls: an int32 on the device that every clamping epilogue ORs into, its pinned
    host copy and the event behind the copy.

    ``post()`` is queued right after the LAST layer that can clamp (the layers behind it write float32); ``wait()`` is called
    once the rest of the call has been queued: the host then waits for the flag while the GPU still has those last layers to
    run, so the check costs no GPU idle time and the result never leaves the call unchecked (r4 looked at the flag one call
    late).  ``check="deferred"`` models post without waiting; ``pending()`` is the synchronous look a caller can take then."""

    def __init__(self, device):
        self.flag = torch.zeros(1, dtype=torch.int32, device=device)
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.event = None

    def post(self):
        self.host.copy_(self.flag, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def wait(self) -> bool:
        """True if a value was clamped since the last look (the device flag is cleared again then)."""
        ev, self.event = self.event, None
        if ev is None:
            return False
        ops.spin_wait(ev)
        if int(self.host.item()) == 0:
            return False
        self.flag.zero_()
        return True

    pending = wait


def overflow_guard(module, device) -> OverflowGuard:
    """The module's guard for ``device`` (kept across rebuilds of the packed split-mode state: a pending flag is never dropped)."""
    guards = module.__dict__.setdefault("_snvc_x3_guard", {})
    g = guards.get(device)
    if g is None:
        g = guards[device] = OverflowGuard(device)
    return g


def x3_exponent(bound: float) -> int:
    """e with bound * 2^e <= 2^15 (half overflows at 65504; a value beyond the bound is clamped and FLAGGED)."""
    if not (bound > 0.0) or not math.isfinite(bound):
        return 0
    return max(-14, min(14, 15 - math.frexp(bound)[1]))


def x3_norm_bound(norm, plan) -> float:
    """|beta| + X3_SIGMAS * |gamma| of a frozen BatchNorm3d, maximised over channels (cached with the folded affine); of a
    GroupNorm likewise (its result is gamma * xhat + beta with xhat normalised per sample and group)."""
    if isinstance(norm, nn.GroupNorm):
        w, b = norm.weight, norm.bias
        gkey = (None if w is None else (w.data_ptr(), w._version), None if b is None else (b.data_ptr(), b._version), _GENERATION[0])
        hit = getattr(plan, "x3_gn_bound", None)
        if hit is None or hit[0] != gkey:
            g = w.detach().abs() if w is not None else torch.ones(1)
            bb = b.detach().abs().to(g.device) if b is not None else torch.zeros(1, device=g.device)
            hit = plan.x3_gn_bound = (gkey, float((bb + X3_SIGMAS * g).max().item()), w, b)      # w, b held: no address reuse
        return hit[1]
    key = getattr(plan, "bn_key", None)
    hit = getattr(plan, "x3_bound", None)
    if hit is None or hit[0] != key or key is None:
        _folded_bn(norm, plan)
        g = norm.weight.detach().abs() if norm.weight is not None else torch.ones(1, device=norm.running_mean.device)
        b = norm.bias.detach().abs() if norm.bias is not None else torch.zeros(1, device=norm.running_mean.device)
        hit = plan.x3_bound = (plan.bn_key, float((b + X3_SIGMAS * g).max().item()))
    return hit[1]


def x3_ok(*modules, group_norm: bool = True) -> bool:
    """Every norm a frozen BatchNorm3d (eval mode, running statistics) or -- r5, ``group_norm`` -- a GroupNorm (its statistics are
    taken from the layer's fp32 result, see fused_conv3d_x3); nothing to differentiate."""
    if torch.is_grad_enabled():
        return False
    for m in modules:
        for n in m.modules():
            if isinstance(n, nn.GroupNorm) and not (group_norm and X3_GROUP_NORM[0]):
                return False
            if isinstance(n, nn.modules.batchnorm._BatchNorm) and (n.training or n.running_mean is None):
                return False
    return True


X3_GROUP_NORM = [True]      # False: GroupNorm models stay on the fp32-MFMA kernels (r4's behaviour; kept for measuring)


def fused_conv3d_x3(conv: nn.Module, norm: Optional[nn.Module], x: SplitT, *, relu=False, sigmoid=False, residual: Optional[SplitT] = None,
                    residual_after_act=False, out=None, out_exp: Optional[int] = None, to_f32=False, flag=None):
    """``fused_conv3d`` in split mode: returns a SplitT (or, ``to_f32`` / a one-channel layer, a float32 tensor).  The result's
    exponent is ``out_exp`` if given (a slice of a larger pair must share the pair's), else chosen from the bound
    |beta| + X3_SIGMAS |gamma| (+ the residual's bound)."""
    plan = conv.__dict__.setdefault("_snvc_plans_x3", {}).setdefault(x.t.device, _Plan())
    w = conv.weight
    key = (w.data_ptr(), w._version, w.device, _GENERATION[0])
    if plan.layer is None or plan.key != key:
        k, s_, p_, d_, transposed = _conv_geometry(conv)
        plan.layer = ops.Conv3dLayerX3(w.detach(), k, s_, p_, d_, transposed)
        plan.key = key
    scale = bias = None
    bound = None
    if isinstance(norm, nn.GroupNorm):
        return _fused_conv3d_x3_gn(plan, norm, x, relu, sigmoid, residual, residual_after_act, out, out_exp, to_f32, flag)
    if norm is not None:
        scale, bias = _folded_bn(norm, plan)
        bound = x3_norm_bound(norm, plan)
    flags = (EPI_RELU if relu else 0) | (EPI_SIGMOID if sigmoid else 0)
    f32 = to_f32 or plan.layer.cout == 1
    res_t = None
    if residual is not None:
        flags |= EPI_ADD_POST if residual_after_act else EPI_ADD_PRE
        if residual.bound is None or bound is None:
            raise RuntimeError("split mode: a residual add needs bounds on both summands")
        bound = bound + residual.bound
        res_t = residual.t
    if not f32:
        if bound is None:
            raise RuntimeError("split mode: a layer without a frozen norm has no a-priori bound; ask for a float32 result")
        e = x3_exponent(bound) if out_exp is None else out_exp
    else:
        e = residual.exp if residual is not None else 0
    y = plan.layer(x.t, 0 if x.mul_dev is not None else x.exp, scale, bias, residual=res_t, flags=flags, out=out, out_exp=e,
                   to_f32=f32, overflow=flag, x_mul_dev=x.mul_dev, res_exp=residual.exp if residual is not None else None)
    return y if f32 else SplitT(y, e, bound)


def _fused_conv3d_x3_gn(plan, norm, x: SplitT, relu, sigmoid, residual, residual_after_act, out, out_exp, to_f32, flag):
    """A ``convbn_3d(..., gn=True)`` layer (reference submodule.py:41-49) in split mode (r5): GroupNorm needs the statistics of the
    convolution's own result, so the layer is  split-mode convolution -> fp32 NCDHW raw result -> snvc_norm_stats (per sample and
    group) -> one pass that applies scale / shift (+ residual, activation) and writes the split pair
    (snvc_f16x3_affine_from_ncdhw) -- or, ``to_f32``, the fp32 tensor (snvc_affine_act).  The convolution is the same three-MFMA
    arithmetic as every split layer; the statistics and the affine are the fp32 path's own kernels."""
    if sigmoid:
        raise NotImplementedError("split mode: GroupNorm + Sigmoid is not on the path")
    raw = plan.layer(x.t, 0 if x.mul_dev is not None else x.exp, None, None, flags=0, out_exp=0, to_f32=True, x_mul_dev=x.mul_dev)
    scale, shift, _, _ = ops.norm_stats(raw, norm.weight, norm.bias, norm.num_groups, True, norm.eps)
    flags = EPI_RELU if relu else 0
    _ROUTES["x3_group_norm"] += 1
    if to_f32:
        res32 = None
        if residual is not None:
            flags |= EPI_ADD_POST if residual_after_act else EPI_ADD_PRE
            res32 = ops.from_split(residual.t, residual.exp)
        return ops.affine_act(raw, scale, shift, res32, flags, per_sample=True, out=out if out is not None else raw)
    bound = x3_norm_bound(norm, plan)
    res_t, res_exp = None, 0
    if residual is not None:
        flags |= EPI_ADD_POST if residual_after_act else EPI_ADD_PRE
        if residual.bound is None:
            raise RuntimeError("split mode: a residual add needs bounds on both summands")
        bound = bound + residual.bound
        res_t, res_exp = residual.t, residual.exp
    e = x3_exponent(bound) if out_exp is None else out_exp
    y = ops.affine_act_split(raw, scale, shift, e, residual=res_t, res_exp=res_exp, flags=flags, per_sample=True, out=out, overflow=flag)
    return SplitT(y, e, bound)


class ConvBN3d(nn.Sequential):
    """``Sequential(Conv3d | ConvTranspose3d, BatchNorm3d | GroupNorm)`` -- the object convbn_3d
    returns in the reference (keys ``0.weight``, ``1.weight``, ``1.bias``, ``1.running_mean``, ...)."""

    def forward(self, x):
        return self.fused(x)

    def fused(self, x, **kw):
        return fused_conv3d(self[0], self[1], x, **kw)

    def fused_f16(self, x, **kw):
        return fused_conv3d_f16(self[0], self[1], x, **kw)

    def fused_x3(self, x, **kw):
        return fused_conv3d_x3(self[0], self[1], x, **kw)


class HipConv3d(nn.Conv3d):
    """A bare nn.Conv3d(bias=False) (classifier, fg_cls_head[2], part_reg_head[2]) on the HIP kernel."""

    def forward(self, x):
        return self.fused(x)

    def fused(self, x, **kw):
        return fused_conv3d(self, None, x, **kw)

    def fused_f16(self, x, **kw):
        return fused_conv3d_f16(self, None, x, **kw)

    def fused_x3(self, x, **kw):
        return fused_conv3d_x3(self, None, x, **kw)


def convbn_3d(in_planes, out_planes, kernel_size, stride, pad, dilation=1, gn=False, groups=32):
    """reference submodule.py:32-50"""
    return ConvBN3d(nn.Conv3d(in_planes, out_planes, kernel_size=kernel_size, padding=pad, dilation=dilation,
                              stride=stride, bias=False),
                    nn.BatchNorm3d(out_planes) if not gn else nn.GroupNorm(groups, out_planes))


def _deconvbn_3d(cin, cout, gn):
    return ConvBN3d(nn.ConvTranspose3d(cin, cout, kernel_size=3, padding=1, output_padding=1, stride=2, bias=False),
                    nn.BatchNorm3d(cout) if not gn else nn.GroupNorm(32, cout))


class ConvBNReLU3d(nn.Sequential):
    """``Sequential(convbn_3d(...), ReLU(inplace=True))``: the ReLU is folded into the conv epilogue."""

    def forward(self, x):
        return self[0].fused(x, relu=True)

    def fused(self, x, **kw):
        kw.setdefault("relu", True)
        return self[0].fused(x, **kw)

    def fused_f16(self, x, **kw):
        kw.setdefault("relu", True)
        return self[0].fused_f16(x, **kw)

    def fused_x3(self, x, **kw):
        kw.setdefault("relu", True)
        return self[0].fused_x3(x, **kw)


class disparityregression(nn.Module):
    """reference submodule.py:76-83 (its constructor builds an unused .cuda() buffer; not kept)."""

    def __init__(self, maxdisp=None, cfg=None):
        super().__init__()
        self.maxdisp = maxdisp

    def forward(self, x, depth):
        return ops.disparity_regression(x, depth)


class hourglass(nn.Module):
    """reference submodule.py:85-168"""

    def __init__(self, inplanes, gn=False):
        super().__init__()
        c = inplanes
        self.conv1 = ConvBNReLU3d(convbn_3d(c, c * 2, kernel_size=3, stride=2, pad=1, gn=gn), nn.ReLU(inplace=True))
        self.conv2 = convbn_3d(c * 2, c * 2, kernel_size=3, stride=1, pad=1, gn=gn)
        self.conv3 = ConvBNReLU3d(convbn_3d(c * 2, c * 2, kernel_size=3, stride=2, pad=1, gn=gn), nn.ReLU(inplace=True))
        self.conv4 = ConvBNReLU3d(convbn_3d(c * 2, c * 2, kernel_size=3, stride=1, pad=1, gn=gn), nn.ReLU(inplace=True))
        self.conv5 = _deconvbn_3d(c * 2, c * 2, gn)
        self.conv6 = _deconvbn_3d(c * 2, c, gn)

    def forward(self, x, presqu, postsqu, residual=None, out=None, head=None, head_residual=None):
        """Returns (out, pre, post).  ``residual``/``out`` (extension): fold the caller's
        ``x + hourglass(x)[0]`` (vernier.py:370,421) into the last deconvolution's epilogue.
        ``head`` (extension): a 1x1x1 one-channel convolution applied to ``out`` (see fused_conv3d: conv6 has no
        activation, so at inference conv6 + head fold into one transposed layer to one channel); ``out`` is then
        ``head(out)``.  ``head_residual`` = ``head(residual)`` when the caller already has it."""
        # training: the two skip connections (x -> conv6's residual when the caller folds it, pre -> conv5's residual) hand their
        # gradients to conv1's / conv3's data-gradient kernels instead of to an autograd add (_SkipTap)
        box_x = skip_box(x) if residual is x else None
        o = self.conv1.fused(x, grad_box=box_x)                             # 1/2 res, ReLU fused
        if box_x is not None:
            residual = _SkipTap.apply(x, box_x)
        pre = self.conv2.fused(o, relu=True, residual=postsqu)              # relu(bn(conv) [+ postsqu]) :153-156
        box_pre = skip_box(pre) if presqu is None else None
        o = self.conv3.fused(pre, grad_box=box_pre)
        pre_skip = _SkipTap.apply(pre, box_pre) if box_pre is not None else pre
        o = self.conv4(o)                                                   # 1/4 res
        post = self.conv5.fused(o, relu=True, residual=presqu if presqu is not None else pre_skip)  # :161-164
        o = self.conv6.fused(post, residual=residual, out=out, head=head, head_residual=head_residual)   # :166
        return o, pre, post

    def forward_x3(self, x, residual=None, out=None, out_exp=None, flag=None):
        """The same graph in split mode (SplitT in / out; presqu / postsqu = None as the callers on the path use it)."""
        o = self.conv1.fused_x3(x, flag=flag)
        pre = self.conv2.fused_x3(o, relu=True, flag=flag)
        o = self.conv4.fused_x3(self.conv3.fused_x3(pre, flag=flag), flag=flag)
        post = self.conv5.fused_x3(o, relu=True, residual=pre, flag=flag)
        o = self.conv6.fused_x3(post, residual=residual, out=out, out_exp=out_exp, flag=flag)
        return o, pre, post

    def forward_f16(self, x, presqu=None, postsqu=None, residual=None, out=None):
        """The same graph on C8 half tensors (fp16-storage mode)."""
        o = self.conv1.fused_f16(x)
        pre = self.conv2.fused_f16(o, relu=True, residual=postsqu)
        o = self.conv4.fused_f16(self.conv3.fused_f16(pre))
        post = self.conv5.fused_f16(o, relu=True, residual=presqu if presqu is not None else pre)
        o = self.conv6.fused_f16(post, residual=residual, out=out)
        return o, pre, post


def get_hg_down_sample(channel_in, channel_out, gn, downsample=True):
    """reference submodule.py:170-181"""
    return ConvBNReLU3d(convbn_3d(channel_in, channel_out, kernel_size=3, stride=2 if downsample else 1, pad=1, gn=gn),
                        nn.ReLU(inplace=True))


def get_hg_up_sample(channel_in, channel_out, gn):
    """reference submodule.py:197-208"""
    return _deconvbn_3d(channel_in, channel_out, gn)


class hourglass_downsample_16(nn.Module):
    """reference submodule.py:223-268"""

    def __init__(self, inplanes, gn=False):
        super().__init__()
        c = inplanes
        self.conv1 = get_hg_down_sample(c, c * 2, gn)
        self.conv2 = get_hg_down_sample(c * 2, c * 2, gn, False)
        self.conv3 = get_hg_down_sample(c * 2, c * 2, gn)
        self.conv4 = get_hg_down_sample(c * 2, c * 2, gn, False)
        self.conv5 = get_hg_down_sample(c * 2, c * 2, gn)
        self.conv6 = get_hg_down_sample(c * 2, c * 2, gn, False)
        self.conv7 = get_hg_down_sample(c * 2, c * 2, gn)
        self.conv8 = get_hg_down_sample(c * 2, c * 2, gn, False)
        self.conv9 = get_hg_up_sample(c * 2, c * 2, gn)
        self.conv10 = get_hg_up_sample(c * 2, c * 2, gn)
        self.conv11 = get_hg_up_sample(c * 2, c * 2, gn)
        self.conv12 = get_hg_up_sample(c * 2, c, gn)

    def forward(self, x, residual=None, out=None):
        o2 = self.conv2(self.conv1(x))
        o4 = self.conv4(self.conv3(o2))
        o6 = self.conv6(self.conv5(o4))
        o8 = self.conv8(self.conv7(o6))
        i10 = self.conv9.fused(o8, residual=o6)     # out_conv9 + out_conv6   :258-259
        i11 = self.conv10.fused(i10, residual=o4)   # out_conv10 + out_conv4  :261-262
        i12 = self.conv11.fused(i11, residual=o2)   # out_conv11 + out_conv2  :264-266
        return self.conv12.fused(i12, residual=residual, out=out)

    def forward_x3(self, x, residual=None, out=None, out_exp=None, flag=None):
        """The same graph in split mode (SplitT in / out)."""
        f = lambda seq, t, **kw: seq.fused_x3(t, flag=flag, **kw)                              # noqa: E731
        o2 = f(self.conv2, f(self.conv1, x))
        o4 = f(self.conv4, f(self.conv3, o2))
        o6 = f(self.conv6, f(self.conv5, o4))
        o8 = f(self.conv8, f(self.conv7, o6))
        i10 = f(self.conv9, o8, residual=o6)
        i11 = f(self.conv10, i10, residual=o4)
        i12 = f(self.conv11, i11, residual=o2)
        return f(self.conv12, i12, residual=residual, out=out, out_exp=out_exp)

    def forward_f16(self, x, residual=None, out=None):
        """The same graph on C8 half tensors (fp16-storage mode)."""
        o2 = self.conv2.fused_f16(self.conv1.fused_f16(x))
        o4 = self.conv4.fused_f16(self.conv3.fused_f16(o2))
        o6 = self.conv6.fused_f16(self.conv5.fused_f16(o4))
        o8 = self.conv8.fused_f16(self.conv7.fused_f16(o6))
        i10 = self.conv9.fused_f16(o8, residual=o6)
        i11 = self.conv10.fused_f16(i10, residual=o4)
        i12 = self.conv11.fused_f16(i11, residual=o2)
        return self.conv12.fused_f16(i12, residual=residual, out=out)


# ------------------------------------------------------------------------------------------
# 2D BEV neck: same module tree / state-dict keys as the reference; inference with eval-mode
# BatchNorm2d runs on the depth-1 form of the HIP conv kernels (an NCHW tensor IS an [N,C,1,H,W] tensor), with the
# folded norm, the conv bias, residual adds, ReLU and Sigmoid in the conv epilogue; GroupNorm (cfg.gn) takes the 3D
# layers' three-launch form (conv, statistics, normalise + residual + activation).  Anything with a gradient to
# compute takes the modules' own torch forward.
# ------------------------------------------------------------------------------------------
def _hip_2d_ok(x: torch.Tensor, *modules) -> bool:
    """The HIP path of the 2D neck: a float32 GPU tensor and every norm of ``modules`` a BatchNorm2d or a GroupNorm.
    Nothing to differentiate and eval-mode BatchNorm: one fused launch per layer (norm folded into the epilogue; GroupNorm:
    conv -> statistics -> normalise + residual + activation, like the 3D layers).  Autograd on with something that requires
    grad, or train-mode BatchNorm: the same kernels behind ``_Conv2dNormActFn`` (r4) -- forward conv -> statistics ->
    normalise, backward on the HIP epilogue-backward reductions, the data gradient as another depth-1 layer of the family
    and the deterministic weight gradient -- so ``VernierScale`` trains through its neck natively (reference
    vernier.py:296-313,438-450).  Every decision is counted in ``_ROUTES``: "neck2d_hip" / "neck2d_torch" per block,
    "neck2d_hip_train" per layer that went through the autograd function."""
    if not x.is_cuda:
        raise RuntimeError("2D neck input must be a GPU tensor: Not implemented on the CPU")
    norms = [n for m in modules for n in _norms2d(m)]
    ok = x.dtype == torch.float32 and all(isinstance(n, (nn.GroupNorm, nn.BatchNorm2d)) for n in norms)
    if ok and not NECK2D_HIP_TRAINING[0]:
        ok = not _wants_grad2d(x, *modules) and all(isinstance(n, nn.GroupNorm) or (not n.training and n.running_mean is not None)
                                                    for n in norms)
    _ROUTES["neck2d_hip" if ok else "neck2d_torch"] += 1
    return ok


NECK2D_HIP_TRAINING = [True]      # False: rounds 2-3's behaviour (anything with a gradient keeps the modules' torch forward)


def _wants_grad2d(x, *modules) -> bool:
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for m in modules for p in m.parameters()))


def _train2d(x, conv, norm, residual) -> bool:
    """This layer goes through the autograd function: something to differentiate, or batch statistics to take."""
    if isinstance(norm, nn.BatchNorm2d) and (norm.training or norm.running_mean is None):
        return True
    mods = (conv,) if norm is None else (conv, norm)
    return _wants_grad2d(x, *mods) or (torch.is_grad_enabled() and residual is not None and residual.requires_grad)


class _Conv2dNormActFn(torch.autograd.Function):
    """One layer of the 2D neck with a backward: y = act(norm(conv(x) [+ conv bias]) [+ res]) [+ res] on 5-D depth-1 tensors.
    ``kind``: ("conv", k, stride) for Conv2d(k in {1, 3}, stride in {1, 2}, padding (k-1)/2), ("deconv",) for
    ConvTranspose2d(k3, s2, p1, op1), ("whole",) for the kernel that covers its whole input (a 1x1 layer over the flattened
    input).  The data gradient is another depth-1 layer of the family (taps flipped / the transposed twin over the same
    weight memory / the stride-2 twin); the weight gradient comes from snvc_conv3d_wgrad on the depth-1 tensors with a cubic
    kernel whose middle depth tap is the 2D kernel (the other two taps see only padding)."""

    @staticmethod
    def forward(ctx, x5, weight, cbias, gamma, beta, res5, conv, norm, flags, plan, kind, layer):
        raw = layer(x5, None, None, None, 0, None)
        if cbias is not None:
            if norm is not None:
                raise NotImplementedError("a 2D layer with both a conv bias and a norm is not in the neck")
            scale, shift = torch.ones_like(cbias.detach()), cbias.detach().float().contiguous()
            y, mean, var, per_sample = ops.affine_act(raw, scale, shift, res5, flags), None, None, False
        else:
            y, scale, shift, mean, var, per_sample = _norm_from_raw(raw, norm, plan, res5, flags)
        ctx.conv, ctx.norm, ctx.flags, ctx.plan, ctx.kind, ctx.per_sample = conv, norm, flags, plan, kind, per_sample
        ctx.has_res, ctx.train_stats, ctx.has_bias = res5 is not None, mean is not None, cbias is not None
        ctx.save_for_backward(x5, raw, scale, shift, mean, var, res5 if (res5 is not None and (flags & EPI_ADD_PRE)) else None)
        _ROUTES["neck2d_hip_train"] += 1
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x5, raw, scale, shift, mean, var, res = ctx.saved_tensors
        conv, norm, flags, plan, kind = ctx.conv, ctx.norm, ctx.flags, ctx.plan, ctx.kind
        needs = ctx.needs_input_grad
        draw, gres, dg, db = _epilogue_backward(raw, gy, res, scale, shift, mean, var, norm, flags, ctx.per_sample, ctx.train_stats,
                                                ctx.has_res and needs[5], needs[3], needs[4])
        gb = draw.sum(dim=(0, 2, 3, 4)) if (ctx.has_bias and needs[2]) else None
        w = conv.weight.detach()
        key = (w.data_ptr(), w._version, w.device, _GENERATION[0])
        if getattr(plan, "dgrad2d_key", None) != key:
            mk = lambda t, k, st, tr: ops.Conv3dLayer(t.contiguous(), k, st, (k - 1) // 2, 1, tr, planar=True)   # noqa: E731
            if kind[0] == "deconv":
                plan.dgrad2d = mk(w, 3, 2, False)                                   # Conv2d(k3,s2,p1) over the same weight memory
            elif kind[0] == "whole":
                plan.dgrad2d = ops.Conv3dLayer(w.reshape(w.size(0), -1).t().reshape(-1, w.size(0), 1, 1, 1).contiguous(), 1, 1, 0, 1,
                                               False, planar=True)
            elif kind[2] == 1:
                plan.dgrad2d = mk(w.transpose(0, 1).flip(2, 3), kind[1], 1, False)
            elif kind[1] == 3:
                plan.dgrad2d = mk(w, 3, 2, True)                                    # ConvTranspose2d(k3,s2,p1,op1), same weight memory
            else:
                plan.dgrad2d = mk(w.transpose(0, 1), 1, 1, False)                   # k1 / stride 2: a 1x1 layer, scattered below
            plan.dgrad2d_key = key
        gx = gw = None
        h, wd = x5.size(3), x5.size(4)
        if needs[0]:
            g = plan.dgrad2d(draw, None, None, None, 0, None)
            if kind[0] == "conv" and kind[2] == 2:
                if kind[1] == 3:        # twice draw's extent: crop the gradient of the padding when x's extent is odd
                    gx = g if (g.size(3), g.size(4)) == (h, wd) else g[:, :, :, :h, :wd].contiguous()
                else:                   # Conv2d(k1, s2) reads the even positions only
                    gx = torch.zeros_like(x5)
                    gx[:, :, :, ::2, ::2] = g
            else:
                gx = g
        if needs[1]:
            if kind[0] == "deconv":     # roles swapped, see snvc_conv3d_wgrad
                gw = ops.conv3d_wgrad(draw, x5, 3, 2, 1, 1)[:, :, 1]
            elif kind[0] == "whole":
                gw = ops.conv3d_wgrad(x5, draw, 1, 1, 0, 1).reshape(w.shape)
            elif kind[1] == 1:
                xs = x5 if kind[2] == 1 else x5[:, :, :, ::2, ::2].contiguous()
                gw = ops.conv3d_wgrad(xs, draw, 1, 1, 0, 1).reshape(w.shape)
            else:
                xs = x5
                if kind[2] == 2 and (h % 2 or wd % 2):
                    xs = F.pad(x5, (0, wd % 2, 0, h % 2))
                gw = ops.conv3d_wgrad(xs, draw, 3, kind[2], 1, 1)[:, :, 1]
            gw = gw.contiguous()
        return gx, gw, gb, dg, db, gres, None, None, None, None, None, None


def _conv2d_train(conv, norm, x5, res5, flags, plan, kind, layer):
    gamma = norm.weight if norm is not None else None
    beta = norm.bias if norm is not None else None
    return _Conv2dNormActFn.apply(x5, conv.weight, conv.bias, gamma, beta, res5, conv, norm, flags, plan, kind, layer)


def _plan2d(conv: nn.Module, device) -> _Plan:
    return conv.__dict__.setdefault("_snvc_plans2d", {}).setdefault(device, _Plan())


def _affine2d(conv, norm, plan: _Plan):
    """(scale, bias) of the epilogue: folded eval BatchNorm2d and / or the conv's own bias."""
    scale = bias = None
    if isinstance(norm, nn.BatchNorm2d):
        scale, bias = _folded_bn(norm, plan)
    if conv.bias is not None:
        key = (conv.bias._version, conv.bias.data_ptr(), None if scale is None else scale.data_ptr(), _GENERATION[0])
        if getattr(plan, "cb_key", None) != key:
            b = conv.bias.detach().float()
            plan.cb = ((b * scale + bias) if scale is not None else b).contiguous()
            plan.cs = scale if scale is not None else torch.ones_like(b)
            plan.cb_key = key
        scale, bias = plan.cs, plan.cb
    return scale, bias


def fused_conv2d(conv: nn.Conv2d, norm, x: torch.Tensor, *, relu=False, sigmoid=False, residual=None,
                 residual_after_act=False, transposed_input=False) -> torch.Tensor:
    """act(norm(conv(x)) [+ residual]) [+ residual] for the 2D neck (reference submodule.py:11-29, hrnet.py:25-69)
    on the depth-1 HIP kernels.  conv: Conv2d(k in {1,3}, stride in {1,2}, padding=(k-1)/2) -- or a Conv2d whose
    kernel covers its whole input (the coordinate head's last layer, vernier.py:87), run as a 1x1 layer.
    ``transposed_input``: returns ``conv(x.transpose(2, 3))`` as a transposed VIEW of the convolution of x itself with the
    kernel's two spatial axes swapped (a convolution commutes with swapping H and W if its kernel is swapped too): the
    copy that ``permute(0, 1, 3, 2).contiguous()`` would make of the C-channel input is not made (vernier.py:441-442)."""
    plan = _plan2d(conv, x.device)
    w = conv.weight
    kh, kw = conv.kernel_size
    whole = (kh, kw) == tuple(x.shape[2:]) and tuple(conv.padding) == (0, 0) and (kh, kw) != (1, 1)
    if transposed_input:
        if whole or kh != kw or residual is not None or isinstance(norm, nn.GroupNorm) or _train2d(x, conv, norm, residual):
            return fused_conv2d(conv, norm, x.transpose(2, 3).contiguous(), relu=relu, sigmoid=sigmoid, residual=residual,
                                residual_after_act=residual_after_act)
        plan = conv.__dict__.setdefault("_snvc_plans2d_t", {}).setdefault(x.device, _Plan())
    key = (w.data_ptr(), w._version, w.device, whole, _GENERATION[0])
    if plan.layer is None or plan.key != key:
        if conv.groups != 1 or tuple(conv.dilation) != (1, 1):
            raise NotImplementedError("grouped / dilated Conv2d is not in the 2D neck")
        if whole:      # one output pixel: a 1x1 layer over the flattened (c, h, w) input
            plan.layer = ops.Conv3dLayer(w.detach().reshape(w.size(0), -1, 1, 1, 1), 1, 1, 0, 1, False, planar=True)
        else:
            k, st = kh, conv.stride[0]
            if kh != kw or conv.stride[0] != conv.stride[1] or tuple(conv.padding) != ((k - 1) // 2,) * 2 or k not in (1, 3) or st not in (1, 2):
                raise NotImplementedError(f"Conv2d geometry {conv} is not in the 2D neck")
            wk = w.detach().transpose(2, 3).contiguous() if transposed_input else w.detach()
            plan.layer = ops.Conv3dLayer(wk, k, st, (k - 1) // 2, 1, False, planar=True)
        plan.key = key
    flags = (EPI_RELU if relu else 0) | (EPI_SIGMOID if sigmoid else 0)
    if residual is not None:
        flags |= EPI_ADD_POST if residual_after_act else EPI_ADD_PRE
    x5 = x.reshape(x.size(0), -1, 1, 1, 1) if whole else x.unsqueeze(2)
    r5 = residual.unsqueeze(2) if residual is not None else None
    if _train2d(x, conv, norm, residual):
        kind = ("whole",) if whole else ("conv", kh, conv.stride[0])
        y = _conv2d_train(conv, norm, x5.contiguous(), r5.contiguous() if r5 is not None else None, flags, plan, kind, plan.layer)
        return y.squeeze(2)
    scale, bias = _affine2d(conv, norm, plan)
    if isinstance(norm, nn.GroupNorm):
        return _group_norm_2d(plan.layer(x5, scale, bias), norm, r5, flags).squeeze(2)
    y = plan.layer(x5, scale, bias, r5, flags).squeeze(2)
    return y.transpose(2, 3) if transposed_input else y


def _group_norm_2d(raw5, norm: nn.GroupNorm, r5, flags):
    """GroupNorm of a conv output on the HIP kernels (reference submodule.py:11-29 with gn=True): per-(sample, group)
    statistics in fp64, then normalise + affine + residual + activation in one pass, in place."""
    scale, shift, _, _ = ops.norm_stats(raw5, norm.weight, norm.bias, norm.num_groups, True, norm.eps)
    return ops.affine_act(raw5, scale, shift, r5, flags, per_sample=True, out=raw5)


def fused_deconv2d(conv: nn.ConvTranspose2d, norm, x: torch.Tensor, *, relu=False, residual=None) -> torch.Tensor:
    """ConvTranspose2d(k3,s2,p1,op1) (+norm) (+residual) (+ReLU) (reference submodule.py:291-314) on the depth-1 form of the
    parity-class transposed kernel (r2-r3 first form: zero-stuffed input + a k3 convolution with the flipped kernel, i.e.
    four times the multiply-adds and one more launch; kept as ``ops.zero_stuff2x`` for the tests)."""
    if (tuple(conv.kernel_size), tuple(conv.stride), tuple(conv.padding), tuple(conv.output_padding)) != ((3, 3), (2, 2), (1, 1), (1, 1)) \
            or conv.groups != 1 or conv.bias is not None:
        raise NotImplementedError("the 2D neck's up-sampling layers are ConvTranspose2d(k3,s2,p1,op1,bias=False)")
    plan = _plan2d(conv, x.device)
    w = conv.weight
    key = (w.data_ptr(), w._version, w.device, _GENERATION[0])
    if plan.layer is None or plan.key != key:
        plan.layer = ops.Conv3dLayer(w.detach().contiguous(), 3, 2, 1, 1, True, planar=True)
        plan.key = key
    flags = (EPI_RELU if relu else 0) | (EPI_ADD_PRE if residual is not None else 0)
    r5 = residual.unsqueeze(2) if residual is not None else None
    if _train2d(x, conv, norm, residual):
        return _conv2d_train(conv, norm, x.unsqueeze(2).contiguous(), r5.contiguous() if r5 is not None else None, flags, plan,
                             ("deconv",), plan.layer).squeeze(2)
    scale, bias = _affine2d(conv, norm, plan)
    if isinstance(norm, nn.GroupNorm):
        return _group_norm_2d(plan.layer(x.unsqueeze(2), scale, bias), norm, r5, flags).squeeze(2)
    return plan.layer(x.unsqueeze(2), scale, bias, r5, flags).squeeze(2)


def _cbr2d(seq, x, **kw):
    """Sequential(convbn(...), ReLU) or a bare convbn Sequential(conv, norm) on the HIP path."""
    if isinstance(seq[0], nn.Sequential):          # Sequential(convbn, ReLU)
        kw.setdefault("relu", True)
        seq = seq[0]
    if isinstance(seq[0], nn.ConvTranspose2d):
        return fused_deconv2d(seq[0], seq[1], x, **kw)
    return fused_conv2d(seq[0], seq[1], x, **kw)


def _norms2d(module):
    return [m for m in module.modules() if isinstance(m, (nn.BatchNorm2d, nn.GroupNorm))]


def convbn(in_planes, out_planes, kernel_size, stride, pad, dilation, gn=False, groups=32):
    """reference submodule.py:11-29"""
    return nn.Sequential(nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride,
                                   padding=dilation if dilation > 1 else pad, dilation=dilation, bias=False),
                         nn.BatchNorm2d(out_planes) if not gn else nn.GroupNorm(groups, out_planes))


def _deconvbn_2d(cin, cout, gn):
    return nn.Sequential(nn.ConvTranspose2d(cin, cout, kernel_size=3, padding=1, output_padding=1, stride=2, bias=False),
                         nn.BatchNorm2d(cout) if not gn else nn.GroupNorm(32, cout))


def get_hg_down_sample_2d(channel_in, channel_out, gn, downsample=True):
    return nn.Sequential(convbn(channel_in, channel_out, kernel_size=3, stride=2 if downsample else 1, pad=1,
                                dilation=1, gn=gn), nn.ReLU(inplace=True))


def get_hg_up_sample_2d(channel_in, channel_out, gn):
    return _deconvbn_2d(channel_in, channel_out, gn)


class hourglass2d(nn.Module):
    """reference submodule.py:317-361"""

    def __init__(self, inplanes, gn=False):
        super().__init__()
        c = inplanes
        self.conv1 = nn.Sequential(convbn(c, c * 2, 3, 2, 1, 1, gn=gn), nn.ReLU(inplace=True))
        self.conv2 = convbn(c * 2, c * 2, 3, 1, 1, 1, gn=gn)
        self.conv3 = nn.Sequential(convbn(c * 2, c * 2, 3, 2, 1, 1, gn=gn), nn.ReLU(inplace=True))
        self.conv4 = nn.Sequential(convbn(c * 2, c * 2, 3, 1, 1, 1, gn=gn), nn.ReLU(inplace=True))
        self.conv5 = _deconvbn_2d(c * 2, c * 2, gn)
        self.conv6 = _deconvbn_2d(c * 2, c, gn)

    def forward(self, x, presqu, postsqu):
        if _hip_2d_ok(x, self):
            out = _cbr2d(self.conv1, x)
            pre = _cbr2d(self.conv2, out, relu=True, residual=postsqu)              # relu(bn(conv) [+ postsqu])
            out = _cbr2d(self.conv4, _cbr2d(self.conv3, pre))
            post = _cbr2d(self.conv5, out, relu=True, residual=presqu if presqu is not None else pre)
            return _cbr2d(self.conv6, post), pre, post
        out = self.conv1(x)
        pre = self.conv2(out)
        pre = F.relu(pre + postsqu, inplace=True) if postsqu is not None else F.relu(pre, inplace=True)
        out = self.conv4(self.conv3(pre))
        post = F.relu(self.conv5(out) + (presqu if presqu is not None else pre), inplace=True)
        return self.conv6(post), pre, post


class hourglass2d_downsample_16(nn.Module):
    """reference submodule.py:270-315"""

    def __init__(self, inplanes, gn=False):
        super().__init__()
        c = inplanes
        self.conv1 = get_hg_down_sample_2d(c, c * 2, gn)
        self.conv2 = get_hg_down_sample_2d(c * 2, c * 2, gn, False)
        self.conv3 = get_hg_down_sample_2d(c * 2, c * 2, gn)
        self.conv4 = get_hg_down_sample_2d(c * 2, c * 2, gn, False)
        self.conv5 = get_hg_down_sample_2d(c * 2, c * 2, gn)
        self.conv6 = get_hg_down_sample_2d(c * 2, c * 2, gn, False)
        self.conv7 = get_hg_down_sample_2d(c * 2, c * 2, gn)
        self.conv8 = get_hg_down_sample_2d(c * 2, c * 2, gn, False)
        self.conv9 = get_hg_up_sample_2d(c * 2, c * 2, gn)
        self.conv10 = get_hg_up_sample_2d(c * 2, c * 2, gn)
        self.conv11 = get_hg_up_sample_2d(c * 2, c * 2, gn)
        self.conv12 = get_hg_up_sample_2d(c * 2, c, gn)

    def forward(self, x):
        if _hip_2d_ok(x, self):
            o2 = _cbr2d(self.conv2, _cbr2d(self.conv1, x))
            o4 = _cbr2d(self.conv4, _cbr2d(self.conv3, o2))
            o6 = _cbr2d(self.conv6, _cbr2d(self.conv5, o4))
            o8 = _cbr2d(self.conv8, _cbr2d(self.conv7, o6))
            i10 = _cbr2d(self.conv9, o8, residual=o6)        # the skip adds live in the up-sampling layers' epilogues
            i11 = _cbr2d(self.conv10, i10, residual=o4)
            i12 = _cbr2d(self.conv11, i11, residual=o2)
            return _cbr2d(self.conv12, i12)
        o2 = self.conv2(self.conv1(x))
        o4 = self.conv4(self.conv3(o2))
        o6 = self.conv6(self.conv5(o4))
        o8 = self.conv8(self.conv7(o6))
        o10 = self.conv10(self.conv9(o8) + o6)
        o11 = self.conv11(o10 + o4)
        return self.conv12(o11 + o2)


class BasicBlock2d(nn.Module):
    """snvc/models/hrnet.py:25-54 (used by the coordinate head, vernier.py:68-93)"""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes, momentum=0.1)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, momentum=0.1)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if _hip_2d_ok(x, self):
            residual = x if self.downsample is None else fused_conv2d(self.downsample[0], self.downsample[1], x)
            out = fused_conv2d(self.conv1, self.bn1, x, relu=True)
            return fused_conv2d(self.conv2, self.bn2, out, relu=True, residual=residual)     # relu(bn2(conv2) + residual)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        residual = x if self.downsample is None else self.downsample(x)
        return self.relu(out + residual)


def basicdownsample(in_planes, out_planes):
    """snvc/models/hrnet.py:56-69"""
    return nn.Sequential(nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=2, bias=False),
                         nn.BatchNorm2d(out_planes))
