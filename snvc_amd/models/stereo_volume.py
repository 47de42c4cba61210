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
import itertools
import math
import warnings
import weakref

import torch
import torch.nn as nn

from ..extension.build_cost_volume import _BuildCostVolume, build_cost_volume  # noqa: F401  (re-exported)
from .. import ops
from .submodule import (_GENERATION, _ROUTES, ConvBNReLU3d, HipConv3d, SplitOverflow, _FactoredFirstConvFn, _ShearedFirstConvBNFn,
                        _ShearedFirstConvFn, _folded_bn, _is_channel_head as _is_head_conv, _Plan, convbn_3d, folded_head_weights,
                        hourglass, overflow_guard, sheared_geometry, sheared_kernels, EPI_RELU)

_PREP_EPOCH = itertools.count(1)      # stamps of the per-model first-layer prep buffers (see _forward_pair_steps)


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

    def _buffer(self, name, shape, device, dtype=torch.float32):
        """Inference workspace: the three full-resolution intermediates (warped half-volume, conv1 / conv2 outputs,
        0.74 GB each at cfg2) are kept across calls instead of going through the caching allocator every step -- with
        GB-sized blocks that are split and re-merged, a steady-state step can land on a fresh hipMalloc (tens of ms).
        They never escape ``forward`` / ``forward_pair`` (the result is the head's own small tensor).
        One buffer per (name, device): replicas on different devices keep their own.  Single-stream: the buffers
        are not tracked by the caching allocator's stream bookkeeping, so calls on two streams of one device would
        alias them.  ``release_workspace()`` / ``invalidate_plans(model)`` / ``train()`` give the memory back."""
        ws = self.__dict__.setdefault("_snvc_ws", {})
        key = (name, tuple(shape), device)
        buf = ws.get(key)
        if buf is None:
            for k in [k for k in ws if k[0] == name and k[2] == device]:
                del ws[k]
            buf = ws[key] = torch.empty(shape, dtype=dtype, device=device)
        return buf

    # the sheared first layer's two small 2D chains on side streams: built and measured in r5 (tools/ab_step.py, interleaved legs):
    # 2.185 ms/step against 2.130 on one stream -- the cross-stream event waits cost more than the chains' overlap buys.  Off.
    prep_streams = False

    def _side_streams(self, device):
        st = self.__dict__.setdefault("_snvc_streams", {})
        if device not in st:
            st[device] = (torch.cuda.Stream(device), torch.cuda.Stream(device))
        return st[device]

    def last_first_layer(self) -> torch.Tensor:
        """The first layer's result of the most recent inference call as a float32 [N,C,D,H,W] tensor (a copy when the call ran
        in split mode and the workspace holds the (hi, lo) pair) -- for tests and debugging."""
        ws = self.__dict__.get("_snvc_ws", {})
        which = self.__dict__.get("_snvc_last_v1", "v1")
        for k, v in ws.items():
            if k[0] == which:
                return ops.from_split(v, self.__dict__["_snvc_x3"]["exp"]["v1"]) if which == "v1s" else v
        raise RuntimeError("no first-layer result in the workspace")

    def release_workspace(self):
        """Drop the persistent inference workspace (2.2 GB at cfg2)."""
        self.__dict__.pop("_snvc_ws", None)
        self.__dict__.pop("_snvc_prep_ws", None)

    def train(self, mode: bool = True):
        if mode:
            self.release_workspace()
        return super().train(mode)

    def __getstate__(self):          # copy.deepcopy / torch.save(model): the workspace is scratch, not state
        state = self.__dict__.copy()
        state.pop("_snvc_ws", None)
        state.pop("_snvc_x3", None)      # packed split-mode layers: rebuilt on first use
        state.pop("_snvc_x3_guard", None)    # the overflow flag, its pinned host copy and an event
        state.pop("_snvc_streams", None)
        state.pop("_snvc_prep_ws", None)
        return state

    # ------------------------------------------------------------------------------------------ split mode ("f16x3", r4)
    # Inference with frozen statistics: conv2 and the hourglass's five MFMA layers run on the split-mode kernels
    # (csrc/conv3d_f16.hip, F16Cfg::PL): the SAME fp32 layers -- values travel as (hi, lo) pairs of halves (22 bits), a product
    # is three v_mfma_f32_32x32x16_f16 with fp32 accumulation, measured 5e-7 of the range against float64 where the fp32
    # Winograd kernels measure 2e-6 -- on a matrix pipe 16 times faster than the fp32 one.  ``arithmetic``: "auto" (default:
    # split mode when the stack qualifies), "fp32" (the fp32-MFMA kernels everywhere), "x3" (split mode or an error).
    # A split tensor's exponent is chosen from its BatchNorm's parameters; a value beyond that range is clamped by the epilogue
    # and FLAGGED.  ``overflow_check`` = "call" (default, r5): the flag is read before the result leaves the call -- the copy is
    # queued behind the last layer that can clamp, the host waits for it after queueing the rest, so the GPU never idles -- and
    # a flagged call is REDONE on the fp32-MFMA kernels ("auto": with a warning, and split mode stays off for this model;
    # "x3": RuntimeError).  No clamped result is ever returned.  "deferred" (r4's behaviour, for measuring what the check
    # costs): the flag is only posted; ``check_overflow()`` or the next call looks at it.
    arithmetic = "auto"
    overflow_check = "call"
    # the first layer's 0.74 GB split pair written with non-temporal stores (SNVC_EPI_STREAM_OUT): the expand pass alone gains 9 % (182 ->
    # 165 us back to back) but the step LOSES 0.013 ms (tools/ab_step.py, profiles/r5/ab_step_v5.txt): conv2 reads the pair right
    # behind it.  Off.
    stream_out = False
    split_prep = True    # split mode: the sheared first layer's depth-1 3 x 7 layers (G, G') on the half pipe too (False: fp32 MFMA, as r4)
    fused_tail = True    # split mode: conv5's epilogue contracts its result with the folded one-channel tail (False: r4's two launches)
    X3_SIGMAS = 64.0     # a tensor's exponent is chosen so that |beta| + X3_SIGMAS * |gamma| of its BatchNorm stays below 2^15

    @staticmethod
    def _x3_exponent(bound: float) -> int:
        """e with bound * 2^e <= 2^15 (half overflows at 65504; a value beyond the bound is clamped and FLAGGED)."""
        if not (bound > 0.0) or not math.isfinite(bound):
            return 0
        return max(-14, min(14, 15 - math.frexp(bound)[1]))

    def _x3_state(self, device):
        """Packed split-mode layers, folded affines and per-tensor exponents of conv2 + the hourglass, or None when the stack
        does not qualify (channels, norm kind, train mode).  Cached on the parameters' versions."""
        hg = self.hg_conv3d
        seqs = {"conv2": self.conv2[0], "h1": hg.conv1[0], "h2": hg.conv2, "h3": hg.conv3[0], "h4": hg.conv4[0], "h5": hg.conv5}
        c = self.conv2[0][0].out_channels
        if c != 32 or self.conv2[0][0].in_channels != 32 or not _is_head_conv(self.classifier, c):
            return None
        norms = [self.conv1[0][1]] + [sq[1] for sq in seqs.values()]
        for nm in norms:
            if not isinstance(nm, nn.BatchNorm3d) or nm.training or nm.running_mean is None:
                return None
        tensors = [t for sq in list(seqs.values()) + [hg.conv6] for t in (sq[0].weight, sq[1].weight, sq[1].bias, sq[1].running_mean, sq[1].running_var)]
        tensors.append(self.classifier.weight)
        tensors += [self.conv1[0][1].weight, self.conv1[0][1].bias, self.conv1[0][1].running_mean, self.conv1[0][1].running_var]
        key = tuple((t.data_ptr(), t._version) for t in tensors if t is not None) + (device, _GENERATION[0])
        st = self.__dict__.get("_snvc_x3")
        if st is not None and st["key"] == key:
            return st

        def bound(nm):
            g = nm.weight.detach().abs() if nm.weight is not None else torch.ones(1, device=device)
            b = nm.bias.detach().abs() if nm.bias is not None else torch.zeros(1, device=device)
            return float((b + self.X3_SIGMAS * g).max().item())
        b = {k: bound(sq[1]) for k, sq in seqs.items()}
        st = {"key": key, "exp": {"v1": self._x3_exponent(bound(self.conv1[0][1]))}, "layers": {}, "affine": {}}
        st["exp"].update({k: self._x3_exponent(v) for k, v in b.items()})
        geo = {"conv2": (1, False), "h1": (2, False), "h2": (1, False), "h3": (2, False), "h4": (1, False), "h5": (2, True)}
        for k, sq in seqs.items():
            w = sq[0].weight.detach().to(device)
            st["layers"][k] = ops.Conv3dLayerX3(w, 3, geo[k][0], 1, 1, geo[k][1])
            st["affine"][k] = _folded_bn(sq[1], sq[0].__dict__.setdefault("_snvc_plans", {}).setdefault(device, _Plan()))
        # `post` = relu(bn(conv5(o)) + pre) exists only inside conv5's epilogue (below); its exponent comes from a HARD bound, not from
        # the statistical one: |conv5(o)| <= (sum of |w| over a parity class's taps and all input channels) * max|o|, and o / pre cannot
        # exceed what their own (checked) exponents allow.  conv5 then cannot clamp once the layers before it have not, so the overflow
        # flag is final BEHIND hg conv4 and the host's look at it overlaps conv5 + the gather (0.17 ms of GPU work) instead of
        # leaving the GPU empty for the next call's first launches (r5: 0.08 ms per step).  The price: ~3 bits of exponent headroom
        # out of half's 30 -- nothing, for activations of ordinary magnitude.
        w5 = hg.conv5[0].weight.detach().double().abs()                       # [Cin, Cout, 3, 3, 3]
        sets = ([1], [0, 2])
        l1 = None
        for pd in range(2):
            for ph in range(2):
                for pw in range(2):
                    t = w5[:, :, sets[pd]][:, :, :, sets[ph]][:, :, :, :, sets[pw]].sum(dim=(0, 2, 3, 4))
                    l1 = t if l1 is None else torch.maximum(l1, t)
        sc5, bi5 = st["affine"]["h5"]
        hard = float(((sc5.double().abs() * l1.to(sc5.device) * (65504.0 / 2.0 ** st["exp"]["h4"]) + bi5.double().abs()).max()
                      + 65504.0 / 2.0 ** st["exp"]["h2"]).item())
        st["exp"]["post"] = min(self._x3_exponent(hard), st["exp"]["h2"])
        # the folded tail classifier(bn(conv6(post)) + v): a transposed layer to one channel whose per-voxel tap contraction is part
        # of conv5's epilogue (snvc_f16x3_deconv3d_tail_forward); `post` is never stored
        st["tail"] = st["tail_bias"] = None
        if isinstance(hg.conv6[1], nn.BatchNorm3d) and not hg.conv6[1].training and hg.conv6[0].in_channels % 32 == 0:
            wf, fb = folded_head_weights(hg.conv6[0], hg.conv6[1], self.classifier)
            st["tail"] = ops.TailWeightsX3(wf.float().to(device))
            st["tail_bias"] = fb.float().reshape(1).to(device)
        st["guard"] = overflow_guard(self, device)      # survives rebuilds of this state: a pending flag is never dropped
        st["flag"] = st["guard"].flag
        self.__dict__["_snvc_x3"] = st
        return st

    @staticmethod
    def _x3_v1_affine(st, scale, bias):
        """The first layer's folded BatchNorm with the exponent of the split first-layer tensor folded in (exact: a power of two)."""
        e1 = st["exp"]["v1"]
        src = st.get("v1_affine_src")      # the folded tensors themselves are kept: an address cannot come back as another tensor
        if src is None or src[0] is not scale or src[1] is not bias or src[2] != (scale._version, bias._version, e1):
            st["v1_affine"] = ((scale * 2.0 ** e1).contiguous(), (bias * 2.0 ** e1).contiguous())
            st["v1_affine_src"] = (scale, bias, (scale._version, bias._version, e1))
        return st["v1_affine"]

    def _x3_select(self, device, arithmetic=None):
        """The split-mode state if this call runs in split mode, else None."""
        mode = arithmetic or self.arithmetic
        if mode == "fp32" or torch.is_grad_enabled() or self.training or self.__dict__.get("_snvc_x3_off"):
            if mode == "x3":
                raise RuntimeError("arithmetic='x3' needs inference (no autograd, eval mode) and no earlier overflow")
            return None
        st = self._x3_state(device)
        if st is None:
            if mode == "x3":
                raise RuntimeError("arithmetic='x3': the stack does not qualify (32 channels, eval-mode BatchNorm3d everywhere)")
            return None
        if st["guard"].event is not None and self._overflowed(st["guard"], mode, "an earlier call's result clamped it"):
            return None                           # overflow_check = "deferred": the previous call's flag
        return st

    def _overflowed(self, guard, mode, what):
        """Look at a posted flag (synchronous).  True: a value was clamped -- split mode is switched off for this model."""
        if not guard.wait():
            return False
        return self._leave_split_mode(mode, what)

    def _leave_split_mode(self, mode, what):
        self.__dict__["_snvc_x3_off"] = True
        msg = ("snvc_amd: split-mode (f16x3) overflow -- an activation exceeded |beta| + %g |gamma| of its BatchNorm; %s.  "
               "This model now runs on the fp32-MFMA kernels (reset_split_mode() turns split mode back on)." % (self.X3_SIGMAS, what))
        if mode == "x3":
            raise RuntimeError("arithmetic='x3': " + msg)
        warnings.warn(msg)
        return True

    def check_overflow(self) -> bool:
        """With ``overflow_check = "deferred"``: wait for the last split-mode call's flag.  True if that call's result was
        clamped (the model leaves split mode, as in the checked mode); always False in the default checked mode."""
        hit = False
        for guard in self.__dict__.get("_snvc_x3_guard", {}).values():
            hit |= self._overflowed(guard, "auto", "the last result clamped it")
        return hit

    def reset_split_mode(self):
        """Turn split mode back on after an overflow switched it off (e.g. after loading matching statistics)."""
        self.__dict__.pop("_snvc_x3_off", None)

    def _tail_x3(self, st, v1s, timing=None):
        """conv2 (+ side head) -> hourglass -> folded one-channel tail on a split-C8 first-layer result ``v1s`` (exponent
        st['exp']['v1']).  Full-resolution intermediates: v1s, v2s (0.74 GB each at cfg2, like their fp32 counterparts)."""
        L, A, E, flag = st["layers"], st["affine"], st["exp"], st["flag"]
        n, dev = v1s.size(0), v1s.device
        d, h, w = v1s.shape[3:6]
        hg = self.hg_conv3d
        if timing is not None and "conv2" in timing:
            timing["conv2"][0].record()
        v2s, hv = L["conv2"](v1s, E["v1"], *A["conv2"], flags=EPI_RELU, out_exp=E["conv2"], head=self.classifier.weight, overflow=flag,
                             out=self._buffer("v2s", (n, 2, 4, d, h, w, 8), dev, torch.float16))
        if timing is not None and "conv2" in timing:
            timing["conv2"][1].record()
        _ROUTES["side_head"] += 1
        o = L["h1"](v2s, E["conv2"], *A["h1"], flags=EPI_RELU, out_exp=E["h1"], overflow=flag)                  # 1/2 res, 2C channels
        pre = L["h2"](o, E["h1"], *A["h2"], flags=EPI_RELU, out_exp=E["h2"], overflow=flag)                     # relu(bn(conv))   :153-156
        o = L["h3"](pre, E["h2"], *A["h3"], flags=EPI_RELU, out_exp=E["h3"], overflow=flag)                     # 1/4 res
        o = L["h4"](o, E["h3"], *A["h4"], flags=EPI_RELU, out_exp=E["h4"], overflow=flag)
        st["guard"].post()      # the last layer that can clamp (conv5's exponent is a hard bound, see _x3_state): the flag leaves for the host here
        if st["tail"] is not None and self.fused_tail:
            # post = relu(bn(deconv(o)) + pre) is formed and contracted with the folded tail's 27 taps inside conv5's launch (r5): 27
            # fp32 planes per parity class leave instead of 64 channels, and the last launch only sums them
            t = L["h5"].forward_tail(o, E["h4"], *A["h5"], st["tail"], residual=pre, res_exp=E["h2"], flags=EPI_RELU | ops.EPI_ADD_PRE,
                                     out_exp=E["post"], overflow=flag, out=self._buffer("tail_t", (n, 27, 8) + tuple(o.shape[3:6]), dev))
            cost = ops.deconv_tail_gather(t, st["tail_bias"], hv)                             # deconv'(post) + b' + classifier(v2)
            _ROUTES["x3_fused_tail"] += 1
        else:
            # post = relu(bn(deconv(o)) + pre): the result leaves as fp32 NCDHW for the one-channel transposed tail (VALU kernel)
            post = L["h5"](o, E["h4"], *A["h5"], residual=pre, flags=EPI_RELU | ops.EPI_ADD_PRE, out_exp=E["h2"], to_f32=True)
            cost = hg.conv6.fused(post, residual=None, head=self.classifier, head_residual=hv)    # deconv'(post) + b' + classifier(v2)
        _ROUTES["x3_tail"] += 1
        if self.overflow_check == "call" and st["guard"].wait():
            # waited for while the transposed layer and the tail still run: this call's result is dropped and redone in fp32
            raise SplitOverflow()
        return cost

    def _tail(self, v, hv=None):
        """v + hourglass(v)[0] -> classifier.  The hourglass's last transposed layer has no activation
        (reference submodule.py:166), so at inference  classifier(bn(deconv(post)) + v) = deconv'(post) + b' +
        classifier(v)  with deconv' a transposed layer to ONE channel (fused_conv3d): the full-resolution C-channel
        tensor is never formed.  ``hv`` = classifier(v) when conv2's launch already produced it (side head)."""
        cost, _, _ = self.hg_conv3d(v, None, None, residual=v, head=self.classifier, head_residual=hv)
        return cost

    def _gn_qualifies(self, device, arithmetic=None):
        """The overflow guard if this call can run a GroupNorm stack in split mode (see _gn_tail_x3), else None."""
        from .submodule import X3_GROUP_NORM, overflow_guard
        mode = arithmetic or self.arithmetic
        norms = [m for m in self.modules() if isinstance(m, (nn.GroupNorm, nn.modules.batchnorm._BatchNorm))]
        if (not norms or not all(isinstance(m, nn.GroupNorm) for m in norms) or not X3_GROUP_NORM[0] or mode == "fp32"
                or torch.is_grad_enabled() or self.training or self.__dict__.get("_snvc_x3_off") or device.type != "cuda"
                or self.conv1[0][0].out_channels % 32 != 0):
            return None
        guard = overflow_guard(self, device)
        if guard.event is not None and self._overflowed(guard, mode, "an earlier call's result clamped it"):
            return None
        return guard

    def _gn_tail_x3(self, v1, timing=None, arithmetic=None):
        """A GroupNorm stack (``GlobalStack(gn=True)``) behind its fp32 first-layer result, in split mode (r5): nothing folds here
        (every norm needs its own conv result's statistics), so the layers run one by one through ``fused_conv3d_x3``'s GroupNorm
        form -- conv2, the hourglass, conv6 + cost0 -- and the 1x1x1 classifier on the fp32 kernel.  Returns None when the
        call does not qualify.  A flagged call raises SplitOverflow (``_checked`` redoes it in fp32)."""
        from .submodule import SplitT, x3_exponent, x3_norm_bound, _Plan
        if isinstance(v1, SplitT):          # conv1 ran in split mode already (_forward_volume_unchecked): its guard is the caller's
            v1s, guard = v1, self.__dict__["_snvc_x3_guard"][v1.t.device]
        else:
            if not (v1.is_cuda and v1.dtype == torch.float32):
                return None
            guard = self._gn_qualifies(v1.device, arithmetic)
            if guard is None:
                return None
            norm1 = self.conv1[0][1]
            b1 = x3_norm_bound(norm1, self.conv1[0][0].__dict__.setdefault("_snvc_plans_x3", {}).setdefault(v1.device, _Plan()))
            e1 = x3_exponent(b1)
            n, c = v1.size(0), v1.size(1)
            v1s = SplitT(ops.to_split(v1, e1, out=self._buffer("v1s", (n, 2, c // 8) + tuple(v1.shape[2:]) + (8,), v1.device, torch.float16)), e1, b1)
        if timing is not None and "conv2" in timing:
            timing["conv2"][0].record()
        v2 = self.conv2.fused_x3(v1s, flag=guard.flag)
        if timing is not None and "conv2" in timing:
            timing["conv2"][1].record()
        out, _, _ = self.hg_conv3d.forward_x3(v2, residual=v2, flag=guard.flag)
        guard.post()                      # nothing clamps behind this point
        cost = self.classifier(ops.from_split(out.t, out.exp))      # Conv3d(C, 1, k1): the fp32 kernel (the split one-channel form is k3)
        _ROUTES["x3_gn_tail"] += 1
        if self.overflow_check == "call" and guard.wait():
            raise SplitOverflow()
        return cost

    @staticmethod
    def _sheared_geometry4(m0: int, d: int, w: int):
        """sheared_geometry for four phases (index 4 w - d - m0): the kernel's taps reach 5 elements either side, so Rq sits 8 elements
        into its padded grid and the last column's window carries 5 elements of context."""
        off = 8
        wu = (off + 4 * (w - 1) + 1 + 5 + 5 + 3) // 4 * 4
        u_lo = 4 * (w - 1) - (d - 1) - m0 - 5
        return off, wu, 8 - u_lo, (d + 10 + 8 + 3) // 4 * 4

    @staticmethod
    def _sheared_layer4(plans, wr):
        """Four phases: G[u] = sum over (kd, kh, kw) of w[kd, kh, kw] Rq[row + kh - 1][u + 4 kw - kd] is a 3 x 11 kernel with nine non-zero
        columns = three 3 x 3 layers P_kw[u] = sum_{kd, kh} w[kd, kh, kw] Rq[row + kh - 1][u - kd] added 4 elements apart:
        G[u] = P_-1[u - 4] + P_0[u] + P_+1[u + 4], and G' (the last column: no kw = +1 taps) = P_-1[u - 4] + P_0[u].  One depth-1 3 x 3
        layer with 3 (kw) x 3 (depth class) x Cout output channels, folded in fp64 like sheared_kernels."""
        if "sheared4" not in plans:
            from .submodule import SHEAR_CLASS_KDS
            w = wr.detach().double()                                        # [Cout, C, kd, kh, kw]
            cout, c = w.shape[0], w.shape[1]
            k = torch.zeros((3, 3, cout, c, 3, 3), dtype=torch.float64, device=w.device)      # [kw][cls][co][c][kh][1 - kd]
            for cls, kds in enumerate(SHEAR_CLASS_KDS):
                for kd in kds:
                    for kw in (-1, 0, 1):
                        k[kw + 1, cls, :, :, :, 1 - kd] += w[:, :, kd + 1, :, kw + 1]
            plans["sheared4"] = ops.Conv3dLayer(k.reshape(9 * cout, c, 3, 3).float().contiguous(), 3, 1, 1, 1, False, planar=True)
        return plans["sheared4"]

    def _sheared4_inputs(self, plans, wr, rq_src, q_up, m0, d, w):
        """(g, gcol, off, off_col) of the four-phase sheared layer: ``rq_src`` upsampled by ``q_up`` (1 or 2) is Rq."""
        lay = self._sheared_layer4(plans, wr)
        off, wu, off_col, wu_col = self._sheared_geometry4(m0, d, w)
        out = []
        for width, o_, last in ((wu, off, False), (wu_col, off_col, True)):
            p = lay(ops.sheared_upsample(rq_src, q_up, width, o_).unsqueeze(2)).squeeze(2)      # [N, 9 Cout, H, width]
            n, c9, h, _ = p.shape
            p = p.view(n, 3, c9 // 3, h, width)
            g = p[:, 1].clone()
            g[..., 4:] += p[:, 0][..., :-4]
            if not last:
                g[..., :-4] += p[:, 2][..., 4:]
            out.append(g.contiguous())
        return out[0], out[1], off, off_col

    def _ds_sheared_first_layer(self, left, right, shift, ds, arithmetic=None):
        """``downsample`` = ds > 1 (features at ds times the volume's resolution, BuildCostVolume_cuda.cu:224-225: out[d,h,w] samples the
        right feature on row ds*h at x = ds*w - shift[d]) on uniformly spaced planes shift[d] = (m0 + d) / q is the sheared layer
        with q*ds phases (r6): index q*ds*w - m0 - d of the row-subsampled right feature upsampled by q alone; the
        left feature is subsampled in both directions.  Two phases (ds = 2, whole-pixel planes -- the sweep that covers cfg2's range
        at twice the resolution): the existing 3 x 7 layers with the feature itself in the upsampled image's place.  Four phases
        (ds = 2 with half-pixel planes, ds = 4 with whole-pixel planes): three 3 x 3 layers added 4 elements apart
        (_sheared_layer4) and the expand kernels' q = 4 instantiation.  Anything else returns None -> the materialised route."""
        conv, bn = self.conv1[0][0], self.conv1[0][1]
        nonneg, structure = self._shift_structure(shift)
        assert nonneg                              # reference __init__.py:12
        if structure is None or structure[0] * ds not in (2, 4):
            return None
        q_up, m0 = structure
        phases = q_up * ds
        n, c = left.size(0), left.size(1)
        h, w, d = left.size(2) // ds, left.size(3) // ds, shift.size(1)
        if w % 8 or w > 512:
            return None
        if phases == 2:
            if not self._sheared_fits(2, m0, d, w, False):
                return None
        else:
            _, wu4, _, _ = self._sheared_geometry4(m0, d, w)
            if 4 * ((wu4 + 3) // 4 * 4 + 16 + d) > 150 * 1024:
                return None
        st = self._x3_select(left.device, arithmetic)
        left_s = left[:, :, ::ds, ::ds].contiguous()
        right_r = right[:, :, ::ds, :].contiguous()          # rows ds*h; every input column
        wt = conv.weight
        plans = conv.__dict__.setdefault("_snvc_factored", {})
        key = (wt.data_ptr(), wt._version, wt.device, _GENERATION[0])
        if plans.get("key") != key:
            plans.clear()
            plans.update(key=key, right=ops.Conv3dLayer(wt.detach()[:, c:].contiguous(), 3, 1, 1, 1, False), plan=_Plan())
        scale, bias = _folded_bn(bn, plans["plan"])
        planes = self._left_planes_layer(plans, wt.detach()[:, :c])(left_s.unsqueeze(2)).view(n, c, 3, h, w)
        if phases == 2:
            lay_g, lay_col = self._sheared_layers(plans, wt.detach()[:, c:], 2)
            off, wu, off_col, wu_col = sheared_geometry(2, m0, d, w)
            g = lay_g(ops.sheared_upsample(right_r, 1, wu, off).unsqueeze(2)).squeeze(2)
            gcol = lay_col(ops.sheared_upsample(right_r, 1, wu_col, off_col).unsqueeze(2)).squeeze(2)
        else:
            g, gcol, off, off_col = self._sheared4_inputs(plans, wt.detach()[:, c:], right_r, q_up, m0, d, w)
        shape = (n, c, d, h, w)
        try:
            if st is not None:
                v1s = self._buffer("v1s", (n, 2, c // 8, d, h, w, 8), left.device, torch.float16)
                ops.sheared_expand_split(g, gcol, planes, *self._x3_v1_affine(st, scale, bias), v1s, phases, m0, off, off_col, ops.EPI_RELU,
                                         st["flag"])
                _ROUTES["ds_sheared_first_conv"] += 1
                self.__dict__["_snvc_last_v1"] = "v1s"
                return self._tail_x3(st, v1s, None)
            v = self._buffer("v1", shape, left.device)
            ops.sheared_expand(g, gcol, planes, scale, bias, v, phases, m0, off, off_col, ops.EPI_RELU)
        except ops.Unsupported:
            return None
        _ROUTES["ds_sheared_first_conv"] += 1
        return self._conv2_tail(v, shape, None, arithmetic)

    def _gn_sheared_first_layer(self, left, right, shift, arithmetic=None):
        """``GlobalStack(gn=True)`` on uniformly spaced disparity planes WITHOUT the 1.47 GB volume (r6): with one
        channel per group -- GroupNorm(32, 32), what convbn_3d(..., gn=True) builds for 32 channels, reference submodule.py:41-49 --
        the first layer's statistics are per (sample, channel) over the volume, which is what the sheared layer's statistics pass
        computes for a train-mode BatchNorm at batch 1 (snvc_sheared_expand_stats: the raw result is never stored).  Per sample:
        statistics, then the expand pass applies scale / shift + ReLU and writes the split pair the split-mode GroupNorm tail reads.
        Returns None when the call does not qualify (another spacing, rows the sheared kernels do not cover, split mode off)."""
        from .submodule import SplitT, x3_exponent, x3_norm_bound
        conv, norm = self.conv1[0][0], self.conv1[0][1]
        guard = self._gn_qualifies(left.device, arithmetic)
        if guard is None:
            return None
        nonneg, structure = self._shift_structure(shift)
        assert nonneg                              # reference __init__.py:12
        n, c, h, w = left.shape
        d = shift.size(1)
        if structure is None or not self._sheared_fits(structure[0], structure[1], d, w, True):
            return None
        q, m0 = structure
        wt = conv.weight
        plans = conv.__dict__.setdefault("_snvc_factored", {})
        key = (wt.data_ptr(), wt._version, wt.device, _GENERATION[0])
        if plans.get("key") != key:
            plans.clear()
            plans.update(key=key, right=ops.Conv3dLayer(wt.detach()[:, c:].contiguous(), 3, 1, 1, 1, False), plan=_Plan())
        planes = self._left_planes_layer(plans, wt.detach()[:, :c])(left.unsqueeze(2)).view(n, c, 3, h, w)
        lay_g, lay_col = self._sheared_layers(plans, wt.detach()[:, c:], q)
        off, wu, off_col, wu_col = sheared_geometry(q, m0, d, w)
        g = lay_g(ops.sheared_upsample(right, q, wu, off).unsqueeze(2)).squeeze(2)
        gcol = lay_col(ops.sheared_upsample(right, q, wu_col, off_col).unsqueeze(2)).squeeze(2)
        b1 = x3_norm_bound(norm, conv.__dict__.setdefault("_snvc_plans_x3", {}).setdefault(left.device, _Plan()))
        e1 = x3_exponent(b1)
        v1s = self._buffer("v1s", (n, 2, c // 8, d, h, w, 8), left.device, torch.float16)
        gam = norm.weight.detach() if norm.weight is not None else None
        bet = norm.bias.detach() if norm.bias is not None else None
        try:
            for i in range(n):                     # GroupNorm statistics are per sample
                sc, sh, _, _ = ops.sheared_expand_stats(g[i:i + 1], gcol[i:i + 1], planes[i:i + 1], gam, bet, (1, c, d, h, w), q, m0, off,
                                                        off_col, norm.eps)
                ops.sheared_expand_split(g[i:i + 1], gcol[i:i + 1], planes[i:i + 1], (sc * 2.0 ** e1).reshape(-1).contiguous(),
                                         (sh * 2.0 ** e1).reshape(-1).contiguous(), v1s[i:i + 1], q, m0, off, off_col, ops.EPI_RELU, guard.flag)
        except ops.Unsupported:
            return None
        _ROUTES["gn_sheared_first_conv"] += 1
        self.__dict__["_snvc_last_v1"] = "v1s"
        return self._gn_tail_x3(SplitT(v1s, e1, b1), None, arithmetic)

    def _conv2_tail(self, v1, shape, timing=None, arithmetic=None):
        gn_cost = self._gn_tail_x3(v1, timing, arithmetic)
        if gn_cost is not None:
            self.__dict__["_snvc_last_v1"] = "v1"
            return gn_cost
        st = self._x3_select(v1.device, arithmetic) if (v1.is_cuda and v1.dtype == torch.float32) else None
        if st is not None:      # split mode from an fp32 first-layer result: one layout pass (the sheared path writes the pair itself)
            n, c = v1.size(0), v1.size(1)
            v1s = ops.to_split(v1, st["exp"]["v1"], out=self._buffer("v1s", (n, 2, c // 8) + tuple(v1.shape[2:]) + (8,), v1.device, torch.float16))
            self.__dict__["_snvc_last_v1"] = "v1"       # the fp32 first-layer result is in the workspace too
            return self._tail_x3(st, v1s, timing)
        self.__dict__["_snvc_last_v1"] = "v1"
        if timing is not None and "conv2" in timing:
            timing["conv2"][0].record()
        v, hv = self.conv2.fused(v1, out=self._buffer("v2", shape, v1.device), side_head=self.classifier)
        if timing is not None and "conv2" in timing:
            timing["conv2"][1].record()
        return self._tail(v, hv)

    def forward(self, volume):
        from ..lazy import LazyCostVolume
        if isinstance(volume, LazyCostVolume):
            # build_cost_volume's result that nobody has looked at yet: the reference's call sequence on the fused path
            # ... or that somebody only PEEKED at (materialised, never written to: same values, so the fused path still applies)
            if volume.is_pristine and not torch.is_grad_enabled() and not self.training:
                left, right, shift, ds = volume.sources
                if left.shape[1] * 2 == self.conv1[0][0].in_channels and left.shape[3] % 4 == 0:
                    from .. import lazy as _lazy
                    _lazy.CONSUMER.ref = weakref.ref(self)      # the next build_cost_volume starts this model's first-layer prep
                    pre = volume.take_prefetch(self)
                    if pre is not None:                         # build_cost_volume already ran the step up to its host sync
                        try:
                            try:
                                while True:
                                    next(pre)
                            except StopIteration as done:
                                _ROUTES["lazy_prefetch_resumed"] += 1
                                return done.value
                        except SplitOverflow:
                            self._leave_split_mode(self.arithmetic, "this call was redone in fp32")
                            _ROUTES["x3_overflow_redo"] += 1
                            return self._forward_pair_unchecked(left, right, shift, ds, shift_checked=True, spacing=volume.spacing,
                                                                arithmetic="fp32")
                    return self.forward_pair(left, right, shift, ds, shift_checked=True, spacing=volume.spacing)
            # built from these two features: its maximum is theirs (interpolation weights are in [0, 1]), if they are untouched
            # ... and the volume itself: one that was written to in place (vol.mul_(8)) no longer has their maximum
            scale_from = tuple(volume.sources[:2]) if volume.is_pristine else None
            # the 1.47 GB volume is built and conv1 runs over all 64 channels (about half the pairs/s of the fused path): said once
            # per model, with the reason, and counted ("documented, not reported per call")
            _ROUTES["lazy_volume_materialized"] += 1
            if not self.__dict__.get("_snvc_lazy_warned"):
                self.__dict__["_snvc_lazy_warned"] = True
                why = ("autograd is enabled" if torch.is_grad_enabled() else "the model is in training mode" if self.training else
                       "the volume or its sources were written to after build_cost_volume" if not volume.is_pristine else
                       "the feature shape is outside the fused first layer's (2C input channels, W % 4 == 0)")
                warnings.warn("snvc_amd: model(build_cost_volume(...)) is taking the MATERIALISED route (the whole concat volume is built): "
                              + why + ".  forward_pair(left, right, shift) is the fused entry point.")
            return self._forward_volume(volume.materialize(), scale_from=scale_from)
        return self._forward_volume(volume)

    def lazy_prefetch(self, left, right, shift):
        """Called by ``build_cost_volume`` (no_grad, fp32, downsample 1) for the model that consumed the previous lazy volume: runs
        this call's step up to and including its one host sync -- the sign / spacing check of ``shift`` (reference __init__.py:12:
        an AssertionError for a negative shift comes out of build_cost_volume exactly as before) -- with the left half's planes
        and the speculative first-layer prep queued in front of the wait, and returns the paused generator (None if this model
        would not take the fused path for these inputs)."""
        conv, bn = self.conv1[0][0], self.conv1[0][1]
        if (self.training or torch.is_grad_enabled() or not isinstance(bn, nn.BatchNorm3d) or bn.training or left.device != conv.weight.device
                or left.shape[1] * 2 != conv.in_channels or left.shape[3] % 4 != 0 or shift.size(1) < 4 or shift.dtype != torch.float32
                or left.size(0) == 0):
            return None
        gen = self._forward_pair_steps(left, right, shift, 1, pause=True)
        if next(gen, None) is None:        # the path taken has no pause point (it ran to its end): nothing to resume
            return None
        return gen

    def _checked(self, fn, arithmetic, *args, **kw):
        """Run a split-mode capable entry point; a call whose overflow flag came back set is redone on the fp32-MFMA kernels
        (its clamped result never leaves).  One extra step, once per model: split mode stays off afterwards."""
        try:
            return fn(*args, arithmetic=arithmetic, **kw)
        except SplitOverflow:
            self._leave_split_mode(arithmetic or self.arithmetic, "this call was redone in fp32")
            _ROUTES["x3_overflow_redo"] += 1
            return fn(*args, arithmetic="fp32", **kw)

    def _forward_volume(self, volume, timing=None, arithmetic=None, scale_from=None):
        return self._checked(self._forward_volume_unchecked, arithmetic, volume, timing=timing, scale_from=scale_from)

    def _forward_volume_unchecked(self, volume, timing=None, arithmetic=None, scale_from=None):
        """conv1 over a materialised [N, 2C, D, H, W] volume, then the tail.  Split mode (r4): the volume is scaled by a power of
        two derived on the device from its own maximum (or from the features it was built from, ``scale_from``: the volume holds
        nothing but their values and interpolations) and split once, conv1 runs on the split-mode kernel and writes the pair
        conv2 reads."""
        if torch.is_grad_enabled() or not volume.is_cuda:
            return self._tail(self.conv2(self.conv1(volume)))
        n, c2 = volume.size(0), volume.size(1)
        shape = (n, c2 // 2) + tuple(volume.shape[2:])
        gguard = self._gn_qualifies(volume.device, arithmetic) if (volume.dtype == torch.float32 and c2 % 16 == 0) else None
        if gguard is not None:      # GroupNorm stack (r5): the volume split once, conv1 and everything behind it in split mode
            from .submodule import SplitT
            mul = ops.split_scale_for(*(scale_from if scale_from is not None else (volume,)))
            vs = ops.to_split(volume, mul_dev=mul, out=self._buffer("vol_s", (n, 2, c2 // 8) + tuple(volume.shape[2:]) + (8,), volume.device,
                                                                      torch.float16))
            if timing is not None and "conv1" in timing:
                timing["conv1"][0].record()
            v1s = self.conv1.fused_x3(SplitT(vs, 0, None, mul), flag=gguard.flag)
            if timing is not None and "conv1" in timing:
                timing["conv1"][1].record()
            return self._gn_tail_x3(v1s, timing, arithmetic)
        st = self._x3_select(volume.device, arithmetic) if (volume.dtype == torch.float32 and c2 % 16 == 0) else None
        if st is not None:
            from .submodule import SplitT
            mul = ops.split_scale_for(*(scale_from if scale_from is not None else (volume,)))
            vs = ops.to_split(volume, mul_dev=mul, out=self._buffer("vol_s", (n, 2, c2 // 8) + tuple(volume.shape[2:]) + (8,), volume.device,
                                                                      torch.float16))
            v1s = self._buffer("v1s", (n, 2, c2 // 16) + tuple(volume.shape[2:]) + (8,), volume.device, torch.float16)
            if timing is not None and "conv1" in timing:
                timing["conv1"][0].record()
            self.conv1.fused_x3(SplitT(vs, 0, None, mul), out=v1s, out_exp=st["exp"]["v1"], flag=st["flag"])
            if timing is not None and "conv1" in timing:
                timing["conv1"][1].record()
            self.__dict__["_snvc_last_v1"] = "v1s"
            return self._tail_x3(st, v1s, timing)
        if timing is not None and "conv1" in timing:
            timing["conv1"][0].record()
        v = self.conv1.fused(volume, out=self._buffer("v1", shape, volume.device))
        if timing is not None and "conv1" in timing:
            timing["conv1"][1].record()
        return self._conv2_tail(v, shape, timing, arithmetic)

    @staticmethod
    def _shift_structure(shift):
        """One device -> host round trip (it replaces the `assert torch.all(shift >= 0)` sync of the reference's wrapper):
        (all shifts >= 0, (q, m0) or None).  (q, m0): every row of ``shift`` is (m0 + d) / q for d = 0..D-1 with q in {1, 2}
        -- uniformly spaced whole- or half-pixel disparity planes -- exactly, in fp32."""
        return GlobalStack._shift_structure_end(ops.shift_structure_begin(shift.detach()), shift.size(1))

    @staticmethod
    def _shift_structure_end(ticket, d):
        return ops.shift_spacing_result(ticket, d)     # one launch (r3 first form: 12 torch kernels, 60 us)

    @staticmethod
    def _sheared_fits(q, m0, d, w, training):
        """The limits of csrc/sheared_conv.hip's launchers (they return SNVC_ERR_UNSUPPORTED beyond them), mirrored so that
        such shapes take the general path instead of aborting the step: rows of at most 512 float4 pieces, and per workgroup
        one row block's window of G plus D plane offsets in <= 150 KB of LDS (inference: at least one row; the training
        kernels' own formulas: snvc_sheared_expand_stats like the forward, snvc_sheared_backward_reduce 4 rows of G, G',
        their sums and D)."""
        if w % 4 or w // 4 > 512:
            return False
        _, wu, _, wu_col = sheared_geometry(q, m0, d, w)
        lw = (wu + q - 1) // q + 4
        if 4 * (q * lw + d) > 150 * 1024:
            return False
        if training and 4 * 4 * (q * lw + d + 2 * wu + 2 * wu_col) > 150 * 1024:
            return False
        return True

    def _sheared_layers(self, plans, wr, q):
        """The depth-1 3 x 7 layers that compute G and G' (csrc/sheared_conv.hip): K[kh][t] = sum over (kd, kw) with
        q*kw - kd = t of the right-half weights, folded in fp64; G' is the kernel without its kw = +1 taps."""
        cache = plans.setdefault("sheared", {})
        if q not in cache:
            k = sheared_kernels(wr, q)                                     # [all | last column][depth class][Cout,C,3,7]
            cout, c = k.shape[2], k.shape[3]
            cache[q] = tuple(ops.Conv3dLayer(k[i].reshape(3 * cout, c, 3, 7).float().contiguous(), 7, 1, 3, 1, False, planar=True,
                                             ksize_h=3) for i in range(2))
        return cache[q]

    def _sheared_layers_x3(self, plans, wr, q):
        """``_sheared_layers`` as split-mode depth-1 layers (ops.Conv2dLayerX3): same folded 3 x 7 kernels."""
        cache = plans.setdefault("sheared_x3", {})
        if q not in cache:
            k = sheared_kernels(wr, q)                                     # [all | last column][depth class][Cout,C,3,7]
            cout, c = k.shape[2], k.shape[3]
            cache[q] = tuple(ops.Conv2dLayerX3(k[i].reshape(3 * cout, c, 3, 7).float().contiguous()) for i in range(2))
        return cache[q]

    @staticmethod
    def _left_planes_weight(wl):
        """[3 * Cout, C, 3, 3]: channel co*3 + cls = the kd taps depth class cls sees, summed in fp64."""
        w = wl.detach().double()                                           # [Cout, C, kd, kh, kw]
        k = torch.stack([w[:, :, 1:].sum(2), w.sum(2), w[:, :, :2].sum(2)], dim=1)          # [Cout, 3, C, kh, kw]
        return k.reshape(-1, w.shape[1], 3, 3).float().contiguous()

    @staticmethod
    def _left_planes_layer(plans, wl):
        """The three depth-class planes of the LEFT half of conv1(volume) as ONE depth-1 3x3 convolution of the left feature
        with 3*Cout output channels (channel co*3 + cls = the kd taps that class sees, summed in fp64): [N,3C,1,H,W] is
        the [N,C,3,H,W] tensor snvc_conv3d_forward_ex takes.  (r1-r2 ran the 3D kernel on the feature stacked 3 deep: a
        one-tile-deep launch whose time is one workgroup's whole channel loop, 50 us.)"""
        if "left2d" not in plans:
            plans["left2d"] = ops.Conv3dLayer(GlobalStack._left_planes_weight(wl), 3, 1, 1, 1, False, planar=True)
        return plans["left2d"]

    @staticmethod
    def _commuted_weights(wr):
        """(P, Q, E) weights of the warp-after-convolution first layer as [*, C, 3, 3] depth-1 kernels (see _commuted_layers)."""
        w = wr.detach()                                                   # [Cout, C, kd, kh, kw]
        cout, c = w.shape[0], w.shape[1]
        wk = w.permute(2, 0, 1, 3, 4).contiguous()                        # [kd, Cout, C, kh, kw]
        kq = torch.zeros_like(wk)
        kq[..., 1] = wk[..., 2]
        ke = torch.zeros((3, 3, cout, c, 3, 3), dtype=w.dtype, device=w.device)
        for kw in range(3):
            ke[:, kw, :, :, :, 1] = wk[..., kw]
        return tuple(t.reshape(-1, c, 3, 3).contiguous() for t in (wk, kq, ke))

    @staticmethod
    def _commuted_layers_x3(plans, wr):
        if "commuted_x3" not in plans:
            plans["commuted_x3"] = tuple(ops.Conv2dLayerX3(t.float()) for t in GlobalStack._commuted_weights(wr))
        return plans["commuted_x3"]

    @staticmethod
    def _commuted_layers(plans, wr):
        """The depth-1 layers of the warp-after-convolution form of the first layer (csrc/sheared_conv.hip, any shift array):
        P_kd = conv2d(right, W[:, :, kd]); Q_kd = the kw = +1 taps alone (on the centre column); E[kd][kw] = the (c, kh)
        contraction with the kw taps, applied to the image's first column."""
        if "commuted" not in plans:
            w = wr.detach()                                                   # [Cout, C, kd, kh, kw]
            cout, c = w.shape[0], w.shape[1]
            wk = w.permute(2, 0, 1, 3, 4).contiguous()                        # [kd, Cout, C, kh, kw]
            kq = torch.zeros_like(wk)
            kq[..., 1] = wk[..., 2]
            ke = torch.zeros((3, 3, cout, c, 3, 3), dtype=w.dtype, device=w.device)
            for kw in range(3):
                ke[:, kw, :, :, :, 1] = wk[..., kw]
            mk = lambda t: ops.Conv3dLayer(t.reshape(-1, c, 3, 3).contiguous(), 3, 1, 1, 1, False, planar=True)   # noqa: E731
            plans["commuted"] = (mk(wk), mk(kq), mk(ke))
        return plans["commuted"]

    def forward_pair(self, left, right, shift, downsample=1, factored=True, timing=None, shift_checked=False, sheared=True,
                     fused_bn=True, spacing="unknown", commuted=True, arithmetic=None):
        return self._checked(self._forward_pair_unchecked, arithmetic, left, right, shift, downsample, factored, timing, shift_checked,
                             sheared, fused_bn, spacing, commuted)
    forward_pair.__doc__ = "see _forward_pair_unchecked"

    def _forward_pair_unchecked(self, *args, **kw):
        """``_forward_pair_steps`` run to its end."""
        gen = self._forward_pair_steps(*args, **kw)
        try:
            while True:
                next(gen)
        except StopIteration as done:
            return done.value

    def _forward_pair_steps(self, left, right, shift, downsample=1, factored=True, timing=None, shift_checked=False, sheared=True,
                            fused_bn=True, spacing="unknown", commuted=True, arithmetic=None, pause=False):
        """A GENERATOR: with ``pause`` it yields once, right after the step's host sync (the shift array's sign / spacing check) with
        the left half's planes and the speculative first-layer prep already queued -- ``build_cost_volume`` runs it up to there for
        the model that consumed the last lazy volume (the reference's `assert` fires at build time as before, and the GPU is
        not left idle while the host waits), ``GlobalStack.forward`` resumes it.  Without ``pause`` it runs straight through
        (``_forward_pair_unchecked``).  The generator's return value is the cost tensor.

        cost-volume build + 3D CNN forward: the unit BASELINE.json's metric counts.

        ``factored=True`` (inference, eval BatchNorm, downsample 1) uses the structure of the CONCAT
        volume: its left half repeats the left feature on every disparity plane
        (BuildCostVolume_cuda.cu:86), so the left half of conv1(volume) does not depend on d except
        at the two zero-padded ends.  It is computed once as three depth-class planes (this same
        conv kernel on the left feature stacked 3 deep with conv1's left-half weights) and added in
        the epilogue of the 3D convolution over the RIGHT half only -- which is also the only half
        of the volume that has to be built.  Same result (fp32 summation order aside), half of
        conv1's work and of the volume's HBM traffic.  ``factored=False`` materialises the full volume
        through ``build_cost_volume`` exactly like the reference would.
        ``sheared=True`` (with ``factored``): when the disparity planes are uniformly spaced by a whole or half pixel --
        ``shift[n, d] = (m0 + d) / q``, q in {1, 2}, what a plane sweep over disparities is (BASELINE.json configs[1]:
        linspace(0, 95.5, 192)) -- the warped half is a shear of ONE 2D image and conv1 over it is a 2D convolution
        evaluated along the shear (csrc/sheared_conv.hip): the warped volume is not built at all and conv1's 318 GFLOP
        become 3.4.  Any other shift array (checked on the device, same sync as the reference's assert) takes the
        general factored path.
        ``arithmetic``: None (the model's ``arithmetic`` attribute, default "auto"), "fp32" or "x3": whether conv2 and the
        hourglass run on the split-mode kernels (see ``_x3_state``).
        ``timing``: optional dict ``{"volume": (start, end), "conv1": (start, end)}`` of events recorded on the
        current stream around the cost-volume launch and the first 3D convolution (the dominant kernel); used by
        bench.py for the roofline figures."""
        def mark(name, which):
            if timing is not None and name in timing:
                timing[name][which].record()

        conv, bn = self.conv1[0][0], self.conv1[0][1]
        training_graph = torch.is_grad_enabled() and (left.requires_grad or right.requires_grad or conv.weight.requires_grad)
        if (training_graph and factored and timing is None and downsample == 1 and left.dtype == torch.float32
                and left.size(3) % 4 == 0 and (left.size(2) * left.size(3)) % 4 == 0 and shift.size(1) >= 2):
            # training (cfg4): the same factoring with a backward pass (half the first layer's dgrad / wgrad, no 1.5 GB volume)
            structure = None
            if sheared and left.size(0) > 0 and shift.size(1) >= 4 and shift.dtype == torch.float32 and left.size(1) <= 32:
                nonneg, structure = self._shift_structure(shift)
                assert nonneg
            else:
                assert torch.all(shift >= 0.)
            plan = conv.__dict__.setdefault("_snvc_plans", {}).setdefault(left.device, _Plan())
            if structure is not None and not self._sheared_fits(structure[0], structure[1], shift.size(1), left.size(3), True):
                structure = None           # rows the sheared kernels do not cover (csrc/sheared_conv.hip's LDS limits): general path
            if (structure is not None and isinstance(bn, nn.BatchNorm3d) and bn.training and left.size(3) % 8 == 0
                    and left.size(3) <= 512 and fused_bn):
                # ... with train-mode BatchNorm + ReLU folded around it: neither the raw result nor its gradient is ever stored
                v = _ShearedFirstConvBNFn.apply(left, right, conv.weight, bn.weight, bn.bias, conv, bn, plan, structure[0], structure[1],
                                                shift.size(1))
                return self._tail(self.conv2(v))
            if structure is not None:      # uniformly spaced planes: the sheared 2D form, forward and backward
                v = _ShearedFirstConvFn.apply(left, right, conv.weight, bn.weight, bn.bias, conv, bn, EPI_RELU, plan, structure[0],
                                              structure[1], shift.size(1))
                return self._tail(self.conv2(v))
            v = _FactoredFirstConvFn.apply(left, right, shift, conv.weight, bn.weight, bn.bias, conv, bn, EPI_RELU, plan, commuted)
            return self._tail(self.conv2(v))
        usable = (factored and downsample == 1 and not torch.is_grad_enabled() and isinstance(bn, nn.BatchNorm3d)
                  and not bn.training and left.dtype == torch.float32 and left.size(3) % 4 == 0 and shift.size(1) >= 2)
        if (downsample in (2, 4) and factored and sheared and timing is None and not torch.is_grad_enabled() and isinstance(bn, nn.BatchNorm3d)
                and not bn.training and left.is_cuda and left.dtype == torch.float32 and left.size(2) % downsample == 0
                and left.size(3) % downsample == 0 and shift.size(1) >= 4 and shift.dtype == torch.float32 and left.size(0) > 0
                and left.size(1) % 8 == 0 and left.size(1) * 2 == conv.in_channels and left.shape == right.shape):
            cost = self._ds_sheared_first_layer(left, right, shift, downsample, arithmetic)      # r6; None: the materialised route below
            if cost is not None:
                return cost
        if (not usable and factored and sheared and downsample == 1 and timing is None and not torch.is_grad_enabled() and not self.training
                and isinstance(bn, nn.GroupNorm) and bn.num_groups == conv.out_channels and left.is_cuda and left.dtype == torch.float32
                and left.size(3) % 8 == 0 and left.size(3) <= 512 and shift.size(1) >= 4 and shift.dtype == torch.float32
                and left.size(0) > 0 and left.size(1) % 8 == 0 and left.size(1) * 2 == conv.in_channels):
            cost = self._gn_sheared_first_layer(left, right, shift, arithmetic)       # r6; None: the paths below
            if cost is not None:
                return cost
        if not usable:
            if timing is None:
                vol = _BuildCostVolume.apply(left, right, shift, downsample)      # the eager volume
                if torch.is_grad_enabled() or not vol.is_cuda:
                    return self.forward(vol)
                return self._forward_volume(vol, None, arithmetic, scale_from=(left, right) if downsample == 1 else None)
            assert torch.all(shift >= 0.)        # the wrapper's own check (a sync) stays outside the event pair
            mark("volume", 0)
            vol = ops.cost_volume_forward(left, right, shift, downsample)
            mark("volume", 1)
            if torch.is_grad_enabled():
                mark("conv1", 0)
                v = self.conv1(vol)
                mark("conv1", 1)
                del vol
                return self._tail(self.conv2(v))
            return self._forward_volume(vol, timing, arithmetic, scale_from=(left, right))
        # The one device -> host sync of the step (the wrapper's `assert shift >= 0`, here also the shift array's spacing) is
        # STARTED first and awaited only after everything that does not need its answer has been queued: the left half's
        # planes, and -- speculatively, for the spacing the previous call saw -- the sheared layer's two small 2D convolutions.
        # The host then waits while the GPU still has work, and queues the big layers while those run (r3: the wait sat in
        # front of six small launches whose launch latency the GPU then had to sit through, ~0.1 ms of a 4 ms step).
        ticket = None
        known = shift_checked and spacing != "unknown"       # build_cost_volume already looked (LazyCostVolume.spacing)
        if known and not sheared:
            spacing = None
        if not known and left.size(0) > 0 and shift.size(1) >= 4 and shift.dtype == torch.float32:
            # (r5: also with sheared=False -- the sign check rides on the same launch instead of a blocking torch.all in front of
            # an empty queue: 93 us of idle GPU per step on the any-shift path)
            ticket = ops.shift_structure_begin(shift.detach())
        elif not shift_checked:                  # a LazyCostVolume was checked when build_cost_volume made it
            assert torch.all(shift >= 0.)
        c = left.size(1)
        w = conv.weight
        plans = conv.__dict__.setdefault("_snvc_factored", {})
        key = (w.data_ptr(), w._version, w.device, _GENERATION[0])
        if plans.get("key") != key:
            wr = w.detach()[:, c:].contiguous()
            plans.clear()
            plans.update(key=key, right=ops.Conv3dLayer(wr, 3, 1, 1, 1, False), plan=_Plan())
        scale, bias = _folded_bn(bn, plans["plan"])
        # the first layer's small 2D convolutions run in split mode whenever the stack behind them does (same arithmetic contract:
        # fp32 accuracy on the half pipe); `arithmetic="fp32"` keeps every layer on the fp32-MFMA kernels
        split_prep = bool(self.split_prep and c % 8 == 0 and (3 * conv.out_channels) % 32 == 0 and right.is_contiguous()
                          and (not left.is_contiguous() or left.data_ptr() % 16 == 0) and right.data_ptr() % 16 == 0      # the scale launch reads float4
                          and self._x3_select(left.device, arithmetic) is not None)
        prep_ws = self.__dict__.setdefault("_snvc_prep_ws", {})
        # r6: the prep results (planes, G / G', P / Q / E) live in per-model buffers that the NEXT call through here overwrites.  Every
        # call takes a new epoch; a step that was paused (build_cost_volume's speculative start) compares it when it is resumed
        # (a process-wide counter: invalidate_plans(model) drops the attribute, and a fresh count must not meet an old step's number)
        epoch = self.__dict__["_snvc_prep_epoch"] = next(_PREP_EPOCH)
        if split_prep:      # one host call: the left feature's scale, its split pair, the 3x3 layer with 3 * Cout output channels
            lx = plans.get("left2d_x3")
            if lx is None:
                lx = plans["left2d_x3"] = ops.Conv2dLayerX3(self._left_planes_weight(w.detach()[:, :c]))
            planes = ops.conv2d_x3_from_f32(left, [lx], prep_ws.setdefault("planes", {}))[0]
        else:
            planes = self._left_planes_layer(plans, w.detach()[:, :c])(left.unsqueeze(2))   # [N,3C,1,H,W] ...
        planes = planes.view(left.size(0), c, 3, left.size(2), left.size(3))                 # ... = [N,C,3,H,W]: first / interior / last
        shape = (left.size(0), c, shift.size(1)) + tuple(left.shape[2:])

        def sheared_inputs(q, m0):
            off, wu, off_col, wu_col = sheared_geometry(q, m0, shift.size(1), left.size(3))
            if split_prep:
                # r5: G and G' (depth-1 3 x 7 layers, 7.8 + 2.5 GFLOP: 84 + 31 us on the fp32 matrix pipe) in split mode like the layers behind
                # them: Rq is written as a split pair scaled by the right feature's own maximum (one launch, no host round trip)
                lx_g, lx_col = self._sheared_layers_x3(plans, w.detach()[:, c:], q)
                # one host call for the five launches (behind the step's host sync the GPU is empty: five Python-level launches of
                # 5-20 us kernels starve it); G / G' live in a per-model workspace until this call's expand pass has read them
                g, gcol = ops.sheared_prep_x3(right, q, wu, off, wu_col, off_col, lx_g, lx_col, prep_ws.setdefault("sheared", {}))
                _ROUTES["sheared_prep_x3"] += 1
                return g, gcol, off, off_col
            lay_g, lay_col = self._sheared_layers(plans, w.detach()[:, c:], q)
            if not self.prep_streams:
                g = lay_g(ops.sheared_upsample(right, q, wu, off).unsqueeze(2)).squeeze(2)             # [N,3C,H,WU]
                gcol = lay_col(ops.sheared_upsample(right, q, wu_col, off_col).unsqueeze(2)).squeeze(2)   # [N,3C,H,WU2]
                return g, gcol, off, off_col
            # r5: G and G' are two independent chains of small launches (an upsample + a depth-1 convolution of ~100 workgroups each:
            # their time is one workgroup's latency, not throughput) beside the left half's planes already queued on this stream.
            # They run on two side streams and join before the expand pass: three chains side by side instead of end to end.
            cur = torch.cuda.current_stream(left.device)
            s1, s2 = self._side_streams(left.device)
            start = torch.cuda.Event()
            start.record(cur)            # `right` (and everything else queued so far) is ready behind this point
            out = []
            for st_, lay, wu_, off_ in ((s1, lay_g, wu, off), (s2, lay_col, wu_col, off_col)):
                with torch.cuda.stream(st_):
                    st_.wait_event(start)
                    t = lay(ops.sheared_upsample(right, q, wu_, off_).unsqueeze(2)).squeeze(2)
                    t.record_stream(cur)     # allocated in the side stream's pool, consumed by the expand pass on `cur`
                    done = torch.cuda.Event()
                    done.record(st_)
                out.append((t, done))
            for _, done in out:
                cur.wait_event(done)
            return out[0][0], out[1][0], off, off_col

        def commuted_inputs():
            # any other shift array: three 2D convolutions of the right feature (P, Q) and of its first columns (E)
            if split_prep and (9 * conv.out_channels) % 32 == 0:
                # the three depth-1 3x3 layers in split mode: P and Q share the right feature's split pair (one host call), E runs on its
                # first four columns (another)
                lx_p, lx_q, lx_e = self._commuted_layers_x3(plans, w.detach()[:, c:])
                p_, q_ = ops.conv2d_x3_from_f32(right, [lx_p, lx_q], prep_ws.setdefault("commuted", {}))
                e_ = ops.conv2d_x3_from_f32(right[:, :, :, :4], [lx_e], prep_ws.setdefault("commuted_e", {}))[0]
                _ROUTES["commuted_prep_x3"] += 1
            else:
                lay_p, lay_q, lay_e = self._commuted_layers(plans, w.detach()[:, c:])
                r5 = right.unsqueeze(2)
                p_, q_ = lay_p(r5).squeeze(2), lay_q(r5).squeeze(2)
                e_ = lay_e(right[:, :, :, :4].contiguous().unsqueeze(2)).squeeze(2)
            return p_, q_, e_

        can_commute = commuted and shift.dtype == torch.float32 and left.size(3) <= 2048
        structure, guess, ready, ready_general = (spacing if known else None), None, None, None
        if ticket is not None:
            guess = plans.get("spacing_seen")    # (q, m0, D, W) of the previous call, or ("general", D, W): a guess, checked below
            mark("volume", 0)
            dims = (shift.size(1), left.size(3))
            if sheared and guess is not None and guess[0] != "general" and guess[2:] == dims:
                ready = sheared_inputs(guess[0], guess[1])
            elif can_commute and (not sheared or (guess is not None and guess[0] == "general" and guess[1:] == dims)):
                ready_general = commuted_inputs()
            nonneg, structure = self._shift_structure_end(ticket, shift.size(1))
            assert nonneg                        # same contract as build_cost_volume (reference __init__.py:12)
            if sheared:
                plans["spacing_seen"] = structure + dims if structure is not None else ("general",) + dims
            else:
                structure = None                 # the caller asked for the general path
        if pause:
            yield "shift checked; planes and speculative prep queued"
            # resumed by GlobalStack.forward(volume).  Anything may have run on this model in between -- another build_cost_volume (two
            # pending volumes), a forward_pair, a parameter update: if the prep buffers are no longer this step's own, or the folded
            # first-layer parameters moved, the step starts over from its inputs (the shift array's answer stays: it was checked)
            w_now = conv.weight
            fresh = (self.__dict__.get("_snvc_prep_epoch") == epoch and plans.get("key") == key
                     and (w_now.data_ptr(), w_now._version, w_now.device, _GENERATION[0]) == key
                     and all(a is b for a, b in zip(_folded_bn(bn, plans["plan"]), (scale, bias))))
            if not fresh:
                _ROUTES["lazy_prefetch_stale"] += 1
                return (yield from self._forward_pair_steps(left, right, shift, downsample, factored, timing, True, sheared, fused_bn,
                                                            structure, commuted, arithmetic, pause=False))
        if structure is not None and not self._sheared_fits(structure[0], structure[1], shift.size(1), left.size(3), False):
            structure = None                     # rows the sheared kernels do not cover: the paths below
        if structure is not None:
            q, m0 = structure
            if ready is None or tuple(guess[:2]) != tuple(structure):
                mark("volume", 0)
                ready = sheared_inputs(q, m0)    # first call, or the spacing changed: the guess is dropped
            g, gcol, off, off_col = ready
            mark("volume", 1)
            mark("conv1", 0)
            st = self._x3_select(left.device, arithmetic)
            if st is not None and c % 8 == 0:
                # split mode: the expand pass writes the (hi, lo) pair conv2 reads (same bytes as the fp32 tensor, no layout pass)
                v1s = self._buffer("v1s", (shape[0], 2, c // 8) + tuple(shape[2:]) + (8,), left.device, torch.float16)
                try:
                    ops.sheared_expand_split(g, gcol, planes, *self._x3_v1_affine(st, scale, bias), v1s, q, m0, off, off_col,
                                             ops.EPI_RELU | (ops.EPI_STREAM_OUT if self.stream_out else 0), st["flag"])
                except ops.Unsupported:
                    st = None
                else:
                    mark("conv1", 1)
                    _ROUTES["sheared_first_conv"] += 1
                    self.__dict__["_snvc_last_v1"] = "v1s"
                    return self._tail_x3(st, v1s, timing)
            v = self._buffer("v1", shape, left.device)
            try:
                ops.sheared_expand(g, gcol, planes, scale, bias, v, q, m0, off, off_col, ops.EPI_RELU)
            except ops.Unsupported:              # a limit _sheared_fits does not mirror: same answer on the general paths
                structure = None
            else:
                mark("conv1", 1)
                _ROUTES["sheared_first_conv"] += 1
                return self._conv2_tail(v, shape, timing, arithmetic)
        if can_commute:
            # any other shift array: interpolation along w commutes with the convolution -- three 2D convolutions of the right
            # feature, three interpolations per output voxel, the warped volume is not built either (csrc/sheared_conv.hip)
            if ready_general is None:
                mark("volume", 0)
                ready_general = commuted_inputs()
            p_, q_, e_ = ready_general
            mark("volume", 1)
            mark("conv1", 0)
            st = self._x3_select(left.device, arithmetic)
            if st is not None and c % 8 == 0:       # split mode: the expand pass writes the (hi, lo) pair conv2 reads
                v1s = self._buffer("v1s", (shape[0], 2, c // 8) + tuple(shape[2:]) + (8,), left.device, torch.float16)
                try:
                    ops.warped_expand_split(p_, q_, e_, planes, shift, *self._x3_v1_affine(st, scale, bias), v1s,
                                            ops.EPI_RELU | (ops.EPI_STREAM_OUT if self.stream_out else 0), st["flag"])
                except ops.Unsupported:
                    pass
                else:
                    mark("conv1", 1)
                    _ROUTES["commuted_first_conv"] += 1
                    self.__dict__["_snvc_last_v1"] = "v1s"
                    return self._tail_x3(st, v1s, timing)
            v = self._buffer("v1", shape, left.device)
            try:
                ops.warped_expand(p_, q_, e_, planes, shift, scale, bias, v, ops.EPI_RELU)
            except ops.Unsupported:              # rows that do not fit the LDS (grows with D): build the right half instead
                pass
            else:
                mark("conv1", 1)
                _ROUTES["commuted_first_conv"] += 1
                return self._conv2_tail(v, shape, timing, arithmetic)
        mark("volume", 0)
        try:
            vol_r = ops.cost_volume_forward_right(right, shift, out=self._buffer("vol_r", shape, left.device))   # [N,C,D,H,W]
        except ops.Unsupported:                  # rows beyond the row builder's width: the materialised volume, as the reference
            return self.forward_pair(left, right, shift, downsample, factored=False, timing=timing, arithmetic=arithmetic)
        mark("volume", 1)
        mark("conv1", 0)
        v = plans["right"](vol_r, scale, bias, None, ops.EPI_RELU, self._buffer("v1", shape, left.device), depth_planes=planes)
        mark("conv1", 1)
        return self._conv2_tail(v, shape, timing, arithmetic)
