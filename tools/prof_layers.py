#!/usr/bin/env python3
"""Runs ONE layer of the cfg2 step a few times so that a rocprofv3 pass stays small.

    rocprofv3 --pmc ... -- python tools/prof_layers.py conv1 --reps 3
Layers: cost_volume, cost_volume_right, cost_volume_bwd, conv1 (k3 64->32), conv1_factored, conv2 (k3 32->32), hg
(hourglass), classifier, gather (degenerate line, r1), gather_proj / gather_uniform (GridProjector / uniform coordinates,
2 crops of the released shape), gather_f16 (cfg5, C8 half), f16_k7 / f16_k5 (cfg5's dominant fp16 layers), trunk.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("layer")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--arithmetic", default=None, help="GlobalStack.arithmetic for the layers that run forward_pair: fp32 | x3")
args = ap.parse_args()
dev = torch.device("cuda:0")

from snvc_amd.extension.build_cost_volume import build_cost_volume  # noqa: E402
from snvc_amd.models.stereo_volume import GlobalStack  # noqa: E402

model = GlobalStack(bench.C)
model.load_state_dict(bench.seeded_state(model))
model.eval().to(dev)
if args.arithmetic:
    model.arithmetic = args.arithmetic
left, right, shift = bench.make_inputs(0, dev)
with torch.no_grad():
    if args.layer in ("gather", "trunk", "trunk_f16"):
        import types
        from snvc_amd.models.vernier import VernierScale
        grid = (32, 128, 192)
        n = 4 if args.layer == "gather" else 1
        cfg = types.SimpleNamespace(vernier_type="BEV_type3", backbone="hrfeat", gn=False, grid_resolution=list(grid),
                                    resolution=(256, 256), x_range=(-1.0, 1.0), z_range=(-1.0, 1.0), num_parts=9)
        cfg.hrfeat = types.SimpleNamespace(output_channel=32, name="identity")
        cfg.n_sample_h, cfg.n_sample_w, cfg.n_sample_l = grid
        vs = VernierScale(cfg)
        vs.load_state_dict(bench.seeded_state(vs))
        vs.eval().to(dev)
        r = np.random.default_rng(7)
        v = grid[0] * grid[1] * grid[2]
        lf = torch.from_numpy(r.standard_normal((n, 32, 64, 64)).astype(np.float32)).to(dev)
        rf = torch.from_numpy(r.standard_normal((n, 32, 64, 64)).astype(np.float32)).to(dev)
        # smooth projections (a plane sweep across the crop) + 6 % outside, like real grids
        base = np.linspace(-8, 264, v, dtype=np.float32)
        gl = torch.from_numpy(np.stack([np.stack([base, base[::-1]])] * n).copy()).to(dev)
        gr = torch.from_numpy(np.stack([np.stack([base[::-1], base])] * n).copy()).to(dev)
        if args.layer == "gather":
            fn = lambda: vs.construct_voxel(lf, rf, gl, gr)  # noqa: E731
        elif args.layer == "trunk_f16":      # released shape, fp16-storage mode: gather (C8 half) + trunk, 2 crops per call
            lf2, rf2, gl2, gr2 = (torch.cat([t, t]) for t in (lf, rf, gl, gr))
            fn = lambda: vs.trunk_3d_f16(vs.construct_voxel_f16(lf2, rf2, gl2, gr2))  # noqa: E731
        else:
            vox = vs.construct_voxel(lf, rf, gl, gr)
            fn = lambda: vs.trunk_3d(vox)  # noqa: E731
    elif args.layer in ("gather_proj", "gather_uniform", "gather_f16", "gather_cfg3"):
        import types
        from snvc_amd import ops
        f16 = args.layer == "gather_f16"
        grid, n, F = ((80, 160, 160), 1, 64) if f16 else ((32, 128, 192), 2, 32)
        if args.layer == "gather_cfg3":          # BASELINE configs[2]: 8 crops of 96^3 per GPU
            grid, n, F = (96, 96, 96), 8, 32
        r = np.random.default_rng(5)
        v = grid[0] * grid[1] * grid[2]
        lf = torch.from_numpy(r.standard_normal((n, F, 64, 64)).astype(np.float32)).to(dev)
        rf = torch.from_numpy(r.standard_normal((n, F, 64, 64)).astype(np.float32)).to(dev)
        if args.layer == "gather_uniform":
            gl = torch.from_numpy(r.uniform(-8, 264, (n, 2, v)).astype(np.float32)).to(dev)
            gr = torch.from_numpy(r.uniform(-8, 264, (n, 2, v)).astype(np.float32)).to(dev)
        else:
            gl, gr = bench.projected_coordinates(n, grid, dev)
        fn = (lambda: ops.voxel_gather_forward_f16(lf, rf, gl, gr, (256, 256))) if f16 else (lambda: ops.voxel_gather_forward(lf, rf, gl, gr, (256, 256)))  # noqa: E731
    elif args.layer in ("f16_k7", "f16_k5"):
        from snvc_amd.models import submodule as S
        k, cin = (7, 128) if args.layer == "f16_k7" else (5, 64)
        m = S.convbn_3d(cin, 64, k, 1, (k - 1) // 2).to(dev).eval()
        xh = torch.randn((1, cin // 8, 80, 160, 160, 8), device=dev).half()
        fn = lambda: m.fused_f16(xh, relu=True)  # noqa: E731
    elif args.layer == "conv2_side":        # r3 dominant kernel: conv2 + the classifier's projection of its own result
        from snvc_amd.models import submodule as S
        v1 = torch.randn(1, bench.C, bench.D, bench.H, bench.W, device=dev)
        out = torch.empty_like(v1)
        fn = lambda: model.conv2.fused(v1, out=out, side_head=model.classifier)  # noqa: E731
    elif args.layer == "hg_s2":             # hourglass conv1: k3 / stride 2, 32 -> 64 (slice-pipelined refill)
        v2 = torch.randn(1, bench.C, bench.D, bench.H, bench.W, device=dev)
        fn = lambda: model.hg_conv3d.conv1(v2)  # noqa: E731
    elif args.layer in ("x3_conv2", "x3_hg2"):      # split-mode (f16x3) layers: conv2 32->32 full size, hg conv2 64->64 half size
        from snvc_amd import ops
        cin, shp = (32, (bench.D, bench.H, bench.W)) if args.layer == "x3_conv2" else (64, (bench.D // 2, bench.H // 2, bench.W // 2))
        xin = torch.relu(torch.randn(1, cin, *shp, device=dev))
        wt = torch.randn(cin, cin, 3, 3, 3, device=dev) * 0.05
        lay = ops.Conv3dLayerX3(wt)
        xs_ = ops.to_split(xin, 4)
        del xin
        ys_ = torch.empty_like(xs_)
        flag_ = torch.zeros(1, dtype=torch.int32, device=dev)
        head_ = torch.randn(cin, device=dev) if cin == 32 else None      # conv2 carries the classifier's side head
        fn = lambda: lay(xs_, 4, flags=ops.EPI_RELU, out=ys_, out_exp=4, head=head_, overflow=flag_)  # noqa: E731
    elif args.layer == "x3_s2":             # split-mode hourglass conv1: k3 / stride 2, 32 -> 64 on the full grid
        from snvc_amd import ops
        xin = torch.relu(torch.randn(1, bench.C, bench.D, bench.H, bench.W, device=dev))
        wt = torch.randn(2 * bench.C, bench.C, 3, 3, 3, device=dev) * 0.05
        lay = ops.Conv3dLayerX3(wt, 3, 2, 1)
        xs_ = ops.to_split(xin, 4)
        del xin
        flag_ = torch.zeros(1, dtype=torch.int32, device=dev)
        ys_ = lay(xs_, 4, flags=ops.EPI_RELU, out_exp=4, overflow=flag_)
        fn = lambda: lay(xs_, 4, flags=ops.EPI_RELU, out=ys_, out_exp=4, overflow=flag_)  # noqa: E731
    elif args.layer in ("x3_hg5_tail", "tail_gather"):     # r5: hourglass conv5 (transposed 64->64 + pre, ReLU) with the tail projection; the gather
        from snvc_amd import ops
        q = (bench.D // 4, bench.H // 4, bench.W // 4)
        xin = torch.relu(torch.randn(1, 64, *q, device=dev))
        wt = torch.randn(64, 64, 3, 3, 3, device=dev) * 0.05
        lay = ops.Conv3dLayerX3(wt, 3, 2, 1, 1, True)
        tail = ops.TailWeightsX3(torch.randn(64, 27, device=dev) * 0.1)
        xs_ = ops.to_split(xin, 4)
        pre_ = ops.to_split(torch.relu(torch.randn(1, 64, *(2 * e for e in q), device=dev)), 3)
        flag_ = torch.zeros(1, dtype=torch.int32, device=dev)
        t_ = lay.forward_tail(xs_, 4, None, None, tail, residual=pre_, flags=ops.EPI_RELU | ops.EPI_ADD_PRE, out_exp=3, overflow=flag_)
        hv_ = torch.randn(1, 1, bench.D, bench.H, bench.W, device=dev)
        if args.layer == "x3_hg5_tail":
            fn = lambda: lay.forward_tail(xs_, 4, None, None, tail, residual=pre_, flags=ops.EPI_RELU | ops.EPI_ADD_PRE, out_exp=3, overflow=flag_, out=t_)  # noqa: E731
        else:
            fn = lambda: ops.deconv_tail_gather(t_, None, hv_)  # noqa: E731
    elif args.layer in ("general_f32", "sheared_f32"):     # the fp32 expand kernels (arithmetic = fp32)
        model.arithmetic = "fp32"
        fn = lambda: model.forward_pair(left, right, shift, 1, sheared=args.layer == "sheared_f32")  # noqa: E731
    elif args.layer == "general":           # any shift array: warp after convolution (three depth-1 convs + warped_expand)
        fn = lambda: model.forward_pair(left, right, shift, 1, sheared=False)  # noqa: E731
    elif args.layer == "sheared_bwd":       # cfg4's first layer: forward + backward of the sheared function with folded BatchNorm
        model.train()
        lt, rt = left.clone().requires_grad_(), right.clone().requires_grad_()
        def fn():
            with torch.enable_grad():
                v = model.forward_pair(lt, rt, shift, 1)
                v.mean().backward()
    elif args.layer == "sheared":           # the sheared first convolution: Rq, G | G', edge slab, expand, copies
        fn = lambda: model.forward_pair(left, right, shift, 1)  # noqa: E731
    elif args.layer == "f16_k7_32":         # released shape, fp16 storage, MI = 1 form (64 -> 32)
        from snvc_amd.models import submodule as S
        m = S.convbn_3d(64, 32, 7, 1, 3).to(dev).eval()
        xh = torch.randn((2, 8, 32, 128, 192, 8), device=dev).half()
        fn = lambda: m.fused_f16(xh, relu=True)  # noqa: E731
    elif args.layer in ("wgrad_conv2", "wgrad_s2", "wgrad_hg"):     # cfg4: weight gradients of conv2 (k3 s1 32->32), of the
        from snvc_amd import ops                                     # hourglass's stride-2 layer and of its 64->64 layer
        if args.layer == "wgrad_conv2":
            xb = torch.randn(1, bench.C, bench.D, bench.H, bench.W, device=dev); gs = torch.randn_like(xb); st = 1
        elif args.layer == "wgrad_hg":
            xb = torch.randn(1, 2 * bench.C, bench.D // 2, bench.H // 2, bench.W // 2, device=dev); gs = torch.randn_like(xb); st = 1
        else:
            xb = torch.randn(1, bench.C, bench.D, bench.H, bench.W, device=dev)
            gs = torch.randn(1, 2 * bench.C, bench.D // 2, bench.H // 2, bench.W // 2, device=dev); st = 2
        fn = lambda: ops.conv3d_wgrad(xb, gs, 3, st, 1, 1)  # noqa: E731
    elif args.layer == "cost_volume_right":
        from snvc_amd import ops
        fn = lambda: ops.cost_volume_forward_right(right, shift)  # noqa: E731
    elif args.layer == "cost_volume_bwd":
        from snvc_amd import ops
        g = torch.randn(1, 2 * bench.C, bench.D, bench.H, bench.W, device=dev)
        fn = lambda: ops.cost_volume_backward(g, shift, 1)  # noqa: E731
    elif args.layer == "pair":
        fn = lambda: model.forward_pair(left, right, shift, 1)  # noqa: E731
    elif args.layer == "conv1_factored":
        from snvc_amd import ops
        model.forward_pair(left, right, shift, 1)                     # builds the factored plans
        plans = model.conv1[0][0].__dict__["_snvc_factored"]
        from snvc_amd.models.submodule import _folded_bn
        scale, bias = _folded_bn(model.conv1[0][1], plans["plan"])
        wl = model.conv1[0][0].weight.detach()[:, :bench.C]
        planes = model._left_planes_layer(plans, wl)(left.unsqueeze(2)).view(1, bench.C, 3, bench.H, bench.W)
        vol_r = ops.cost_volume_forward_right(right, shift)
        fn = lambda: plans["right"](vol_r, scale, bias, None, ops.EPI_RELU, None, depth_planes=planes)  # noqa: E731
    elif args.layer == "cost_volume":
        fn = lambda: build_cost_volume(left, right, shift, 1)  # noqa: E731
    else:
        vol = build_cost_volume(left, right, shift, 1)
        if args.layer == "conv1":
            fn = lambda: model.conv1(vol)  # noqa: E731
        else:
            v1 = model.conv1(vol)
            del vol
            if args.layer == "conv2":
                fn = lambda: model.conv2(v1)  # noqa: E731
            else:
                v2 = model.conv2(v1)
                if args.layer == "hg":
                    fn = lambda: model.hg_conv3d(v2, None, None, residual=v2)  # noqa: E731
                elif args.layer == "classifier":
                    fn = lambda: model.classifier(v2)  # noqa: E731
                else:
                    raise SystemExit("unknown layer")
    for _ in range(args.reps):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{args.layer}: {a.elapsed_time(b) / args.reps:.3f} ms")
