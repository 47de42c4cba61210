#!/usr/bin/env python3
"""Times every kernel family of the path at its BASELINE-config size and prints achieved GB/s or
TFLOP/s against the roof that bounds it.  Events on the launch stream."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from snvc_amd import ops  # noqa: E402
from snvc_amd.extension.build_cost_volume import build_cost_volume_cuda as CV  # noqa: E402
from snvc_amd.extension.roiaware_pool3d import roiaware_pool3d_utils as RU  # noqa: E402
from snvc_amd.models import submodule as S  # noqa: E402

dev = torch.device("cuda:0")
HBM, MFMA = 8000.0, 157.3


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def row(name, ms, gbytes=None, gflop=None):
    if gbytes is not None:
        r = gbytes / ms * 1e3 / 1e3
        print(f"{name:44s} {ms:8.3f} ms  {gbytes / ms:8.1f} GB/s... ".replace("GB/s... ", "") + f"{gbytes / (ms * 1e-3):9.0f} GB/s  {100 * gbytes / (ms * 1e-3) / HBM:5.1f}% of 8 TB/s")
    else:
        print(f"{name:44s} {ms:8.3f} ms  {gflop / ms:9.1f} TFLOP/s  {100 * gflop / ms / MFMA:5.1f}% of fp32 MFMA")


with torch.no_grad():
    left, right, shift = bench.make_inputs(0, dev)
    vol = CV.build_cost_volume_forward(left, right, shift, 1)
    row("cost_volume fwd  cfg2 [1,64,192,96,312]", timed(lambda: CV.build_cost_volume_forward(left, right, shift, 1)), gbytes=1.4799)
    g = torch.randn_like(vol)
    row("cost_volume bwd  cfg2", timed(lambda: CV.build_cost_volume_backward(g, shift, 1)), gbytes=1.4799)
    del g
    r = np.random.default_rng(0)

    def conv_case(name, cin, cout, k, s, p, dil, shape, transposed=False, n=1):
        m = (S._deconvbn_3d(cin, cout, False) if transposed else S.convbn_3d(cin, cout, k, s, p, dilation=dil)).to(dev).eval()
        x = torch.randn((n, cin) + shape, device=dev)
        y = m.fused(x, relu=True)
        vox = y[0, 0].numel() * n if not transposed else x[0, 0].numel() * n
        gf = 2.0 * vox * cin * cout * (27 if transposed else k ** 3) / 1e9
        row(name, timed(lambda: m.fused(x, relu=True)), gflop=gf)

    conv_case("conv k3 64->32  192x96x312 (cfg2 conv1)", 64, 32, 3, 1, 1, 1, (192, 96, 312))
    del vol
    conv_case("conv k3 32->32  192x96x312 (cfg2 conv2)", 32, 32, 3, 1, 1, 1, (192, 96, 312))
    conv_case("conv k3s2 32->64 192x96x312 (hg conv1)", 32, 64, 3, 2, 1, 1, (192, 96, 312))
    conv_case("conv k3 64->64  96x48x156 (hg conv2)", 64, 64, 3, 1, 1, 1, (96, 48, 156))
    conv_case("deconv 64->32   96x48x156 (hg conv6)", 64, 32, 3, 2, 1, 1, (96, 48, 156), transposed=True)
    conv_case("deconv 64->64   48x24x78 (hg conv5)", 64, 64, 3, 2, 1, 1, (48, 24, 78), transposed=True)
    conv_case("conv k7 64->32  32x128x192 (local conv1)", 64, 32, 7, 1, 3, 1, (32, 128, 192))
    conv_case("conv k5 32->32  32x128x192 (local conv2)", 32, 32, 5, 1, 2, 1, (32, 128, 192))
    conv_case("conv k5d2 32->32 32x128x192 (local conv3)", 32, 32, 5, 1, 4, 2, (32, 128, 192))
    conv_case("conv k1 64->32  32x128x192 (vimg_feat)", 64, 32, 1, 1, 0, 1, (32, 128, 192))
    conv_case("conv k7 64->32  96^3 (cfg3 crop)", 64, 32, 7, 1, 3, 1, (96, 96, 96))

    # wgrad
    x = torch.randn(1, 64, 96, 96, 312, device=dev)
    gy = torch.randn(1, 32, 96, 96, 312, device=dev)
    row("wgrad k3 64->32 96x96x312 (half of cfg2 conv1)", timed(lambda: ops.conv3d_wgrad(x, gy, 3, 1, 1, 1), 3), gflop=2.0 * 96 * 96 * 312 * 64 * 32 * 27 / 1e9)
    del x, gy

    # gather (released shape, 4 instances) + backward
    grid, n = (32, 128, 192), 4
    v = grid[0] * grid[1] * grid[2]
    lf = torch.randn(n, 32, 64, 64, device=dev); rf = torch.randn(n, 32, 64, 64, device=dev)
    base = np.linspace(-8, 264, v, dtype=np.float32)
    gl = torch.from_numpy(np.stack([np.stack([base, base[::-1]])] * n).copy()).to(dev)
    gr = torch.from_numpy(np.stack([np.stack([base[::-1], base])] * n).copy()).to(dev)
    gb = n * (v * 272 + 2 * 32 * 64 * 64 * 4) / 1e9
    row("voxel gather fwd 4x[64,32,128,192]", timed(lambda: ops.voxel_gather_forward(lf, rf, gl, gr, (256, 256))), gbytes=gb)
    go = torch.randn(n, 64, v, device=dev)
    row("voxel gather bwd (atomics)", timed(lambda: ops.voxel_gather_backward(go, gl, gr, (n, 32, 64, 64), (256, 256))), gbytes=gb)
    del go
    # glue kernels
    x = torch.randn(1, 32, 32, 128, 192, device=dev)
    row("avgpool_depth4 [1,32,32,128,192]", timed(lambda: ops.avgpool_depth4(x)), gbytes=x.numel() * 4 * 1.25 / 1e9)
    occ = torch.rand(1, 1, 32, 128, 192, device=dev)
    row("mul_broadcast", timed(lambda: ops.mul_broadcast(x, occ)), gbytes=x.numel() * 4 * (2 + 1 / 32) / 1e9)
    sc = torch.rand(32, device=dev); sh = torch.rand(32, device=dev)
    row("affine_act (norm apply + relu)", timed(lambda: ops.affine_act(x, sc, sh, None, 1)), gbytes=x.numel() * 8 / 1e9)
    row("norm_stats (batch statistics)", timed(lambda: ops.norm_stats(x, sc, sh, 32, False, 1e-5)), gbytes=x.numel() * 4 / 1e9)
    d = torch.randn(1, 192, 96, 312, device=dev); dep = torch.rand(192, device=dev)
    row("disparity_regression [1,192,96,312]", timed(lambda: ops.disparity_regression(d, dep)), gbytes=d.numel() * 4 / 1e9)
    h = torch.randn(64 * 9, 192 * 128, device=dev)
    row("argmax_rows [576, 24576]", timed(lambda: ops.argmax_rows(h)), gbytes=h.numel() * 4 / 1e9)
    # roiaware: 128 boxes, 16384 points, 16 channels, 14^3 voxels
    B, P, C = 128, 16384, 16
    rois = torch.zeros(B, 7, device=dev)
    rois[:, :3] = torch.rand(B, 3, device=dev) * 40 - 20
    rois[:, 3:6] = torch.rand(B, 3, device=dev) * 3 + 1.5
    rois[:, 6] = torch.rand(B, device=dev) * 6.28
    pts = torch.rand(P, 3, device=dev) * 44 - 22
    feat = torch.randn(P, C, device=dev)
    pool = RU.RoIAwarePool3d(14, 128)
    ms = timed(lambda: pool(rois, pts, feat, "max"))
    print(f"{'roiaware_pool3d max B=128 P=16384 C=16 14^3':44s} {ms:8.3f} ms  (mask {B * P * 4 / 1e6:.1f} MB + lists {B * 2744 * 128 * 4 / 1e6:.0f} MB zero-fill by caller)")
