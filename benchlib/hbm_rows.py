"""`roofline_hbm` entries for the SURVEY 8(a) HBM rows that no leg of the headline step launches: a2 (build_cost_volume backward)
and a10 (roiaware_pool3d).  Each: algorithmic bytes per launch / HIP-event time of the launch on torch's current stream."""
import numpy as np
import torch

from .common import C, CV_BYTES, D, H, PEAK_HBM_GBS, W, timed_ms


def cost_volume_backward_row(device, reps=10):
    """a2: the cfg2 volume's gradient [1,64,192,96,312] (1.47 GB read) -> gL, gR [1,32,96,312] (BuildCostVolume_cuda.cu:152-205 there:
    float atomics; here a deterministic gather).  Algorithmic bytes: the volume read once + the two feature gradients written."""
    from snvc_amd import ops
    g = torch.randn(1, 2 * C, D, H, W, device=device)
    shift = torch.from_numpy(np.linspace(0.0, (D - 1) / 2.0, D, dtype=np.float32)[None].copy()).to(device)
    ms, _ = timed_ms(lambda: ops.cost_volume_backward(g, shift, 1), reps, 3)
    nbytes = 4.0 * (2 * C * D * H * W + 2 * C * H * W + D)
    assert nbytes == CV_BYTES
    del g
    torch.cuda.empty_cache()
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"kernel": "cost_volume_bwd (a2): build_cost_volume backward at cfg2 size, deterministic gather (no atomics)",
            "achieved": gbs, "frac": gbs / PEAK_HBM_GBS, "bytes_per_launch": nbytes, "avg_launch_ms": ms}


def roiaware_row(device, boxes=128, points=16384, chan=32, out=14, max_pts=128, reps=10):
    """a10: roiaware_pool3d forward (max pool), `boxes` car-sized rotated boxes over `points` points with `chan` features -> pooled
    [B,14,14,14,C] + argmax (int32, same shape) + pts_idx_of_voxels [B,14,14,14,128] (roiaware_pool3d_kernel.cu:16-190).
    Algorithmic bytes = the three outputs written once + points and features read once PER BOX's membership scan (B x P x 12: the
    scan is the algorithm: every box tests every point) + the features of the points that landed in a box."""
    from snvc_amd import ops
    r = np.random.default_rng(17)
    ctr = r.uniform(-20, 20, (boxes, 3)).astype(np.float32)
    ctr[:, 2] = r.uniform(-1, 1, boxes)
    rois = np.concatenate([ctr, np.tile(np.array([[3.9, 1.6, 1.5]], np.float32), (boxes, 1)),
                           r.uniform(-np.pi, np.pi, (boxes, 1)).astype(np.float32)], axis=1)
    # half of the points near some box (so that voxels fill), half anywhere
    near = ctr[r.integers(0, boxes, points // 2)] + r.normal(0, 0.8, (points // 2, 3)).astype(np.float32)
    far = r.uniform(-22, 22, (points - points // 2, 3)).astype(np.float32)
    pts = torch.from_numpy(np.concatenate([near, far]).astype(np.float32)).to(device)
    feat = torch.from_numpy(r.standard_normal((points, chan)).astype(np.float32)).to(device)
    rois_t = torch.from_numpy(rois).to(device)
    vox = (boxes, out, out, out)
    pooled = torch.zeros(vox + (chan,), device=device)
    argmax = torch.zeros(vox + (chan,), dtype=torch.int32, device=device)
    idx = torch.zeros(vox + (max_pts,), dtype=torch.int32, device=device)

    def run():
        # the caller's zero fill (roiaware_pool3d_utils.py:124-126) is part of the op's contract and of its HBM traffic
        pooled.zero_()
        argmax.zero_()
        idx.zero_()
        ops.roiaware_pool3d_forward(rois_t, pts, feat, argmax, idx, pooled, 0)
        return pooled
    ms, _ = timed_ms(run, reps, 3)
    inside = int((idx[..., 0] > 0).sum().item())
    nbytes = 4.0 * (2 * pooled.numel() + idx.numel()) + 12.0 * boxes * points + 4.0 * chan * points
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"kernel": f"roiaware_pool3d forward (a10): {boxes} boxes x {points} points, C={chan}, out {out}^3, max pool (zero fill of the three "
                      "outputs included: the op's contract)",
            "achieved": gbs, "frac": gbs / PEAK_HBM_GBS, "bytes_per_launch": nbytes, "avg_launch_ms": ms, "occupied_voxels": inside}
