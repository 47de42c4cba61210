"""Pins the C oracle of roiaware_pool3d with hand-derived known answers (the reference ships no
vectors for this op; each case cites the lines of roiaware_pool3d_kernel.cu it exercises)."""
import numpy as np

from oracle import native as O


def test_axis_aligned_voxel_indices():
    # box centred at origin, 4 x 6 x 8, heading 0, out 2x3x4 -> voxel size 2x2x2 (.cu:60-72)
    roi = np.array([[0, 0, 0, 4, 6, 8, 0]], np.float32)
    pts = np.array([[-1.0, -2.0, -3.0],     # -> (0,0,0)
                    [1.0, 0.0, 1.0],        # -> (1,1,2)
                    [1.9, 2.9, 3.9],        # -> (1,2,3)
                    [0.0, 1.0, 0.0],        # on voxel boundaries -> int() floors: (1,2,2)
                    [2.5, 0.0, 0.0],        # outside in x
                    [0.0, 0.0, 4.5]], np.float32)   # outside in z (|z-cz| > dz/2, .cu:32)
    mask = O.roiaware_mask(roi, pts, (2, 3, 4))[0]
    enc = lambda x, y, z: (x << 16) + (y << 8) + z  # noqa: E731
    assert mask.tolist() == [enc(0, 0, 0), enc(1, 1, 2), enc(1, 2, 3), enc(1, 2, 2), -1, -1]


def test_z_face_is_inclusive_and_xy_margin():
    roi = np.array([[0, 0, 0, 2, 2, 2, 0]], np.float32)
    pts = np.array([[0, 0, 1.0],            # |z-cz| == dz/2 is NOT rejected (strict >, .cu:32)
                    [1.0, 0, 0],            # |lx| == dx/2 < dx/2 + 1e-5 -> inside (.cu:27,34)
                    [1.00002, 0, 0]], np.float32)   # beyond the 1e-5 margin
    mask = O.roiaware_mask(roi, pts, (2, 2, 2))[0]
    assert mask[0] == (1 << 16) + (1 << 8) + 1      # clamped to out-1 (.cu:68-70)
    assert mask[1] == (1 << 16) + (1 << 8) + 1
    assert mask[2] == -1


def test_heading_half_pi_swaps_axes():
    # rotate by -heading (.cu:16-20): heading pi/2 maps global +y onto local +x
    roi = np.array([[0, 0, 0, 4, 2, 2, np.pi / 2]], np.float32)
    # local_x = sx*cos(-h) - sy*sin(-h) = sy ; local_y = sx*sin(-h) + sy*cos(-h) = -sx
    pts = np.array([[-0.5, 1.5, 0.0], [1.5, 0.0, 0.0]], np.float32)
    mask = O.roiaware_mask(roi, pts, (4, 2, 2))[0]
    assert mask[0] == (3 << 16) + (1 << 8) + 1      # local (1.5, 0.5) -> voxel (3, 1) of (4, 2)
    assert mask[1] == -1                            # local y = -1.5 is outside dy/2 = 1


def test_lists_are_ascending_and_truncate_at_127():
    roi = np.array([[0, 0, 0, 2, 2, 2, 0]], np.float32)
    n = 300
    pts = np.tile(np.array([[0.5, 0.5, 0.5]], np.float32), (n, 1))
    feat = np.arange(n, dtype=np.float32)[:, None] * np.array([[1.0, -1.0]], np.float32)
    pooled, argmax, lists = O.roiaware_pool3d_forward(roi, pts, feat, (2, 2, 2), 128, "max")
    cell = lists[0, 1, 1, 1]
    assert cell[0] == 127 and cell[1:128].tolist() == list(range(127))      # .cu:86-100
    assert lists[0, 0, 0, 0, 0] == 0
    # max over the kept points only: channel 0 -> last kept (126), channel 1 -> first (0)
    assert argmax[0, 1, 1, 1].tolist() == [126, 0] and pooled[0, 1, 1, 1].tolist() == [126.0, 0.0]
    assert argmax[0, 0, 0, 0].tolist() == [-1, -1] and pooled[0, 0, 0, 0].tolist() == [0.0, 0.0]   # .cu:136-151
    pooled_avg, _, _ = O.roiaware_pool3d_forward(roi, pts, feat, (2, 2, 2), 128, "avg")
    assert pooled_avg[0, 1, 1, 1, 0] == np.float32(sum(range(127))) / 127                          # .cu:180-189


def test_max_tie_keeps_first_and_backward():
    roi = np.array([[0, 0, 0, 2, 2, 2, 0]], np.float32)
    pts = np.array([[0.5, 0.5, 0.5]] * 3 + [[-0.5, -0.5, -0.5]], np.float32)
    feat = np.array([[1.0], [5.0], [5.0], [2.0]], np.float32)
    pooled, argmax, lists = O.roiaware_pool3d_forward(roi, pts, feat, (2, 2, 2), 128, "max")
    assert argmax[0, 1, 1, 1, 0] == 1                # strict > : first maximum wins (.cu:142)
    g = np.ones_like(pooled)
    gin = O.roiaware_pool3d_backward(lists, argmax, g, 4, "max")
    assert gin[:, 0].tolist() == [0.0, 1.0, 0.0, 1.0]                                              # .cu:255-257
    gin = O.roiaware_pool3d_backward(lists, argmax, g, 4, "avg")
    np.testing.assert_allclose(gin[:, 0], [1 / 3, 1 / 3, 1 / 3, 1.0], rtol=1e-6)                   # .cu:281-285


def test_points_in_boxes_variants():
    boxes = np.array([[[0, 0, 0, 2, 2, 2, 0], [0.5, 0, 0, 2, 2, 2, 0]]], np.float32)
    pts = np.array([[[0.9, 0, 0], [1.4, 0, 0], [5, 5, 5], [1.005, 0, 0]]], np.float32)
    got = O.points_in_boxes_gpu(pts, boxes)
    assert got.tolist() == [[0, 1, -1, 1]]           # first containing box (.cu:329-335)
    flags = O.points_in_boxes_cpu(pts[0], boxes[0])  # margin 1e-2 (roiaware_pool3d.cpp:131)
    assert flags.tolist() == [[1, 0, 0, 1], [1, 1, 0, 1]]
