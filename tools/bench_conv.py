#!/usr/bin/env python3
"""Steady-state timing of single conv layers (development tool).  The shader clock ramps from
~2.1 to ~2.4 GHz over the first tens of milliseconds of load (tools/probes/mfma_clock_probe.hip),
so each case is warmed for ~100 ms first and the median / minimum of per-launch events is printed."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from snvc_amd.models import submodule as S  # noqa: E402

CASES = {
    "conv1": (64, 32, 3, 1, 1, (192, 96, 312), False),
    "conv2": (32, 32, 3, 1, 1, (192, 96, 312), False),
    "hg_s2": (32, 64, 3, 2, 1, (192, 96, 312), False),
    "hg_c2": (64, 64, 3, 1, 1, (96, 48, 156), False),
    "hg_s2b": (64, 64, 3, 2, 1, (96, 48, 156), False),
    "hg_c4": (64, 64, 3, 1, 1, (48, 24, 78), False),
    "dc5": (64, 64, 3, 2, 1, (48, 24, 78), True),
    "dc6": (64, 32, 3, 2, 1, (96, 48, 156), True),
    # local model (released shape): cin, cout, k, stride, pad, shape, transposed[, dilation]
    "l_k7": (64, 32, 7, 1, 3, (32, 128, 192), False),
    "l_k5": (32, 32, 5, 1, 2, (32, 128, 192), False),
    "l_k5d2": (32, 32, 5, 1, 4, (32, 128, 192), False, 2),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*", default=[c for c in CASES if not c.startswith("l_")])
    ap.add_argument("--reps", type=int, default=60)
    ap.add_argument("--side-head", action="store_true",
                    help="conv2 with and without the side head (snvc_conv3d_forward_side_head), interleaved in one process")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.side_head:
        with torch.no_grad():
            m = S.ConvBNReLU3d(S.convbn_3d(32, 32, 3, 1, 1), torch.nn.ReLU()).to(dev).eval()
            head = S.HipConv3d(32, 1, kernel_size=1, padding=0, stride=1, bias=False).to(dev)
            x = torch.randn((1, 32, 192, 96, 312), device=dev)
            out = torch.empty_like(x)
            fns = {"plain": lambda: m.fused(x, out=out), "side_head": lambda: m.fused(x, out=out, side_head=head),
                   "plain+pointwise": lambda: head(m.fused(x, out=out))}
            for f in fns.values():
                for _ in range(10):
                    f()
            times = {k: [] for k in fns}
            for _ in range(args.reps // 3):
                for k, f in fns.items():
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(); f(); b.record()
                    torch.cuda.synchronize()
                    times[k].append(a.elapsed_time(b))
            for k, t in times.items():
                t = sorted(t)
                print(f"conv2 {k:16s}: median {t[len(t) // 2]:7.3f} ms  min {t[0]:7.3f} ms", flush=True)
        return
    with torch.no_grad():
        for name in args.cases:
            cin, cout, k, s, p, shape, tr = CASES[name][:7]
            dil = CASES[name][7] if len(CASES[name]) > 7 else 1
            m = (S._deconvbn_3d(cin, cout, False) if tr else S.convbn_3d(cin, cout, k, s, p, dilation=dil)).to(dev).eval()
            x = torch.randn((1, cin) + shape, device=dev)
            y = m.fused(x, relu=True)
            vox = x[0, 0].numel() if tr else y[0, 0].numel()
            gf = 2.0 * vox * cin * cout * (27 if tr else k ** 3) / 1e9
            for _ in range(40):
                m.fused(x, relu=True)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
            for a, b in ev:
                a.record(); m.fused(x, relu=True); b.record()
            torch.cuda.synchronize()
            t = sorted(a.elapsed_time(b) for a, b in ev)
            med, mn = t[len(t) // 2], t[0]
            print(f"{name:7s} {cin:3d}->{cout:3d} {'deconv' if tr else 'k3s%d' % s:6s} {shape}: median {med:7.3f} ms "
                  f"({gf / med:6.1f} TF)  min {mn:7.3f} ms ({gf / mn:6.1f} TF)", flush=True)
            del m, x, y


if __name__ == "__main__":
    main()
