import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snvc_amd import _lib, ops
from snvc_amd.models import submodule as S
dev = torch.device("cuda:0")
torch.manual_seed(0)
l1, l2 = S.convbn_3d(32, 32, 3, 1, 1).to(dev).train(), S.convbn_3d(32, 32, 3, 1, 1).to(dev).train()
x = torch.randn(1, 32, 6, 8, 40, device=dev, requires_grad=True)
names = [n for n, _ in list(l1.named_parameters()) + list(l2.named_parameters())]
def run(bits, notag=False):
    for p in list(l1.parameters()) + list(l2.parameters()):
        p.grad = None
    if notag:
        orig = ops.amax_of; ops.amax_of = lambda t: None
    with ops.conv_variant(bits):
        y1 = l1(x); out = l2(y1); out.square().mean().backward()
    if notag: ops.amax_of = orig
    return [p.grad.clone() for p in list(l1.parameters()) + list(l2.parameters())]
a, b, c = run(0), run(_lib.ALGO_WGRAD_FP32), run(0, True)
d = run(_lib.ALGO_DIRECT)
for n, ga, gb, gc, gd in zip(names, a, b, c, d):
    m = gd.abs().max().item()
    print(n, tuple(ga.shape), "x3-vs-direct", ((ga - gd).abs().max() / m).item(), "fp32wino-vs-direct", ((gb - gd).abs().max() / m).item(), "x3notag-vs-direct", ((gc - gd).abs().max() / m).item(), "max", m)
