import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import torch.nn.functional as F
from oracle import torch_ref as T
from snvc_amd.models import submodule as S
dev = torch.device("cuda:0")
r = np.random.default_rng(71)
ours, ref = S.convbn_3d(32, 32, 3, 1, 1), T.convbn_3d(32, 32, 3, 1, 1)
sd = T.seeded_state_dict(ref, 72)
ours.load_state_dict(sd); ref.load_state_dict(sd)
ours.eval(); ref.eval(); ours = ours.to(dev)
xs = (2, 32, 3, 6, 37)
x = torch.from_numpy(r.standard_normal(xs).astype(np.float32))
res = torch.from_numpy(r.standard_normal(xs).astype(np.float32))
gy = torch.from_numpy(r.standard_normal(xs).astype(np.float32))
xr, rr = x.clone().requires_grad_(), res.clone().requires_grad_()
yr = F.relu(ref(xr) + rr); (yr * gy).sum().backward()
xo, ro = x.to(dev).requires_grad_(), res.to(dev).requires_grad_()
yo = ours.fused(xo, relu=True, residual=ro); (yo * gy.to(dev)).sum().backward()
print("fwd err", (yo.cpu() - yr).abs().max().item())
print("dres err", (ro.grad.cpu() - rr.grad).abs().max().item(), "mismatch count", ((ro.grad.cpu() != 0) != (rr.grad != 0)).sum().item())
e = (xo.grad.cpu() - xr.grad).abs()
print("dx err", e.max().item(), "at", np.unravel_index(e.argmax().item(), xs), "n bad", (e > 1e-3).sum().item())
for k, p in ours.named_parameters():
    q = dict(ref.named_parameters())[k]
    print(k, (p.grad.cpu() - q.grad).abs().max().item(), q.grad.abs().max().item())
# dgrad alone
from snvc_amd import ops
g = ro.grad.detach()  # = g (ADD_PRE)
sc = (ref[1].weight / torch.sqrt(ref[1].running_var + ref[1].eps)).detach()
draw_ref = rr.grad * sc.view(1, -1, 1, 1, 1)
gx_ref = F.conv_transpose3d(draw_ref, ref[0].weight.detach(), None, 1, 1)
print("torch dgrad self-check", (gx_ref - xr.grad).abs().max().item())
