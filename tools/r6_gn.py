import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from snvc_amd.models.stereo_volume import GlobalStack
from snvc_amd.models import submodule as S
dev = torch.device("cuda:0")
left, right, shift = bench.make_inputs(0, dev)
g = GlobalStack(bench.C, gn=True); g.load_state_dict(bench.seeded_state(g)); g.eval().to(dev)
with torch.no_grad():
    for tag, kw in (("sheared first layer (r6)", {}), ("materialised (r5)", {"sheared": False}), ("fp32 mfma", {"arithmetic": "fp32"})):
        b = S._ROUTES["gn_sheared_first_conv"]
        ms, out = bench.timed_ms(lambda: g.forward_pair(left, right, shift, 1, **kw), 10, 3)
        print(tag, round(ms, 3), "ms", round(1e3 / ms, 1), "pairs/s", "route taken", S._ROUTES["gn_sheared_first_conv"] > b, flush=True)
    a = g.forward_pair(left, right, shift, 1); bb = g.forward_pair(left, right, shift, 1, sheared=False)
    print("rel diff sheared vs materialised", ((a - bb).abs().max() / bb.abs().max()).item())
