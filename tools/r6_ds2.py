import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from benchlib.local import off_fast_path
out = off_fast_path(torch.device("cuda:0"), 10)
for k in ("cfg2_downsample_2", "cfg2_downsample_2_whole_pixel_planes", "cfg2_groupnorm"):
    print(k, {kk: vv for kk, vv in out[k].items() if kk != "note"})
