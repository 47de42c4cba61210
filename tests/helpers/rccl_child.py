"""One RCCL rank in a fresh process (started by tests/conftest.py BEFORE pytest's own process touches the GPU):
``init_process_group("nccl")`` with a one-rank group on cuda:0, then the library's gradient collectives run for real
(`force=True` bypasses the one-rank early-out), at the 3D stack's bucket size and at SURVEY.md 8(d)'s 133.5 MB
full-model size, in both algorithms.  Writes a JSON report to argv[1]; tests/test_gpu_rccl.py reads it."""
import json
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main(out_path, port):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    rep = {"ok": False}
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        from snvc_amd import parallel as P
        from snvc_amd.models.stereo_volume import GlobalStack
        rep["backend"], rep["world"] = dist.get_backend(), dist.get_world_size()
        torch.manual_seed(0)
        model = GlobalStack(32).to(dev).train()
        left = torch.randn(1, 32, 8, 40, device=dev, requires_grad=True)
        right = torch.randn(1, 32, 8, 40, device=dev, requires_grad=True)
        shift = torch.arange(8, dtype=torch.float32, device=dev)[None] * 0.5
        model.forward_pair(left, right, shift, 1).pow(2).mean().backward()
        params = [p for p in model.parameters() if p.requires_grad]
        local = [None if p.grad is None else p.grad.clone() for p in params]
        rep["early_out_bytes"] = P.all_reduce_gradients(params)                       # one rank, not forced: nothing moves
        for algo in ("all_reduce", "rs_ag"):
            moved = P.all_reduce_gradients(params, average=True, force=True, algorithm=algo)
            same = all((g is None and p.grad is None) or (g is not None and torch.equal(g, p.grad))
                       for g, p in zip(local, params))
            rep[algo] = {"bytes": moved, "gradients_unchanged": bool(same)}
        rep["param_bytes"] = 4 * sum(p.numel() for p in params)
        rep["none_grads"] = sum(g is None for g in local)
        # a parameter nobody gave a gradient keeps .grad = None through the forced collective
        extra = torch.nn.Parameter(torch.zeros(3, device=dev))
        P.all_reduce_gradients(params + [extra], force=True)
        rep["unused_stays_none"] = extra.grad is None
        for name, nbytes in (("stack_2p5MB", rep["param_bytes"]), ("full_model_133p5MB", 133_500_000)):
            for algo in ("all_reduce", "rs_ag"):
                us, used = P.all_reduce_bucket(nbytes, dev, algorithm=algo, reps=5)
                rep[f"{name}_{algo}_us"] = us
        us, used = P.all_reduce_bucket(133_500_000, dev, algorithm="auto", reps=2)
        rep["auto_large"] = used
        out = torch.arange(12, device=dev, dtype=torch.float32).view(4, 3)
        rep["gather_identity"] = bool(torch.equal(P.gather_outputs(out, 4), out))
        dist.destroy_process_group()
        rep["ok"] = True
    except Exception:
        rep["error"] = traceback.format_exc()
    with open(out_path, "w") as fh:
        json.dump(rep, fh)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
