"""Process start-up of bench.py: `--gpus N` without a launcher, and the process group (RCCL; gloo for the CPU rehearsal)."""
import os
import socket
import subprocess
import sys


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(script, argv, n, need_gpus=True):
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU) BEFORE anything in this process touches
    the GPU (torch.cuda.device_count() does not initialise it), wait for them and return the worst exit code.  Rank 0 inherits
    stdout (the result line); ranks > 0 have theirs discarded and share stderr, where every line they print carries a `[tag] `
    prefix (benchlib/emit.py).  A rank that dies takes the others down with it (they would wait in a collective forever)."""
    if need_gpus:
        import torch
        ndev = torch.cuda.device_count()
        if ndev < n:
            raise SystemExit(f"bench.py --gpus {n}: only {ndev} GPU(s) visible on this node")
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc, live, ended_here = 0, list(procs), set()
    import time
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if p.pid not in ended_here:        # a rank this function terminated does not speak for the run
                rc = max(rc, abs(code))
            if code != 0:                      # exactly the processes this function started, by handle
                for q in live:
                    ended_here.add(q.pid)
                    q.terminate()
        time.sleep(0.05)
    return rc


def init_group(backend, rank, world, device=None):
    """(dist module or None, ranks that joined, note).  world == 1 on a GPU: still a real one-rank RCCL group, so that the gradient
    collective of the `train` leg runs through RCCL and the init path the N > 1 runs take is exercised; a failure there must never
    take the headline down."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
        return dist, dist.get_world_size(), f"{backend} x{dist.get_world_size()}"
    try:
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(free_port())
        kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, rank=0, world_size=1, **kw)
        return dist, 1, f"one-rank {backend} ({'RCCL' if backend == 'nccl' else backend}) group"
    except Exception as e:
        return None, 1, f"one-rank {backend} group failed: {type(e).__name__}: {e}"
