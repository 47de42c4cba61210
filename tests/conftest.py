import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _start_rccl_child(config)


def _start_rccl_child(config):
    """tests/test_gpu_rccl.py needs one RCCL rank in a process of its own.  It is started HERE, before anything in this
    process initialises the GPU (torch.cuda.device_count() does not; torch.cuda.is_available() below does): a process
    that has touched the GPU must not start other programs on the GPU box."""
    expr = config.getoption("-m", default="") or ""
    if "gpu" not in expr or "not gpu" in expr or os.environ.get("SNVC_NO_RCCL_CHILD"):
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:
            return
    except Exception:
        return
    import socket
    import subprocess
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    path = os.path.join(tempfile.mkdtemp(prefix="snvc_rccl_"), "report.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_child.py"), path, str(port)], env=env,
                            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    config._snvc_rccl_child = (proc, path)


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; if someone runs the whole suite on a CPU-only
    # container they are skipped rather than failed.
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
