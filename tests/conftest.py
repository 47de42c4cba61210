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
    # nothing that will not reach tests/test_gpu_rccl.py starts a rank: a listing, an xdist worker (one child EACH otherwise)
    if config.getoption("collectonly", default=False) or hasattr(config, "workerinput") or os.environ.get("PYTEST_XDIST_WORKER"):
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
    log = open(os.path.join(os.path.dirname(path), "child.log"), "wb")      # beside report.json: a crash of the child is readable
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_child.py"), path, str(port)], env=env,
                            stdout=log, stderr=subprocess.STDOUT)
    log.close()
    config._snvc_rccl_child = (proc, path)


def pytest_unconfigure(config):
    """The RCCL child must not outlive the session (-x on an earlier failure, -k / --deselect of the test that waits for it):
    wait briefly, then kill exactly that process."""
    child = getattr(config, "_snvc_rccl_child", None)
    if child is None:
        return
    proc = child[0]
    if proc.poll() is None:
        try:
            proc.wait(timeout=20)
        except Exception:
            proc.kill()
            try:
                proc.wait(timeout=10)
            except Exception:
                pass


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
