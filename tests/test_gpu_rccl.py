"""RCCL on the GPU box: a real one-rank ``nccl`` group in a child process (tests/helpers/rccl_child.py, started by
conftest.py before this process initialises the GPU -- a process that has touched the GPU must not start programs) runs the
library's flat-bucket gradient collectives.  The reference's counterpart is DataParallel's implicit reduce
(tools/inference_agnostic.py:472 for the scatter; training is the reference's DataParallel too)."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu


def test_rccl_single_rank_gradient_collectives(request):
    child = getattr(request.config, "_snvc_rccl_child", None)
    if child is None:
        pytest.skip("the RCCL child is only started when the run selects the gpu marker (-m gpu)")
    proc, path = child
    try:
        proc.wait(timeout=300)
    except Exception:
        proc.kill()                 # exactly the process this run started
        raise
    assert os.path.exists(path), "the RCCL child wrote no report"
    rep = json.load(open(path))
    assert rep.get("ok"), rep.get("error", rep)
    assert rep["backend"] == "nccl" and rep["world"] == 1
    assert rep["early_out_bytes"] == 0
    grad_bytes = rep["param_bytes"]
    for algo in ("all_reduce", "rs_ag"):
        assert rep[algo]["bytes"] == grad_bytes and rep[algo]["bytes"] > 2_000_000
        assert rep[algo]["gradients_unchanged"], f"{algo}: averaging over one rank must return the local gradients"
    assert rep["unused_stays_none"] and rep["gather_identity"]
    assert rep["auto_large"] == "rs_ag"
    assert all(rep[k] > 0 for k in rep if k.endswith("_us"))
