"""bench.py's output rule (benchlib/emit.py), held to the driver's view of a run: stdout, then `---- stderr ----`, then stderr; the
LAST line json.loads accepts is the result.  r5 lost its driver measurement to `rank_record` JSON lines on stderr
(BENCH_r05.json: "the bench printed no result line").  CPU only: the launcher / group / timing protocol / emission run end to
end through ``bench.py --rehearse`` (gloo, a sleep as the step, no kernel)."""
import glob
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from benchlib import emit  # noqa: E402


def _capture(stdout, stderr):
    return stdout + "\n\n---- stderr ----\n" + stderr


def _run(args, env_extra=None, timeout=240):
    env = dict(os.environ, **(env_extra or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=timeout)
    return p.returncode, p.stdout, p.stderr


def _bare_json_lines(text):
    out = []
    for ln in text.splitlines():
        try:
            obj = json.loads(ln)
        except ValueError:
            continue
        if isinstance(obj, dict):
            out.append(obj)
    return out


def test_r5_shape_of_failure_is_what_last_json_line_sees():
    """The r5 capture in miniature: a result line on stdout, bare-JSON rank records on stderr -> the driver picks a rank record."""
    result = json.dumps({"metric": "m", "value": 1.0})
    bad = _capture(result, json.dumps({"rank_record": {"rank": 0}}) + "\n" + json.dumps({"rank_record_sustained": {"rank": 0}}))
    assert "metric" not in emit.last_json_line(bad)                       # what happened
    import io
    err = io.StringIO()
    emit.rank_note("rank_record", {"rank": 0}, stream=err)
    emit.rank_note("rank_record_sustained", {"rank": 0, "x": float("nan")}, stream=err)
    good = _capture(result, err.getvalue())
    assert emit.last_json_line(good) == {"metric": "m", "value": 1.0}    # what the prefix buys
    assert _bare_json_lines(err.getvalue()) == []
    assert err.getvalue().count("[rank_record") == 2 and "NaN" not in err.getvalue()


def test_result_text_is_compact_nan_free_and_complete():
    import numpy as np
    line = {k: 1 for k in emit.REQUIRED}
    line.update(value=np.float32(2.5), ms_per_step=float("nan"), config={"workload": "w", "note": "n" * 5000, "prewarm": "p" * 900},
                roofline={"frac": float("inf"), "kernel": "k" * 1000, "note": "x" * 3000, "traffic": None, "nested": {"note": "y", "a": (1, 2)}},
                cpu_baseline={"value": 0.2, "unit": "u", "cores": 128, "kind": "port", "sample": "s" * 150})
    text = emit.result_text(line)
    rec = json.loads(text)                                                # strict JSON
    json.dumps(rec, allow_nan=False)
    assert rec["value"] == 2.5 and rec["ms_per_step"] is None and rec["roofline"]["frac"] is None
    assert "note" not in rec["config"] and "prewarm" not in rec["config"] and "note" not in rec["roofline"]["nested"]
    assert rec["roofline"]["nested"]["a"] == [1, 2] and len(rec["roofline"]["kernel"]) <= emit.MAX_STR
    assert rec["cpu_baseline"]["sample"] == "s" * 150                     # the contract's own fields stay
    assert len(text) < 2000
    assert json.loads(emit.result_text(line, provisional=True))["provisional"] is True
    del line["roofline"]
    with pytest.raises(KeyError, match="roofline"):
        emit.result_text(line)
    json.loads(emit.result_text(line, required=emit.CONTRACT))


def test_nothing_but_emit_prints_bare_json():
    """Static: every print of bench.py and benchlib/ goes through emit (rank_note / detail_note / emit_result)."""
    files = [os.path.join(ROOT, "bench.py")] + sorted(glob.glob(os.path.join(ROOT, "benchlib", "*.py")))
    for path in files:
        src = open(path).read()
        if path.endswith("emit.py"):
            continue
        assert not re.search(r"print\s*\(\s*json\.dumps", src), path
        assert "json.dumps" not in src, path
        for m in re.finditer(r"^\s*print\(", src, re.M):
            raise AssertionError(f"{path}: a bare print at offset {m.start()} (use benchlib.emit)")


@pytest.mark.parametrize("world", [1, 2])
def test_rehearsal_end_to_end_result_line_is_last(world):
    rc, out, err = _run(["--gpus", str(world), "--steps", "5", "--warmup", "2", "--rehearse"])
    assert rc == 0, (rc, out[-2000:], err[-2000:])
    cap = _capture(out, err)
    last = emit.last_json_line(cap)
    assert last is not None and last["metric"].startswith("REHEARSAL") and last["n_gpus"] == world and last["steps"] == 5
    for k in emit.CONTRACT:
        assert k in last, k
    assert last["not_finite_example"] is None                             # NaN never reaches the line
    assert len(last["ranks"]) == world and sorted(r["rank"] for r in last["ranks"]) == list(range(world))
    assert len(_bare_json_lines(cap)) == 1                                 # one result line, nothing else parses
    assert len(re.findall(r"^\[rank_record\] ", err, re.M)) == world       # N rank records, prefixed
    assert out.strip().splitlines()[-1].startswith("{")                    # and it is the last thing on stdout
    assert "[bench_detail] " in out
    # wall-clock consistency: K steps of >= 1 ms
    assert last["ms_per_step"] >= 1.0 and abs(last["value"] - world * 1e3 / last["ms_per_step"]) < 1e-6 * last["value"]


def test_rehearsal_exit_code_is_the_worst_ranks():
    rc, out, err = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse"], {"SNVC_REHEARSE_FAIL_RANK": "1"})
    assert rc == 3, (rc, err[-1500:])
    rc, out, err = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse"], {"SNVC_REHEARSE_FAIL_RANK": "0"})
    assert rc != 0 and _bare_json_lines(out) == []                          # rank 0 failed: no result line at all, non-zero exit


def test_committed_gpu_capture_parses():
    """A real capture of `python bench.py` on an MI355X (tests/golden/bench_capture_*.txt: stdout + the driver's separator + stderr,
    committed by the round that produced it): the last JSON line is the result line with roofline and cpu_baseline."""
    caps = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "bench_capture_*.txt")))
    if not caps:
        pytest.skip("no committed capture yet")
    for path in caps:
        last = emit.last_json_line(open(path).read())
        assert last is not None, path
        for k in emit.REQUIRED:
            assert k in last, (path, k)
        assert "provisional" not in last, path
        json.dumps(last, allow_nan=False)
        r, c = last["roofline"], last["cpu_baseline"]
        assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1 and r["achieved"] > 0 and r["peak"] > 0
        assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] in ("port", "reference") and c["sample"]
        assert last["value"] > 0 and abs(last["value"] * last["ms_per_step"] / 1e3 / last["n_gpus"] - 1.0) < 1e-6
        assert last["dtype"].startswith("f32") and last["value_fp32_mfma"] > 0
