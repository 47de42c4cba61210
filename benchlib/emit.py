"""What bench.py prints, and the rule that governs it.

THE RULE (r6; r5 broke it): the driver captures stdout, appends stderr, and takes the LAST line of that text that
``json.loads`` accepts as the result.  Therefore

  * exactly one kind of line in this process may be bare JSON: the result line, printed on stdout by rank 0 through
    ``emit_result`` -- last;
  * everything else (per-rank records, leg-by-leg detail, breakdowns) goes through ``rank_note`` / ``detail_note``, which
    prefix the text with ``[tag] `` so that no JSON parser accepts the line, on any rank, on stdout or stderr;
  * the result line is COMPACT (a few KB: numbers, kernel names, samples -- no essays); the long form with every note goes out
    as a ``[bench_detail] {...}`` line in front of it and, best effort, to ``gpurun_out/bench_detail.json``;
  * it contains no NaN / Infinity (``allow_nan=False``): a bracket a leg did not record is ``null``.

``last_json_line`` is the driver's view of a captured run; tests/test_bench_emit.py holds real and synthetic captures to it.
"""
import json
import math
import os
import sys

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")
REQUIRED = CONTRACT + ("roofline", "cpu_baseline")      # the headline line at N = 1 (cpu_baseline: rank 0 at N = 1 only)
# keys whose values are prose: kept in the detail line, dropped from the compact result line
PROSE_KEYS = frozenset({"note", "rule", "prewarm", "entry_points", "measured_in", "prep", "tolerance", "coords", "source", "arithmetic_note",
                        "traffic_source_note"})
# ... and below the second level (configs.<name>.*, off_fast_path.<name>.*): descriptions of legs that are not the headline
DEEP_PROSE_KEYS = frozenset({"sample", "dominant_kernel", "workload", "arithmetic", "outputs_gathered", "kernel"})
MAX_STR = 200


def sanitize(obj):
    """JSON-safe copy: NaN / +-Infinity -> None, numpy scalars -> Python numbers, tuples -> lists, unknown objects -> str."""
    if obj is None or isinstance(obj, (bool, str)):
        return obj
    if isinstance(obj, int):
        return obj
    if isinstance(obj, float):
        return obj if math.isfinite(obj) else None
    if isinstance(obj, dict):
        return {str(k): sanitize(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [sanitize(v) for v in obj]
    if hasattr(obj, "item"):            # numpy / torch scalars
        try:
            return sanitize(obj.item())
        except Exception:
            pass
    try:
        return sanitize(float(obj))
    except Exception:
        return str(obj)


def compact(obj, depth=0):
    """The result line's form of a (sanitized) record: prose keys dropped below the top level, long strings cut, floats below the
    top level to 6 significant digits (the contract's own top-level numbers keep every digit)."""
    if isinstance(obj, dict):
        return {k: compact(v, depth + 1) for k, v in obj.items()
                if not (depth >= 1 and k in PROSE_KEYS) and not (depth >= 2 and k in DEEP_PROSE_KEYS and isinstance(v, str)
                                                                 and not _is_roofline_entry(obj))}
    if isinstance(obj, list):
        return [compact(v, depth + 1) for v in obj]
    if isinstance(obj, str) and len(obj) > MAX_STR:
        return obj[:MAX_STR - 3] + "..."
    if isinstance(obj, float) and depth >= 2:
        return float(f"{obj:.6g}")
    return obj


def _is_roofline_entry(d):
    """roofline_hbm.<row> keeps its kernel name: the judge reads the row by it."""
    return "achieved" in d and "bytes_per_launch" in d


def _prefixed(tag, obj):
    text = f"[{tag}] " + json.dumps(sanitize(obj))
    assert not _parses(text)
    return text


def _parses(text):
    try:
        json.loads(text)
        return True
    except ValueError:
        return False


def rank_note(tag, obj, stream=None):
    """One line of a rank's own record (any rank): ``[tag] {json}`` on stderr -- readable, greppable, never the result line."""
    print(_prefixed(tag, obj), file=stream or sys.stderr, flush=True)


def detail_note(obj, root=None, stream=None):
    """The long form of the result (every note, every leg) in front of the result line, and as a file when the tree is writable."""
    print(_prefixed("bench_detail", obj), file=stream or sys.stdout, flush=True)
    if root is not None:
        try:
            d = os.path.join(root, "gpurun_out")
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "bench_detail.json"), "w") as fh:
                json.dump(sanitize(obj), fh, indent=1)
        except OSError:
            pass


def result_text(line, provisional=False, required=REQUIRED):
    """The one bare-JSON line: compact, NaN-free, with every key of the contract present."""
    rec = compact(sanitize(line))
    if provisional:
        rec["provisional"] = True
    missing = [k for k in required if k not in rec]
    if missing:
        raise KeyError(f"bench result line lacks {missing}")
    return json.dumps(rec, allow_nan=False)


def emit_result(line, provisional=False, stream=None, required=REQUIRED):
    """Print the result line (rank 0 only calls this).  stderr is flushed first so that, in a merged capture, nothing of this
    process trails the line; after the FINAL call nothing else may be printed."""
    sys.stderr.flush()
    out = stream or sys.stdout
    print(result_text(line, provisional, required), file=out, flush=True)


def last_json_line(text):
    """The driver's view: the last line of a captured run (stdout, then stderr) that is a JSON object, parsed; None if there is none."""
    for ln in reversed(text.splitlines()):
        ln = ln.strip()
        if not ln.startswith("{"):
            continue
        try:
            obj = json.loads(ln)
        except ValueError:
            continue
        if isinstance(obj, dict):
            return obj
    return None
