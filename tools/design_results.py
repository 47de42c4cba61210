#!/usr/bin/env python3
"""Fills DESIGN.md's results table (between the RESULTS markers) from a committed bench line:
   python tools/design_results.py profiles/r5/bench_line.json"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(sys.argv[1]))
r, s, c = d["roofline"], d.get("sustained", {}), d["configs"]
oc = d.get("overflow_check", {})


def g(x, *ks, default=None):
    for k in ks:
        if not isinstance(x, dict) or k not in x:
            return default
        x = x[k]
    return x


def f(v, n=1):
    return "n/a" if v is None else f"{v:.{n}f}"


rows = [
    ("**value** (driver form: 20 steps after 5 warm-up + the disclosed pre-warm)", f"**{f(d['value'])} pairs/s**, {f(d['ms_per_step'], 3)} ms/step"),
    ("`sustained` (≥ 5 s, ≥ 2000 steps, warm part)", f"{f(s.get('pairs_per_s'))} pairs/s; first / last 100 steps {f(s.get('first_100_ms_per_step'), 3)} / {f(s.get('last_100_ms_per_step'), 3)} ms; die clock {f(g(s, 'sclk_mhz', 'mean'), 0)} MHz"),
    ("`parity_vs_cpu_baseline` (all 5.75 M outputs of the timed run vs the oracle)", f"{g(d, 'parity_vs_cpu_baseline', 'rel_err'):.2e}"),
    ("`roofline` conv2 (`conv3d_x3q_kernel<side head>`)", f"{f(r['avg_launch_ms'], 3)} ms; necessary 3× flops {f(r['achieved'], 0)} TFLOP/s = **{f(r['frac'], 3)}** of 2.5 PF (executed {f(r.get('executed_frac'), 3)}); "
     f"HBM traffic {f((r.get('traffic') or 0) / 1e9, 2)} GB/launch; all-zero operands {f(g(r, 'power_probe', 'zeros_operands_ms'), 3)} ms vs random {f(g(r, 'power_probe', 'random_operands_ms'), 3)} (power)"),
    ("`roofline_hbm` expand pass", f"{f(g(d, 'roofline_hbm', 'avg_launch_ms'), 3)} ms = {f(g(d, 'roofline_hbm', 'achieved'), 0)} GB/s = {f(g(d, 'roofline_hbm', 'frac'), 3)} of 8 TB/s; prep {f(g(d, 'roofline_hbm', 'prep_ms'), 3)} ms"),
    ("`overflow_check` cost (checked − deferred)", f"{f(oc.get('cost_ms_per_step'), 4)} ms/step"),
    ("`two_launch_tail` (r4's tail)", f"{f(g(d, 'two_launch_tail', 'value'))} pairs/s"),
    ("`general_shift` (any shift array)", f"{f(g(d, 'general_shift', 'value'))} pairs/s, {f(g(d, 'general_shift', 'ms_per_step'), 3)} ms"),
    ("`reference_api` (the reference's two calls verbatim)", f"{f(g(d, 'reference_api', 'value'))} pairs/s"),
    ("`fp32_mfma` / `built_right_half` / `materialized`", f"{f(g(d, 'fp32_mfma', 'value'))} / {f(g(d, 'built_right_half', 'value'))} / {f(g(d, 'materialized', 'value'))} pairs/s"),
    ("`train` (cfg4; r6: forward, data and weight gradients of conv2 + hourglass on split f16x3 operands; the default run, other legs before it)", f"{f(g(d, 'train', 'ms_per_step'), 2)} ms/step (fwd {f(g(d, 'train', 'fwd_ms'), 2)}, bwd {f(g(d, 'train', 'bwd_ms'), 2)})"),
    ("`configs.cfg4_train_step` weight gradient 32->32 full grid", f"split-operand form {f(g(c, 'cfg4_train_step', 'wgrad_ms'), 3)} ms (finding its own maxima: {f(g(c, 'cfg4_train_step', 'wgrad_ms_finding_its_own_maxima'), 3)}), fp32 Winograd form {f(g(c, 'cfg4_train_step', 'wgrad_fp32_form_ms'), 3)} ms"),
    ("`value_fp32_mfma` / `dtype`", f"{f(d.get('value_fp32_mfma'))} pairs/s on the fp32-MFMA kernels; dtype = {d.get('dtype')}"),
]
for name, e in c.items():
    if not isinstance(e, dict):
        continue
    rate = e.get("crops_per_s") or e.get("rois_per_s") or e.get("value")
    par = g(e, "parity_vs_cpu_baseline", "rel_err")
    if par is None and isinstance(e.get("parity_vs_cpu_baseline"), dict):      # local legs: worst of the two outputs
        errs = [v["rel_err"] for v in e["parity_vs_cpu_baseline"].values() if isinstance(v, dict) and "rel_err" in v]
        par = max(errs) if errs else None
    if rate is None and "ms_per_step" in e:
        rate = 1e3 / e["ms_per_step"]
    cpu = g(e, "cpu_baseline", "value")
    extra = []
    if name == "cfg4_train_step":
        continue
    for k in ("crops_per_s_f16", "f16_crops_per_s", "rois_per_s_f32", "f32_rois_per_s"):
        if k in e:
            extra.append(f"{k} {f(e[k])}")
    rows.append((f"`configs.{name}`", f"{f(rate)} /s" + (f"; parity {par:.1e}" if par is not None else "") + (f"; CPU oracle {cpu:.3g} /s" if cpu is not None else "")
                 + ("; " + ", ".join(extra) if extra else "")))
for key, label in (("warped_expand", "any-shift expand"), ("full_volume", "a1 cost-volume builder"), ("right_half_builder", "right-half builder"),
                   ("gather", "a3 gather, projected coordinates"), ("gather_uniform", "a3 gather, uniform coordinates"),
                   ("cost_volume_backward", "a2 cost-volume backward"), ("roiaware_pool3d", "a10 roiaware_pool3d")):
    e = g(d, "roofline_hbm", key)
    if isinstance(e, dict) and e.get("frac") is not None:
        rows.append((f"`roofline_hbm.{key}` ({label})", f"{f(e.get('avg_launch_ms'), 3)} ms = {f(e.get('achieved'), 0)} GB/s = {f(e.get('frac'), 3)} of 8 TB/s"))
cb = d["cpu_baseline"]
rows.append(("`cpu_baseline`", f"{cb['value']:.3g} {cb['unit']} on {cb['cores']} threads ({cb['kind']}); {cb.get('sample', '')[:110]}"))
off = d.get("off_fast_path", {})
if off:
    bits = []
    for k, e in off.items():
        if isinstance(e, dict):
            v = e.get("crops_per_s") or e.get("pairs_per_s")
            if v:
                bits.append(f"{k} {f(v)} /s")
    rows.append(("`off_fast_path`", "; ".join(bits)))
rnd = os.path.basename(os.path.dirname(os.path.abspath(sys.argv[1])))
table = f"| Entry of the line | {rnd} |\n|---|---|\n" + "\n".join(f"| {a} | {b} |" for a, b in rows)
path = os.path.join(ROOT, "DESIGN.md")
text = open(path).read()
block = f"<!-- RESULTS:BEGIN ({os.path.relpath(sys.argv[1], ROOT)}) -->\n{table}\n<!-- RESULTS:END -->"
if "RESULTS_TABLE" in text:
    text = text.replace("RESULTS_TABLE", block, 1)
else:
    text = re.sub(r"<!-- RESULTS:BEGIN.*?<!-- RESULTS:END -->", lambda m: block, text, flags=re.S)
open(path, "w").write(text)
print(table)
