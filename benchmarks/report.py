"""The bench's output contract: ONE compact final JSON line (the driver parses the last stdout line and keeps an 8 KB tail),
everything longer on earlier stdout lines (one short-keyed JSON object per workload) and in a side file.

`compact_line(res)` -> str, at most LINE_LIMIT bytes, strict JSON (no NaN / Infinity), carrying the contract's keys
(metric ... config, roofline, cpu_baseline) plus a <= SLICE_LIMIT-byte summary per workload slice.  `detail_lines(res)` ->
the long form, one line per workload, printed BEFORE the compact line.  tests/test_bench_line.py pins both on a recorded run.
"""
import json
import math

LINE_LIMIT = 4096          # bytes of the final line (the r05 line was 22 KB and the driver could not parse it)
SLICE_LIMIT = 150          # bytes per slice summary, key included


def sig(x, digits=4):
    """x rounded to `digits` significant digits (ints stay ints, non-finite floats become None: strict JSON)."""
    if isinstance(x, bool) or x is None or isinstance(x, (str, int)):
        return x
    try:
        x = float(x)
    except (TypeError, ValueError):
        return None
    if not math.isfinite(x):
        return None
    if x == 0.0:
        return 0.0
    y = float("{:.{}g}".format(x, digits))
    return int(y) if abs(y) >= 10 ** digits and y == int(y) else y


def _clean(o):
    """Strict-JSON form of a result tree: non-finite floats -> None, numpy scalars -> python."""
    if isinstance(o, dict):
        return {str(k): _clean(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_clean(v) for v in o]
    if isinstance(o, float):
        return o if math.isfinite(o) else None
    if hasattr(o, "item") and not isinstance(o, (str, bytes)):
        try:
            return _clean(o.item())
        except (ValueError, AttributeError):
            return str(o)
    return o


def _short(s, n):
    s = "" if s is None else str(s)
    return s if len(s) <= n else s[:n - 1] + "~"


def _pick(d, keys):
    return {k: d[k] for k in keys if k in d and d[k] is not None}


def slice_summary(rec):
    """<= SLICE_LIMIT bytes: value, ms per step, dominant kernel's time, the two roofline fractions, oracle replay."""
    if "error" in rec:
        return dict(err=_short(rec["error"], 80))
    out = dict(v=sig(rec.get("value")), ms=sig(rec.get("ms_per_step")), kms=sig(rec.get("kernel_ms")),
               frac=sig(rec.get("frac"), 3), tfrac=sig(rec.get("traffic_frac"), 3), par=rec.get("parity_sample"))
    out = {k: v for k, v in out.items() if v is not None}      # (an absent tfrac = no committed counter pass for that launch)
    if rec.get("ranks_seen") is not None:                      # N > 1: the sharded slices (C4 / C5)
        out["rk"] = rec["ranks_seen"]
        out["xc"] = rec.get("cross_check")
        if rec.get("all_gather_ms") is not None:
            out["ag_ms"] = sig(rec["all_gather_ms"], 3)
    return out


def compact(res):
    """The dict of the final line."""
    res = _clean(res)
    cfg, roof = res.get("config") or {}, res.get("roofline") or {}
    out = {k: res.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                   "scaling", "vs_baseline", "dtype")}
    out["value"], out["ms_per_step"] = sig(out["value"], 6), sig(out["ms_per_step"], 6)
    out["data"] = _short(res.get("data"), 80)
    c = _pick(cfg, ("workload", "n_roots_per_gpu", "n_roots_total", "states", "actions", "episodes", "horizon", "gamma",
                    "sweeps", "mdps", "planners", "budget"))
    for k in ("env_steps_per_step", "algorithmic_bytes_per_env_step", "measured_mean_selection_depth",
              "measured_expansions_per_episode"):
        if cfg.get(k) is not None:
            c[k] = sig(cfg[k], 6)
    c["value_definition"] = _short(cfg.get("value_definition_short") or "device-resident: inputs and results in HBM; "
                                   "host-inclusive form = value_host_inclusive", 120)
    c["parallelism"] = _short(cfg.get("parallelism_short") or cfg.get("parallelism"), 64)
    out["config"] = c
    r = _pick(roof, ("bound", "unit", "kernel", "kernel_variant"))
    r["kernel"] = _short(r.get("kernel"), 60)
    for k in ("achieved", "peak", "frac", "traffic", "traffic_frac", "kernel_ms", "algorithmic_bytes_per_launch", "frac_hbm_side",
              "mfma_frac_of_f64_peak"):
        if k in roof:
            r[k] = sig(roof[k], 5)
    out["roofline"] = r
    cpu = res.get("cpu_baseline")
    if isinstance(cpu, dict):
        b = dict(value=sig(cpu.get("value"), 5), unit=cpu.get("unit"), cores=cpu.get("cores"), kind=cpu.get("kind"),
                 sample=_short(cpu.get("sample"), 90))
        if cpu.get("value_1core") is not None:
            b["value_1core"] = sig(cpu["value_1core"], 5)
        ref = cpu.get("reference_python")
        if isinstance(ref, dict):
            rp = {k: sig(v, 5) for k, v in ref.items() if isinstance(v, (int, float)) and not isinstance(v, bool)}
            rp["where"] = "build container, tests/golden/gen/time_reference.py"
            b["reference_python"] = rp
        out["cpu_baseline"] = b
    else:
        out["cpu_baseline"] = None
    for k in ("value_host_inclusive", "host_inclusive_ms_per_step", "value_roots4096", "value_roots4096_host_inclusive"):
        if res.get(k) is not None:
            out[k] = sig(res[k], 5)
    per_root = res.get("plan_wall_ms_per_root")
    if isinstance(per_root, dict):
        out["plan_wall_ms_per_root"] = {k: sig(v, 4) for k, v in per_root.items()}
    par = res.get("parity_sample")
    out["parity_sample"] = par.get("result") if isinstance(par, dict) else par
    ranks = res.get("ranks")
    if isinstance(ranks, dict):
        out["ranks"] = _pick(ranks, ("ranks_seen", "backend", "distinct_devices", "cross_check", "cross_check_roots",
                                      "cross_check_of_rank", "dry_run_same_device"))
    ex = res.get("exchange")
    if isinstance(ex, dict):
        out["exchange"] = {k: sig(ex[k], 4) for k in ("row_bytes", "bytes_sent_per_rank_per_step", "exchange_ms", "hidden_ms",
                                                      "kernel_ms", "step_ms", "on_side_stream", "backend") if k in ex}
    gm = res.get("general_model_kernel")
    if isinstance(gm, dict):
        out["general_model_kernel"] = dict(kernel_variant=gm.get("kernel_variant"), kernel_ms=sig(gm.get("kernel_ms")),
                                           value=sig(gm.get("value")))
    w = res.get("workloads")
    if isinstance(w, dict):
        out["workloads"] = {name: slice_summary(rec) for name, rec in w.items()}
        out["workloads_key"] = "v=value ms=ms/step kms=kernel ms frac=algorithmic/peak tfrac=PMC bytes/peak par=oracle replay"
    out["detail"] = "earlier stdout lines (one JSON object per workload) and bench_detail.json"
    return out


def compact_line(res):
    """The final stdout line.  If it would exceed LINE_LIMIT, the least important blocks go (they stay in the detail lines)."""
    out = compact(res)
    for drop in (None, "general_model_kernel", "workloads_key", "plan_wall_ms_per_root"):
        if drop is not None:
            out.pop(drop, None)
        line = json.dumps(out, separators=(",", ":"), allow_nan=False)
        if len(line) <= LINE_LIMIT:
            return line
    # still too long: shorten the slice summaries to the three judged numbers
    if "workloads" in out:
        out["workloads"] = {k: {kk: v[kk] for kk in ("kms", "frac", "tfrac", "par", "err") if kk in v}
                            for k, v in out["workloads"].items()}
    line = json.dumps(out, separators=(",", ":"), allow_nan=False)
    if len(line) > LINE_LIMIT:
        raise ValueError("bench line is {} bytes (> {})".format(len(line), LINE_LIMIT))
    return line


def detail_lines(res):
    """The long form: the headline without its slices, then one line per slice; each `{"detail": name, "record": {...}}`."""
    res = _clean(res)
    head = {k: v for k, v in res.items() if k != "workloads"}
    # (wrapped under "record": a detail line has no top-level "metric" / "value", so nothing can mistake it for THE line)
    lines = [json.dumps(dict(detail="headline", record=head), allow_nan=False)]
    for name, rec in (res.get("workloads") or {}).items():
        lines.append(json.dumps(dict(detail=name, record=rec), allow_nan=False))
    return lines
