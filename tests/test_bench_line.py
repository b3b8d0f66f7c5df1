"""bench.py's output contract: the LAST stdout line is one compact strict-JSON object of at most 4 KB that
carries the contract's keys, `roofline` and `cpu_baseline`; the long per-workload records go on earlier lines.  Checked on
recorded results of real runs (profiles/*_bench_default.json: the long form the bench produced on the MI355X)."""
import glob
import json
import os

import pytest

from benchmarks import report

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDED = [p for p in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_bench_default.json")) +
                              glob.glob(os.path.join(REPO, "profiles", "r*_bench_2rank_same_device.json")))
            if os.path.basename(p) >= "r05"]            # (earlier rounds' records predate `exchange` / per-slice parity)
DEFAULT = [p for p in RECORDED if p.endswith("_bench_default.json")]


def _load(path):
    """A recorded run: either the old one-line form or the new detail lines ({"detail": name, "record": {...}} ... compact)."""
    rows = [json.loads(ln) for ln in open(path).read().splitlines() if ln.strip().startswith("{")]
    details = [r for r in rows if "detail" in r and "record" in r]
    if details:
        head = dict(next(r["record"] for r in details if r["detail"] == "headline"))
        head["workloads"] = {r["detail"]: r["record"] for r in details if r["detail"] != "headline"}
        return head
    return max(rows, key=lambda r: len(json.dumps(r)))


def _strict(s):
    def bad(c):
        raise ValueError("non-strict JSON constant " + c)
    return json.loads(s, parse_constant=bad)


@pytest.mark.parametrize("path", RECORDED, ids=[os.path.basename(p) for p in RECORDED])
def test_final_line_is_compact_strict_json(path):
    res = _load(path)
    line = report.compact_line(res)
    assert "\n" not in line and len(line.encode()) < report.LINE_LIMIT
    out = _strict(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["config"]["workload"] and "model" not in out["config"] and out["config"]["value_definition"]
    roof = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] in ("hbm", "mfma") and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    if res.get("cpu_baseline"):
        cpu = out["cpu_baseline"]
        assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"]
    assert abs(out["value"] - res["value"]) <= 1e-5 * abs(res["value"])
    for name, rec in (out.get("workloads") or {}).items():
        assert len(json.dumps({name: rec}, separators=(",", ":"))) <= report.SLICE_LIMIT, name
        assert "err" in rec or {"kms", "frac"} <= set(rec), (name, rec)
        assert "err" in rec or out["n_gpus"] > 1 or rec.get("par") == "ok", (name, rec)
    if out["n_gpus"] > 1:
        assert out["ranks"]["ranks_seen"] == out["n_gpus"] and "cross_check" in out["ranks"]
        assert {"row_bytes", "exchange_ms", "hidden_ms"} <= set(out["exchange"])


def test_detail_lines_cannot_be_mistaken_for_the_line():
    res = _load(DEFAULT[-1])
    for ln in report.detail_lines(res):
        rec = _strict(ln)
        assert set(rec) == {"detail", "record"}


def test_non_finite_numbers_do_not_reach_the_line():
    res = _load(DEFAULT[-1])
    res["roofline"]["traffic_frac"] = float("nan")
    res["value_roots4096"] = float("inf")
    res["workloads"] = dict(res.get("workloads") or {}, broken=dict(error="RuntimeError: " + "x" * 500))
    out = _strict(report.compact_line(res))
    assert out["roofline"]["traffic_frac"] is None and "value_roots4096" not in out or out["value_roots4096"] is None
    assert len(out["workloads"]["broken"]["err"]) <= 80


def test_line_degrades_instead_of_growing():
    res = _load(DEFAULT[-1])
    base = dict(next(iter(res["workloads"].values())))
    res["workloads"] = dict(res["workloads"], **{"extra_slice_number_{}".format(i): base for i in range(12)})
    line = report.compact_line(res)
    assert len(line) <= report.LINE_LIMIT
    assert len(_strict(line)["workloads"]) == len(res["workloads"])
