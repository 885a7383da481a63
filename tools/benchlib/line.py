"""bench.py: the ONE JSON line the driver parses, cut from the run's full report.

The full report (every leg, every roofline object, notes) goes to a side file (gpurun_out/bench_extras.json); the line printed last on stdout
carries only what the bench contract names, in a fixed key set, and stays far below 4 KB (tests/test_benchlib_cpu.py builds it from a canned
report and checks size, round trip and keys)."""
import json

LINE_LIMIT = 4096

TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline", "s_scene_frac", "decode_kernel_frac", "throughput_mode_value", "sustained", "sharded", "verify_ok", "extras_file")
CONFIG_KEYS = ("workload", "pipeline", "scene_name", "rig", "executed_path", "valid_fraction", "extras")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "frac_mean", "frac_s_scene", "isolated_frac", "traffic", "traffic_over_algorithmic", "traffic_over_algorithmic_s_scene",
                 "headline_scene", "traffic_source", "kernel", "avg_launch_ms",
                 "median_launch_ms", "launch_ms_used", "max_launch_ms", "launches_timed", "outliers", "algorithmic_bytes_per_launch")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "reference_measured", "why_port_differs", "c_oracle_value", "c_oracle_all_cores_value", "c_oracle_all_cores")
SHARDED_KEYS = ("rccl_nranks", "distinct_devices", "rank_devices", "exchange", "exchange_impl", "wire", "overlap", "with_exchange_value", "compute_only_value")


def _cut(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 1] + "…"


def _pick(d, keys):
    return {k: d.get(k) for k in keys}


def compact_line(report, extras_file=None):
    """report = run.py's full dict -> the dict of the printed line (fixed keys, short strings)."""
    cfg = report.get("config") or {}
    line = {k: report.get(k) for k in TOP_KEYS[:12]}
    line["config"] = {"workload": _cut(cfg.get("workload", ""), 120), "pipeline": _cut(cfg.get("pipeline", ""), 100), "scene_name": cfg.get("scene_name"),
                      "rig": cfg.get("rig"), "executed_path": (cfg.get("executed") or {}).get("path"), "valid_fraction": cfg.get("valid_fraction"),
                      "extras": cfg.get("extras")}
    roof = report.get("roofline")
    if roof:
        r = _pick(roof, ROOFLINE_KEYS)
        r["kernel"] = _cut(r.get("kernel") or "", 100)
        src = roof.get("traffic_source")
        r["traffic_source"] = None if not src else ("live rocprofv3 --pmc children of this run" if str(src).startswith("this run") else
                                                      "committed constant (profiles/traffic.json)" if str(src).startswith("committed") else _cut(src, 60))
        line["roofline"] = r
    else:
        line["roofline"] = None
    cpu = report.get("cpu_baseline")
    if cpu:
        c = _pick(cpu, CPU_KEYS)
        c["sample"] = _cut(c.get("sample") or "", 160)
        rm = cpu.get("reference_measured")
        c["reference_measured"] = rm.get("value") if isinstance(rm, dict) else rm          # Mpixels/s: the reference itself, survey container
        c["why_port_differs"] = _cut(c.get("why_port_differs") or "", 260)
        line["cpu_baseline"] = c
    else:
        line["cpu_baseline"] = None
    scenes = report.get("scenes") or {}
    line["s_scene_frac"] = (scenes.get("s-scene") or {}).get("frac")
    dk = report.get("decode_kernel_alone") or report.get("decode_kernel_headline")
    line["decode_kernel_frac"] = ((dk or {}).get("roofline") or {}).get("frac")
    thr = report.get("throughput_mode") or {}
    line["throughput_mode_value"] = (thr.get("batched") or {}).get("value") or thr.get("value")
    su = report.get("sustained")
    line["sustained"] = None if not su else {"value": su.get("value"), "seconds": su.get("seconds"), "sclk_mhz_mean": (su.get("gpu") or {}).get("sclk_mhz_mean"),
                                             "gpu_busy_percent_mean": (su.get("gpu") or {}).get("gpu_busy_percent_mean")}
    sh = report.get("sharded")
    line["sharded"] = _pick(sh, SHARDED_KEYS) if sh else None
    if sh and isinstance(sh.get("rank_devices"), list):   # PCI bus ids without the domain ("75:00.0"), at most 16 ranks: 8 GPUs stay under 100 bytes
        line["sharded"]["rank_devices"] = [_cut(str(d).split(":", 1)[-1] if str(d).count(":") == 2 else d, 12) for d in sh["rank_devices"][:16]]
    alts = report.get("sharded_alternatives")
    if sh and isinstance(alts, dict):                     # the other exchange forms timed after the counted region: label -> Mpixels/s (or "error")
        line["sharded"]["alternatives"] = {_cut(k, 24): (v.get("value") if isinstance(v, dict) and "value" in v else "error") for k, v in list(alts.items())[:6]}
    ver = report.get("verify")
    line["verify_ok"] = None if ver is None else bool(ver.get("ok"))
    line["extras_file"] = extras_file
    if report.get("error"):
        line["error"] = _cut(report["error"], 200)
    return line


def dump_line(report, extras_file=None):
    s = json.dumps(compact_line(report, extras_file), ensure_ascii=True)
    if len(s) >= LINE_LIMIT:                         # cannot happen with the cuts above; if it ever does, the contract's keys survive
        d = compact_line(report, extras_file)
        for k in ("sharded", "cpu_baseline"):
            if isinstance(d.get(k), dict):
                d[k] = {kk: vv for kk, vv in d[k].items() if not isinstance(vv, str) or len(vv) < 40}
        s = json.dumps(d, ensure_ascii=True)
    return s
