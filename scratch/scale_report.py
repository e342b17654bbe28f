"""Digest of bench.py lines from multi-GPU runs (the driver's SCALE_rNN.json, or any file with bench JSON lines):
    python scratch/scale_report.py SCALE_r01.json
For every line: the headline, every schedule the tuning pass timed, the step trace of rank 0 (time waiting for panels
against time updating, the slowest steps), the link probe, the partitioned C5 leg and the replicas leg."""
import json, sys


def lines_of(path):
    txt = open(path).read()
    out = []
    try:
        doc = json.loads(txt)
        stack = [doc]
        while stack:
            v = stack.pop()
            if isinstance(v, dict):
                if "metric" in v and "value" in v:
                    out.append(v)
                else:
                    stack.extend(v.values())
            elif isinstance(v, list):
                stack.extend(v)
            elif isinstance(v, str) and v.lstrip().startswith("{") and '"metric"' in v:
                try:
                    out.append(json.loads(v))
                except ValueError:
                    pass
    except ValueError:
        for l in txt.splitlines():
            l = l.strip()
            if l.startswith("{") and '"metric"' in l:
                try:
                    out.append(json.loads(l))
                except ValueError:
                    pass
    return sorted(out, key=lambda d: d.get("n_gpus", 0))


for d in lines_of(sys.argv[1]):
    n = d.get("n_gpus")
    print("=" * 100)
    print("n_gpus %s  %s" % (n, d["config"]["workload"]))
    print("  value %.0f GFLOP/s = %.1f %% of %s x 78.6 TFLOP/s   %.2f ms per evaluation   [%s]%s" % (
        d["value"], d.get("pct_fp64_mfma_peak", float("nan")), n, d["ms_per_step"], d["config"].get("parallelism"),
        "   WATCHDOG: " + d["watchdog"] if d.get("watchdog") else ""))
    ref = d.get("single_gpu_same_workload")
    if ref:
        print("  same workload on one GPU: %.2f ms -> speed-up %.2fx (efficiency %.0f %%)" % (
            ref["ms_per_step"], ref["ms_per_step"] / d["ms_per_step"], 100 * ref["ms_per_step"] / d["ms_per_step"] / n))
    for k, v in sorted((d.get("schedules_ms") or {}).items(), key=lambda kv: kv[1]):
        print("    %-44s %9.2f ms" % (k, v))
    for k, v in (d.get("schedules_failed") or {}).items():
        print("    FAILED %-37s %s" % (k, v))
    tr = d.get("trace")
    if tr and tr.get("arrived_ms"):
        arr, app = tr["arrived_ms"], tr["applied_ms"]
        print("  rank 0: first panel at %.2f ms, waiting for panels %.2f ms, updating %.2f ms, end %.2f ms" % (
            tr["first_panel_ms"], tr["waiting_for_panels_ms"], tr["updating_ms"], tr["end_ms"]))
        step = [(arr[k + 1] - arr[k], k) for k in range(len(arr) - 1)]
        waits = [(arr[k + 1] - app[k], k) for k in range(min(len(app), len(arr) - 1))]
        q = len(step) // 4 or 1
        for name, lo, hi in (("first quarter", 0, q), ("second", q, 2 * q), ("third", 2 * q, 3 * q), ("last", 3 * q, len(step))):
            seg = step[lo:hi]
            wseg = waits[lo:hi]
            if seg:
                print("    steps %-13s: %.3f ms per step on average, of which %.3f ms waiting for the next panel" % (
                    name, sum(s for s, _ in seg) / len(seg), sum(max(w, 0.0) for w, _ in wseg) / max(len(wseg), 1)))
    cp = d.get("comm_probe")
    if cp:
        print("  link probe (max over ranks):")
        for k in sorted(cp, key=lambda s: (s.split("_")[-1], s)):
            v = cp[k]
            if isinstance(v, dict):
                print("    %-28s %8.3f ms  %7.1f GB/s" % (k, v["ms"], v["GBps"]))
            else:
                print("    %-28s %8.3f ms" % (k, v))
    for leg in ("c5_partitioned", "replicas_c3"):
        v = d.get(leg)
        if v and "error" not in v:
            print("  %s: %s" % (leg, {k: (round(x, 2) if isinstance(x, float) else x) for k, x in v.items() if k != "note"}))
        elif v:
            print("  %s: %s" % (leg, v))
    if d.get("probe_error"):
        print("  probe_error:", d["probe_error"])
