"""bench.py's counter bookkeeping (no GPU): which launches of a rocprofv3 counter_collection CSV feed which
roofline object.  The DNA walk kernel runs two shapes under one name (config 3 / config 3b: told apart by
grid), and a kernel's one-off launch on a small side batch must not dilute its per-launch mean."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)


def row(name, grid, counter, value):
    return {"Kernel_Name": name, "Grid_Size": str(grid), "Counter_Name": counter, "Counter_Value": str(value)}


WALK = "void fmx_locate_f3u_kernel<4, true>(HIP_vector_type<unsigned int, 4u> const*, ...)"
COUNT = "void fmx_count_f3_kernel<1, false, false>(HIP_vector_type<unsigned int, 4u> const*, ...)"
EPC = "void fmx_count_ep_kernel<1, 2, 2, false>(FmxDev, ...)"


def test_walk_kernel_is_split_by_grid_and_small_launches_are_left_out():
    rows = ([row(COUNT, 524288, "FETCH_SIZE", 2012500.0 + i) for i in range(3)]
            + [row(COUNT, 524288, "FETCH_SIZE", 41000.0)]            # the count that prepares config 3b
            + [row(WALK, 262144, "FETCH_SIZE", 263400.0 + i) for i in range(3)]
            + [row(WALK, 524288, "FETCH_SIZE", 27.8e6 + i) for i in range(2)]
            + [row(EPC, 262144, "FETCH_SIZE", 4.25e6)] * 3
            + [row(COUNT, 524288, "WRITE_SIZE", 1.0)])                # another counter: ignored
    agg = B.pmc_aggregate(rows, "FETCH_SIZE")
    kn, v = B.pmc_per_dispatch(agg, B.PMC_LEGS["dna_count"])
    assert "fmx_count_f3_kernel" in kn and abs(v - 2012501.0) < 1.0          # the small launch is not averaged in
    kn, v = B.pmc_per_dispatch(agg, B.PMC_LEGS["dna_locate"], B.PMC_WHICH["dna_locate"])
    assert kn.endswith("@grid 262144") and abs(v - 263401.0) < 1.0
    kn, v = B.pmc_per_dispatch(agg, B.PMC_LEGS["dna_locate_3b"], B.PMC_WHICH["dna_locate_3b"])
    assert kn.endswith("@grid 524288") and abs(v - (27.8e6 + 0.5)) < 1.0
    kn, v = B.pmc_per_dispatch(agg, B.PMC_LEGS["rlfm_count"])
    assert "fmx_count_ep_kernel" in kn and v == 4.25e6
    assert B.pmc_per_dispatch(agg, B.PMC_LEGS["rlfm_locate"]) == (None, None)   # kernel not in the trace


def test_config_3b_needs_a_second_grid():
    rows = [row(WALK, 262144, "FETCH_SIZE", 263400.0)] * 3
    agg = B.pmc_aggregate(rows, "FETCH_SIZE")
    assert B.pmc_per_dispatch(agg, [B.WALK_KERNEL], "grid_min")[1] == 263400.0
    assert B.pmc_per_dispatch(agg, [B.WALK_KERNEL], "grid_max") == (None, None)   # --no-3b: nothing to report


def test_roofline_fraction_never_exceeds_the_traffic_it_was_given():
    r = B.make_roofline("k", 1.0, 10, 384, 1000, None, {"rdreq": {"all": 31250000.0, "32B": 0.0, "64B": 0.0, "128B": 31250000.0},
                                                        "source": "test"})
    assert r["achieved"] == 4000.0 and r["frac"] == 0.5 and r["traffic"] == 4_000_000_000
    r = B.make_roofline("k", 1.0, 10, 384, 1000, {"requested_lines": 10**9}, None)
    assert r["frac"] is None and r["achieved"] is None          # requested lines alone are never a roofline fraction


def test_counters_are_priced_by_request_width():
    """ONE basis for every roofline object: the read requests counted by width, priced at their
    widths, + WRITE_SIZE.  The census share of rounds 2-5 is gone: a launch of lane-wise 16-byte probes whose fabric
    requests are all 128 bytes wide moves 128 bytes per request."""
    w128 = {"rdreq": {"all": 1000000.0, "32B": 0.0, "64B": 0.0, "128B": 1000000.0}, "write_kb": 1000.0, "source": "test"}
    half_probes = {"requested_lines": 100, "requested_records": 50, "requested_probes": 50, "distinct_lines": 50}
    r = B.make_roofline("k", 1.0, 10, 384, 0, half_probes, w128)
    assert r["traffic"] == int(128 * 1000000.0 + 1000.0 * 1024)          # whatever the census says about probes
    assert r["share_of_128B_requests"] == 1.0 and "frac_upper" not in r and "traffic_upper" not in r
    assert r["achieved"] == round(r["traffic"] / 1e-3 / 1e9, 1) and r["frac"] == round(r["achieved"] / 8000.0, 4)
    assert r["fetch_kb_raw"] == 1000000.0 * 64 / 1024                    # what FETCH_SIZE would have said
    assert r["frac_of_gather_ceiling"] == round(1000000.0 / 1e-3 / 55e9, 4)
    assert r["requested_bytes"] == 50 * 128 + 50 * 16                    # (the census still describes the requests made)
    mixed = {"rdreq": {"all": 1000.0, "32B": 100.0, "64B": 300.0, "128B": 500.0}, "write_kb": 0.0, "source": "test"}
    r = B.make_roofline("k", 1.0, 10, 384, 0, None, mixed)
    assert r["traffic"] == 32 * 100 + 64 * (300 + 100) + 128 * 500        # 100 requests of no counted width: 64 B each
    assert r["share_of_128B_requests"] == 0.5
    # the figures of profiles/r05/kernel_pmc_bytes_rlfm.json (config 4 count): 0.82 of the fabric peak, not 0.61
    rl = {"rdreq": {"all": 68018987.0, "32B": 0.0, "64B": 4075.0, "128B": 68014912.0}, "write_kb": 24576.0, "source": "r05"}
    r = B.make_roofline("fmx_count_ep_kernel", 1.334, 1 << 24, 2560, 0, None, rl)
    assert 0.815 <= r["frac"] <= 0.822, r["frac"]
    # an entry of an earlier round (FETCH_SIZE only): the guide's rule, and the basis says so
    old = {"fetch_kb_raw": 1000000.0, "write_kb": 1000.0, "source": "test"}
    r = B.make_roofline("k", 1.0, 10, 384, 0, None, old)
    assert r["traffic"] == int(2 * 1000000.0 * 1024 + 1000.0 * 1024) and "2 x FETCH_SIZE" in r["basis"]


def test_the_width_counters_of_a_leg_come_from_the_same_launches():
    names = ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"]
    vals = {"TCC_EA0_RDREQ_sum": 32.2e6, "TCC_EA0_RDREQ_32B_sum": 0.0, "TCC_EA0_RDREQ_64B_sum": 1200.0,
            "TCC_EA0_RDREQ_128B_sum": 32.2e6 - 1200.0}
    rows = [row(COUNT, 524288, nm, vals[nm]) for nm in names for _ in range(3)]
    rows += [row(COUNT, 524288, nm, vals[nm] / 50) for nm in names]          # the small side launch
    rows += [row(COUNT, 524288, "WRITE_SIZE", 24576.0)] * 3
    raw = {nm: B.pmc_aggregate(rows, nm) for nm in names + ["WRITE_SIZE"]}
    from benchmarks.legs import roofline as R
    ent = R.pmc_entry(raw, B.PMC_LEGS["dna_count"], "largest")
    assert ent["rdreq"] == {"all": 32.2e6, "32B": 0.0, "64B": 1200.0, "128B": 32.2e6 - 1200.0} and ent["write_kb"] == 24576.0
    assert ent["fetch_kb_raw"] == round(32.2e6 * 64 / 1024, 1)
    assert R.pmc_entry(raw, ["no_such_kernel"], "largest") is None
    assert R.fabric_read_bytes(ent) == 64 * 1200.0 + 128 * (32.2e6 - 1200.0)


def test_two_stream_roofline_uses_the_same_bytes_over_the_shorter_time():
    ent = {"rdreq": {"all": 4000000.0, "32B": 0.0, "64B": 0.0, "128B": 4000000.0}, "write_kb": 8192.0, "source": "test"}
    cen = {"requested_lines": 4, "requested_records": 3, "requested_probes": 1, "distinct_lines": 4}
    leg = {"roofline": B.make_roofline("walk", 0.14, 1, 1, 0, cen, ent), "two_streams": {"ms_per_batch": 0.10}}
    B.two_stream_roofline(leg)
    t = leg["two_streams"]["roofline"]
    assert t["traffic"] == leg["roofline"]["traffic"] and t["frac"] > leg["roofline"]["frac"]
    assert abs(t["frac"] / leg["roofline"]["frac"] - 1.4) < 0.01
    B.two_stream_roofline(None)
    B.two_stream_roofline({"roofline": {"traffic": None}, "two_streams": {"ms_per_batch": 0.1}})


def test_host_cpu_reports_what_the_process_may_use():
    h = B.host_cpu()
    assert 1 <= h["effective_cpus"] <= h["affinity"] <= (h["os_cpu_count"] or h["affinity"])
    if h["cgroup_cpu_quota"] is not None:
        assert h["effective_cpus"] <= max(1, int(h["cgroup_cpu_quota"]))
    assert "cpu_model" in h and "physical_cores" in h


def test_pretouch_is_optional_preparation_without_a_gpu():
    """bench.py's child process that writes the device's free memory once before the run (pretouch_device): on a
    machine without a GPU it fails in the child and the run goes on -- nothing is raised, the failure is recorded"""
    B.PRETOUCH.clear()
    B.pretouch_device(0)
    assert B.PRETOUCH.get("gib") == 0.0 and (B.PRETOUCH.get("returncode") or B.PRETOUCH.get("error"))
    B.PRETOUCH.clear()


def test_config_3b_counters_come_from_the_larger_grid_of_the_one_launch_kernel():
    """round 5: config 3 and config 3b run the same kernel (fmx_locate_f3u_kernel picks the walk per ticket); the two
    shapes are told apart by their grids; a measurement-build trace with the round-4 lane kernel still resolves"""
    rows = ([row(WALK, 262144, "FETCH_SIZE", 263400.0 + i) for i in range(3)]
            + [row(WALK, 71392256, "FETCH_SIZE", 9.1e6 + i) for i in range(2)])
    agg = B.pmc_aggregate(rows, "FETCH_SIZE")
    kn, v = B.pmc_per_dispatch(agg, B.PMC_LEGS["dna_locate_3b"], B.PMC_WHICH["dna_locate_3b"])
    assert kn.endswith("@grid 71392256") and abs(v - (9.1e6 + 0.5)) < 1.0
    kn, v = B.pmc_per_dispatch(agg, B.PMC_LEGS["dna_locate"], B.PMC_WHICH["dna_locate"])
    assert kn.endswith("@grid 262144") and abs(v - 263401.0) < 1.0
    lane = "fmx_locate_walk_lane_kernel(HIP_vector_type<unsigned int, 4u> const*, ...)"
    rows = [row(lane, 2097152, "FETCH_SIZE", 9.1e6)]
    agg = B.pmc_aggregate(rows, "FETCH_SIZE")
    kn, v = B.pmc_per_dispatch(agg, B.PMC_LEGS["dna_locate_3b"], B.PMC_WHICH["dna_locate_3b"])
    assert "fmx_locate_walk_lane_kernel" in kn and v == 9.1e6


# ---- the line the driver parses (the 21.9 KB line of round 4 left BENCH_r04.parsed = null) ----
CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}
ROOFLINE_KEYS = {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_kernel_ms"}
CPU_KEYS = {"value", "unit", "cores", "kind", "sample"}


def _last_line_of(detail):
    import subprocess
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--headline-of", detail],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    lines = p.stdout.decode().rstrip("\n").splitlines()
    assert len(lines) == 1                       # nothing else on stdout
    return lines[-1]


def test_the_last_stdout_line_is_compact_and_complete():
    """bench.py's print path on the largest result object on record (round 4's 21.9 KB line): the line must stay below
    4 KB, carry the contract's keys with `roofline` and `cpu_baseline`, and every number must be the detail's"""
    import json
    detail = os.path.join(ROOT, "profiles", "r04", "bench_default.json")
    last = _last_line_of(detail)
    assert len(last) < 4096, len(last)
    line = json.loads(last)
    full = json.loads(open(detail).read().strip().splitlines()[-1])
    assert CONTRACT_KEYS <= set(line) and ROOFLINE_KEYS <= set(line["roofline"]) and CPU_KEYS <= set(line["cpu_baseline"])
    for k in CONTRACT_KEYS - {"config", "roofline", "cpu_baseline"}:
        assert line[k] == full[k], k
    assert line["config"]["workload"] == full["config"]["workload"][:300]
    for k in ("achieved", "frac", "traffic", "avg_kernel_ms", "kernel"):
        assert line["roofline"][k] == full["roofline"][k]
    assert line["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and line["cpu_baseline"]["cores"] == 16
    assert abs(line["locate_hits_per_s"] / full["locate"]["hits_per_s"] - 1) < 1e-3
    assert abs(line["rlfm_value"] / full["rlfm"]["value"] - 1) < 1e-3
    assert line["detail"] == detail
    # no nested object beyond the three the contract names
    assert all(not isinstance(v, (dict, list)) for k, v in line.items() if k not in ("config", "roofline", "cpu_baseline"))


def test_the_line_survives_oversized_and_failed_legs(tmp_path):
    """a leg that failed is named (`legs_failed`), absent legs are simply absent, and a pathologically long workload
    string cannot push the line past the limit"""
    import json
    full = json.loads(open(os.path.join(ROOT, "profiles", "r04", "bench_default.json")).read().strip().splitlines()[-1])
    full["wide"] = {"error": "RuntimeError('x')"}
    del full["rlfm"]
    full["per_rank"] = [{"rank": r, "kernel_ms": 0.6 + r, "gather_ms": 0.1, "wall_ms_per_step": 1.0} for r in range(8)]
    full["config"]["workload"] = "w" * 3000
    p = tmp_path / "d.json"
    p.write_text(json.dumps(full))
    last = _last_line_of(str(p))
    assert len(last) < 4096
    line = json.loads(last)
    assert line["legs_failed"] == ["wide"] and "rlfm_value" not in line and line["kernel_ms_max"] == 7.6
    assert CONTRACT_KEYS <= set(line)
