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


WALK = "void fmx_locate_f3p_kernel<4, true>(HIP_vector_type<unsigned int, 4u> const*, ...)"
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
    r = B.make_roofline("k", 1.0, 10, 384, 1000, None, {"bytes": 4_000_000_000, "fetch_kb_raw": 1953125.0,
                                                        "source": "test"})
    assert r["achieved"] == 4000.0 and r["frac"] == 0.5 and r["traffic"] == 4_000_000_000
    r = B.make_roofline("k", 1.0, 10, 384, 1000, {"requested_lines": 10**9}, None)
    assert r["frac"] is None and r["achieved"] is None          # requested lines alone are never a roofline fraction
