#!/usr/bin/env python3
"""What bounds each query kernel: hardware counters per fmx_* kernel of `bench.py --pmc-child [--workload W]`.

    python3 benchmarks/gpu/kernel_pmc.py --tag dna [--workload dna|rep-rlfm|bytes-rlfm] [--out gpurun_out/r05]

One rocprofv3 pass per counter set (counters never combined with trace domains other than --kernel-trace; sets sized to
the per-block slots of MI355X_MICROARCH.md "rocprofv3 PMC slots": SQ 8, TCC 4, GRBM 2).  Per kernel (heavy launches
only, averaged) the summary derives:
  fabric_bytes   = 32 B x RDREQ_32B + 64 B x RDREQ_64B + 128 B x RDREQ_128B   (counted widths: no census, no x2 rule)
  l2_hit         = TCC_HIT / (TCC_HIT + TCC_MISS)
  valu_busy      = SQ_ACTIVE_INST_VALU x 4 / SIMDs / SQ_BUSY_CYCLES-ish (per-SIMD issue share; see `derive`)
  wave states    = WAIT_ANY (parked on s_waitcnt) / WAIT_INST_ANY (issue stall) / ACTIVE_INST_ANY shares of WAVE_CYCLES
  ta_busy        = TA_TA_BUSY_sum / (TAs x kernel cycles): the address unit of the vector memory path
  l1: TCP_TOTAL_CACHE_ACCESSES, TCP_TCC_READ_REQ (L1 misses sent to L2), the TCP stall cycles
The process under the profiler is python3 itself (never a shell or env hop)."""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SETS = {
    "ea_widths": ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"],
    "l2": ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_EA0_RDREQ_DRAM_sum"],
    "ea_wr": ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_TAG_STALL_sum", "TCC_BUBBLE_sum"],
    "sq_insts": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS",
                 "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"],
    "sq_states": ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM",
                  "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"],
    "sq_smem": ["SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INST_CYCLES_SALU", "SQ_WAVE_CYCLES"],
    "ta": ["TA_TA_BUSY_sum", "TA_FLAT_READ_WAVEFRONTS_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum",
           "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TA_BUSY_max", "GRBM_GUI_ACTIVE"],
    "tcp": ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum",
            "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "TCP_TCR_TCP_STALL_CYCLES_sum"],
    "utcl1": ["TCP_UTCL1_REQUEST_sum", "TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum",
              "TCP_TCC_READ_REQ_LATENCY_sum"],
}
N_CU, N_SIMD, N_XCD = 256, 1024, 8


def kernel_key(kn, grid):
    """kernels by name; the one-launch DNA walk kernel also by grid (config 3 and config 3b run under one name)"""
    key = kn.split("(")[0].replace("void ", "")[:100]
    if "fmx_locate_f3u_kernel" in key and grid:
        key += " @grid %s" % grid
    return key


PASS_TIMEOUT = [600]


def run_pass(name, counters, child, work):
    d = os.path.join(work, name)
    cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "-d", d, "--output-format", "csv", "--"] + child
    p = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, start_new_session=True)
    try:
        _, err = p.communicate(timeout=PASS_TIMEOUT[0])
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, 9)
        p.communicate()
        return None, "timeout"
    if p.returncode != 0:
        return None, err.decode(errors="replace")[-400:]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))      # kernel -> counter -> [per launch]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection*.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                kn = r.get("Kernel_Name", "")
                if "fmx_" not in kn:
                    continue
                key = kernel_key(kn, r.get("Grid_Size"))
                vals[key][r["Counter_Name"]].append(float(r.get("Counter_Value", 0) or 0))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace*.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                kn = r.get("Kernel_Name", "")
                if "fmx_" in kn:
                    dur[kernel_key(kn, r.get("Grid_Size"))].append(
                        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    shutil.rmtree(d, ignore_errors=True)
    return (vals, dur), None


def heavy_mean(v):
    """mean over a kernel's heavy launches (a kernel may also run once on a small side batch)"""
    if not v:
        return None
    mx = max(v)
    h = [x for x in v if x >= 0.5 * mx] if mx > 0 else v
    return sum(h) / len(h)


def derive(c, ms):
    o = {}
    g = c.get

    def ratio(a, b):
        return round(a / b, 4) if a is not None and b else None
    if g("TCC_EA0_RDREQ_sum") is not None:
        n32, n64, n128 = g("TCC_EA0_RDREQ_32B_sum") or 0, g("TCC_EA0_RDREQ_64B_sum") or 0, g("TCC_EA0_RDREQ_128B_sum") or 0
        o["fabric_read_requests"] = g("TCC_EA0_RDREQ_sum")
        o["fabric_read_bytes"] = 32 * n32 + 64 * n64 + 128 * n128
        o["read_request_widths"] = {"32B": n32, "64B": n64, "128B": n128,
                                    "other": g("TCC_EA0_RDREQ_sum") - n32 - n64 - n128}
        if ms:
            o["fabric_read_GBps"] = round(o["fabric_read_bytes"] / ms / 1e6, 1)
            o["fabric_read_requests_per_s"] = g("TCC_EA0_RDREQ_sum") / (ms / 1e3)
        o["dram_share_of_read_requests"] = ratio(g("TCC_EA0_RDREQ_DRAM_sum"), g("TCC_EA0_RDREQ_sum"))
    if g("TCC_HIT_sum") is not None:
        o["l2_hit"] = ratio(g("TCC_HIT_sum"), (g("TCC_HIT_sum") or 0) + (g("TCC_MISS_sum") or 0))
        o["l2_requests"] = g("TCC_REQ_sum")
    if g("SQ_WAVE_CYCLES"):
        wc = g("SQ_WAVE_CYCLES")
        o["wave_cycle_shares"] = {"parked_on_waitcnt": ratio(g("SQ_WAIT_ANY"), wc), "issue_stall": ratio(g("SQ_WAIT_INST_ANY"), wc),
                                  "issuing": ratio(g("SQ_ACTIVE_INST_ANY"), wc)}
    if g("GRBM_GUI_ACTIVE"):
        cyc = g("GRBM_GUI_ACTIVE") / N_XCD           # the counter is summed over the 8 XCDs
        o["gpu_cycles"] = cyc
        # SQ_ACTIVE_INST_* count per-wave cycles in units of 4 (one quad-cycle): x 4 / SIMDs / kernel cycles = issue share
        for k, n in (("SQ_ACTIVE_INST_VALU", "valu_busy"), ("SQ_ACTIVE_INST_VMEM", "vmem_issue_busy"),
                     ("SQ_ACTIVE_INST_SCA", "scalar_busy"), ("SQ_ACTIVE_INST_LDS", "lds_issue_busy")):
            if g(k) is not None:
                o[n] = round(g(k) * 4 / N_SIMD / cyc, 4)
        if g("SQ_INSTS_VALU") is not None:
            o["valu_insts_x4_per_simd_cycle"] = round(g("SQ_INSTS_VALU") * 4 / N_SIMD / cyc, 4)
        if g("TA_TA_BUSY_sum") is not None:
            o["ta_busy"] = round(g("TA_TA_BUSY_sum") / N_CU / cyc, 4)
            o["ta_addr_stalled_by_tc"] = round((g("TA_ADDR_STALLED_BY_TC_CYCLES_sum") or 0) / N_CU / cyc, 4)
            o["ta_data_stalled_by_tc"] = round((g("TA_DATA_STALLED_BY_TC_CYCLES_sum") or 0) / N_CU / cyc, 4)
        if g("SQ_WAVE_CYCLES") and g("SQ_BUSY_CYCLES"):
            o["mean_waves_per_simd"] = round(g("SQ_WAVE_CYCLES") * 4 / N_SIMD / cyc, 2) if cyc else None   # (units of 4 cycles)
    if g("SQ_INSTS_VMEM_RD"):
        o["valu_per_vmem_rd"] = ratio(g("SQ_INSTS_VALU"), g("SQ_INSTS_VMEM_RD"))
        if g("TCP_TOTAL_CACHE_ACCESSES_sum") is not None:
            o["l1_accesses_per_vmem_rd_inst"] = ratio(g("TCP_TOTAL_CACHE_ACCESSES_sum"), g("SQ_INSTS_VMEM_RD"))
    if g("TCP_TOTAL_CACHE_ACCESSES_sum"):
        o["l1_miss_share"] = ratio(g("TCP_TCC_READ_REQ_sum"), g("TCP_TOTAL_CACHE_ACCESSES_sum"))
    if g("TCP_UTCL1_REQUEST_sum"):
        o["utcl1_miss_share"] = ratio(g("TCP_UTCL1_TRANSLATION_MISS_sum"), g("TCP_UTCL1_REQUEST_sum"))
    if g("TCP_TCC_READ_REQ_LATENCY_sum") and g("TCP_TCC_READ_REQ_sum"):
        o["l1_to_l2_read_latency_cycles"] = round(g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum"), 1)
    return o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="dna")
    ap.add_argument("--workload", default="dna")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06"))
    # (the TA / TCP sets take > 10 minutes per pass on this profiler: ask for them by name)
    ap.add_argument("--sets", default="ea_widths,l2,ea_wr,sq_insts,sq_states,utcl1")
    ap.add_argument("--child-args", default="", help="extra flags for bench.py --pmc-child, comma-separated: no-rlfm,no-accel,log2n=27")
    ap.add_argument("--pass-timeout", type=int, default=600, help="seconds per counter pass (the TA / TCP sets are slow)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    PASS_TIMEOUT[0] = a.pass_timeout
    work = os.path.join("/tmp", "fmx_kernel_pmc_%d" % os.getpid())
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--workload", a.workload]
    for x in a.child_args.split(","):              # "no-rlfm" -> --no-rlfm; "log2n=27" -> --log2n 27
        if x:
            child += ["--" + x.split("=", 1)[0]] + x.split("=", 1)[1:]
    counters = collections.defaultdict(dict)
    durs = collections.defaultdict(list)
    notes = {}
    for name in a.sets.split(","):
        res, err = run_pass(name, SETS[name], child, work)
        if res is None:
            notes[name] = err
            continue
        vals, dur = res
        for kn, cs in vals.items():
            for cn, v in cs.items():
                counters[kn][cn] = heavy_mean(v)
        for kn, v in dur.items():
            durs[kn].append(heavy_mean(v))
    shutil.rmtree(work, ignore_errors=True)
    out = {"workload": a.workload, "child": " ".join(child[1:]), "failed_passes": notes, "kernels": {}}
    for kn in sorted(counters):
        ms = sum(durs[kn]) / len(durs[kn]) if durs.get(kn) else None
        out["kernels"][kn] = {"ms_under_pmc": round(ms, 4) if ms else None, "derived": derive(counters[kn], ms),
                              "counters": counters[kn]}
    path = os.path.join(a.out, "kernel_pmc_%s.json" % a.tag)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    for kn, o in out["kernels"].items():
        if o["ms_under_pmc"] and o["ms_under_pmc"] > 0.02:
            print(kn, o["ms_under_pmc"], json.dumps(o["derived"]))
    print("failed:", notes, "->", path)


if __name__ == "__main__":
    main()
