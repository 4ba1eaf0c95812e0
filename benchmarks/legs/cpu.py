"""CPU baseline of bench.py: the oracle (a port of the reference's algorithm; test infrastructure) timed on the
host cores this process may really use."""
import os
import time

from .common import wl_oracle

# --------------------------------------------------------------------------------------------
# CPU baseline: the oracle (port of the reference algorithm) on this box's cores
# --------------------------------------------------------------------------------------------
def host_cpu():
    """what the CPU column really ran on: the cores this process may use (affinity mask, capped by the
    cgroup CPU quota -- os.cpu_count() sees neither), the CPU model, sockets and threads per core"""
    info = {"os_cpu_count": os.cpu_count()}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        info["affinity"] = os.cpu_count() or 1
    quota = None
    try:                                    # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                                # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    info["cgroup_cpu_quota"] = quota
    cores = info["affinity"]
    if quota is not None:
        cores = max(1, min(cores, int(quota)))
    info["effective_cpus"] = cores
    model, phys, siblings, cpu_cores = None, set(), None, None
    try:
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and model is None:
                model = v
            elif k == "physical id":
                phys.add(v)
            elif k == "siblings" and siblings is None:
                siblings = int(v)
            elif k == "cpu cores" and cpu_cores is None:
                cpu_cores = int(v)
    except (OSError, ValueError):
        pass
    info["cpu_model"] = model
    info["sockets"] = len(phys) or None
    info["threads_per_core"] = (siblings // cpu_cores) if siblings and cpu_cores else None
    info["physical_cores"] = (len(phys) * cpu_cores) if phys and cpu_cores else None
    return info


def cpu_baseline(wl, args, kind):
    """the oracle (CPU port of the reference algorithm) on this box's cores: the headline number on the CPUs this
    process may really use (affinity mask capped by the cgroup CPU quota), a thread sweep through and beyond that
    number with the parallel efficiency, and the single-thread rate (the reference itself is single-threaded).
    The threads of a batch pin themselves one per CPU, spread evenly over the allowed CPUs
    (oracle/fm_oracle.c: orc_set_thread_spread); the oracle's bit planes are first touched by the static thread
    decomposition that fills them, so their pages are spread over the sockets like the threads that probe them
    at random."""
    import numpy as np
    oi, t_ob = wl_oracle(wl, kind)
    host = host_cpu()
    cores = host["effective_cpus"]
    oi.set_thread_spread(True)
    m = wl.m
    pat_h = wl.pat.cpu().numpy()
    s_h = wl.d_s.cpu().numpy().view(np.uint64)
    e_h = wl.d_e.cpu().numpy().view(np.uint64)

    def cpu_run(k, threads):
        offk = np.arange(k + 1, dtype=np.uint64) * np.uint64(m)
        t = time.perf_counter()
        so, eo = oi.count_batch(pat_h[:k * m], offk, nthreads=threads)
        return time.perf_counter() - t, so, eo
    budget = args.cpu_seconds
    k0 = min(1 << 14, wl.npat)
    t_probe, so, eo = cpu_run(k0, cores)
    k = int(min(wl.npat, max(k0, k0 * budget / 5 / max(t_probe, 1e-6))))
    t_all, so, eo = cpu_run(k, cores)
    assert (so == s_h[:k]).all() and (eo == e_h[:k]).all(), "GPU != oracle on the CPU sample"
    times = [t_all]
    while sum(times) < budget * 0.5 and len(times) < 25:
        times.append(cpu_run(k, cores)[0])
    t_all = sorted(times)[len(times) // 2]
    value = k * m / t_all
    # one thread, then the sweep: each point ~ budget / 12 seconds of work at the rate of the point before
    k1 = max(1024, int(k0 * (budget / 12) / max(t_probe * cores, 1e-6)))
    k1 = min(k1, wl.npat)
    t_one, _, _ = cpu_run(k1, 1)
    one = k1 * m / t_one
    sweep = [{"threads": 1, "value": one, "scaling_vs_1t": 1.0, "parallel_efficiency": 1.0}]
    rate = one
    # through the effective CPU count and beyond it, up to every CPU the affinity mask shows: where the curve
    # flattens is what this box gives this process, whatever os.cpu_count() says
    phys = host.get("physical_cores") or cores
    limit = host["affinity"]
    eff_cores = min(cores, phys)
    for th in sorted({t for t in (2, 4, 8, 16, 32, 64, 128, phys, cores, limit) if 1 < t <= limit}):
        kk = int(min(wl.npat, max(2048, rate * min(th / sweep[-1]["threads"], 2.0) * (budget / 16) / m)))
        dt, _, _ = cpu_run(kk, th)
        rate = kk * m / dt
        sweep.append({"threads": th, "value": rate, "scaling_vs_1t": round(rate / one, 2),
                      "parallel_efficiency": round(rate / one / min(th, eff_cores), 3)})
    best = max(sweep, key=lambda p: p["value"])
    team = oi.team_size(cores)
    oi.set_thread_spread(False)
    return {"value": max(value, best["value"]), "unit": "pattern-chars/s", "cores": cores, "threads_used": team or cores,
            "kind": "port", "cpu_model": host["cpu_model"], "sockets": host["sockets"],
            "physical_cores": host["physical_cores"], "threads_per_core": host["threads_per_core"],
            "host": {k_: host[k_] for k_ in ("os_cpu_count", "affinity", "cgroup_cpu_quota", "effective_cpus")},
            "placement": "one thread per CPU, spread evenly over the allowed CPUs (sched_setaffinity per batch)",
            "sample": "first %d of the %d patterns (same text, same %s built from the index's exported "
                      "BWT), median of %d runs of %.2f s on %d threads; GPU (s,e) bit-identical on the sample"
                      % (k, wl.npat, "RLFM structure" if kind == "rlfm" else "wavelet matrix", len(times), t_all, cores),
            "all_threads_value": value, "best_threads": best["threads"],
            "single_thread_value": one, "scaling_vs_1t": round(max(value, best["value"]) / one, 2),
            "parallel_efficiency": round(max(value, best["value"]) / one / eff_cores, 3),
            "parallel_efficiency_note": "best rate / (single-thread rate x effective CPUs = min(affinity, cgroup quota, "
                                        "physical cores))",
            "thread_sweep": sweep, "oracle_build_s": round(t_ob, 1)}

