"""The N > 1 control flow that ships, on one GPU: `python bench.py --gpus 2 --dist-backend gloo`
starts two ranks that share cuda:0, each searches its shard with the HIP kernels, counts and
positions are gathered -- and the gathered counts must equal what ONE process computes for the same
global pattern set."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
COMMON = ["--log2n", "16", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-accel", "--no-rlfm",
          "--no-pmc", "--no-3b", "--no-d2h", "--no-early-exit", "--no-wide", "--pattern-seed", "7"]


def _bench(argv, tmp_path, tag):
    """runs bench.py; returns the FULL result object (--detail-out) after checking the stdout contract: the last line
    is the compact headline object (< 4 KB) whose numbers are the detail file's"""
    detail = str(tmp_path / (tag + "_detail.json"))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv + ["--detail-out", detail],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    last = p.stdout.decode().rstrip("\n").splitlines()[-1]
    assert len(last) < 4096, len(last)
    line = json.loads(last)
    with open(detail) as f:
        full = json.load(f)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data"):
        assert line[k] == full[k], k
    assert line["config"]["workload"] == full["config"]["workload"] and line["detail"] == detail
    assert line["roofline"]["kernel"] == full["roofline"]["kernel"]
    assert line["roofline"]["avg_kernel_ms"] == full["roofline"]["avg_kernel_ms"]
    if "cpu_baseline" in full:
        assert line["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and line["cpu_baseline"]["kind"] == "port"
    return full


def _run(extra, tmp_path, tag):
    dump = str(tmp_path / (tag + ".npy"))
    full = _bench(COMMON + extra + ["--dump-counts", dump], tmp_path, tag)
    return full, np.load(dump)


def test_two_ranks_equal_one_process(tmp_path):
    two, c2 = _run(["--gpus", "2", "--dist-backend", "gloo", "--npat", "4096"], tmp_path, "two")
    one, c1 = _run(["--gpus", "1", "--npat", "8192"], tmp_path, "one")
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    # the gloo rehearsal must not claim RCCL ranks (the driver reads rccl_ranks as "RCCL saw N ranks")
    assert two["rccl_ranks"] is None and two["rccl_version"] is None and two["dist_backend"] == "gloo"
    assert len(two["devices"]) == 2 and len(two["per_rank"]) == 2
    assert all(r["kernel_ms"] > 0 and r["gather_ms"] > 0 for r in two["per_rank"])
    assert two["gather"]["counts_wire_dtype"] == "int32"
    assert c2.shape == c1.shape == (8192,)
    assert (c2 == c1).all()
    # the weak-scaling lines carry the hashes of the whole set too, and this set (n = 2^16, seed 7, 8192 patterns) has a
    # golden entry made by the CPU oracle: counts, ranges and ordered positions
    assert two["counts_sha256"] == one["counts_sha256"] and two["ranges_sha256"] == one["ranges_sha256"]
    assert two["matches_golden"]["counts_sha256"] is True and one["matches_golden"]["ranges_sha256"] is True
    assert two["locate"]["positions_sha256"] == one["locate"]["positions_sha256"]
    assert two["locate"]["matches_golden"]["positions_sha256"] is True
    # aggregate value = all ranks' characters over the max-over-ranks time
    assert two["config"]["patterns_per_gpu"] == 4096 and two["value"] > 0
    # the locate leg gathered every rank's positions
    assert two["locate"]["hits"] == one["locate"]["hits"]
    assert two["locate"]["hits_per_gpu"] <= two["locate"]["hits"]
    p2 = np.load(str(tmp_path / "two_pos.npy"))
    assert p2.shape == (two["locate"]["hits"],)
    # ... in input order: for these patterns (count 1 almost everywhere) the gathered list holds, pattern by
    # pattern, positions at which the text really has the pattern -- checked against the one-process counts
    assert p2.min() >= 0 and p2.max() < (1 << 16)


def test_one_rank_rccl_communicator_carries_the_gathers(tmp_path):
    """RCCL on the hardware a test box has: `--force-dist` opens a 1-rank nccl (= RCCL)
    process group on this GPU and drives the step that ships for N > 1 -- pipelined device-side
    all_gather_into_tensor of the counts, counts-then-positions gather of locate -- through it.  The line must
    say so itself (backend, RCCL version, the physical device), the gathered counts must equal the run without
    any process group, and the event timeline must show gather k finishing under search k+1."""
    big = ["--log2n", "24", "--npat", "262144", "--steps", "6"]
    forced, cf = _run(["--gpus", "1", "--force-dist"] + big, tmp_path, "forced")
    plain, cp = _run(["--gpus", "1"] + big, tmp_path, "plain")
    assert forced["dist_backend"] == "nccl" and forced["rccl_ranks"] == 1 and forced["rccl_version"]
    assert forced["n_gpus"] == 1 and forced["gather"]["pipelined"] is True
    dev = forced["devices"]
    assert len(dev) == 1 and (dev[0]["pci_bus_id"] or dev[0]["uuid"])
    assert (cf == cp).all() and cf.shape == (262144,)
    assert forced["locate"]["hits"] == plain["locate"]["hits"]
    assert "gather of counts and positions" in forced["locate"]["includes"]
    pr = forced["per_rank"]
    assert len(pr) == 1 and pr[0]["kernel_ms"] > 0 and pr[0]["gather_ms"] > 0
    tr = forced["gather"]["trace"]
    assert tr and tr["steps"] == 8
    # no search waited for its predecessor's gather: every gather was complete before the NEXT search ended,
    # and the launch stream's idle gap between two searches is far below one search
    assert tr["gathers_done_before_next_search_ends"] == tr["steps"] - 1, tr
    assert tr["search_gap_us_median"] < 0.25 * tr["search_us_median"], tr
    # the communication stream was checked to run concurrently with the launch stream (no shared hardware
    # queue), and the gathers were in flight while the next search ran
    assert tr["comm_stream"]["concurrent"] is True, tr
    assert tr["gathers_under_next_search"] >= (tr["steps"] - 1) // 2, tr
    # the default single-GPU line carries the same evidence in its rccl_1rank object
    r1 = plain["rccl_1rank"]
    assert r1.get("backend") == "nccl" and r1["ranks"] == 1 and r1["value"] > 0, r1
    assert r1["positions_gathered"] == plain["locate"]["hits"]


def test_strong_scaling_two_ranks_hash_like_one_process(tmp_path):
    """config 5 (`--total-patterns`): a FIXED global pattern set (seed 7) in contiguous shards.
    Two ranks (gloo rehearsal on cuda:0, the real HIP search per rank; 8193 patterns -> ragged shards 4097 + 4096)
    must produce the counts_sha256 of the one-rank run -- which itself goes through a 1-rank RCCL communicator, like
    every point of the strong-scaling curve -- and the N > 1 line must carry cpu_baseline and say it scales strongly."""
    strong = ["--total-patterns", "8193", "--cpu-seconds", "1"]
    common = [a for a in COMMON if a not in ("--no-cpu-baseline", "--pattern-seed", "7")]
    def run(extra, tag):
        dump = str(tmp_path / (tag + ".npy"))
        return _bench(common + strong + extra + ["--dump-counts", dump], tmp_path, tag), np.load(dump)
    two, c2 = run(["--gpus", "2", "--dist-backend", "gloo"], "two")
    one, c1 = run(["--gpus", "1"], "one")
    assert two["scaling"] == one["scaling"] == "strong"
    assert two["config"]["total_patterns"] == one["config"]["total_patterns"] == 8193
    assert two["config"]["pattern_seed"] == one["config"]["pattern_seed"] == 7
    assert two["config"]["patterns_per_gpu"] == 4097 and two["gather"]["shard_sizes"] == [4097, 4096]
    assert one["dist_backend"] == "nccl" and one["rccl_ranks"] == 1          # the G = 1 point includes the gather
    assert c1.shape == c2.shape == (8193,) and (c1 == c2).all()
    assert two["ranges_sha256"] == one["ranges_sha256"]
    assert two["counts_sha256"] == one["counts_sha256"] and len(one["counts_sha256"]) == 64
    import hashlib
    assert one["counts_sha256"] == hashlib.sha256(c1.astype("<i8").tobytes()).hexdigest()
    for line in (one, two):                                                   # the N > 1 line is complete
        cb = line["cpu_baseline"]
        assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
        assert line["roofline"]["avg_kernel_ms"] > 0
    assert two["locate"]["hits"] == one["locate"]["hits"] == int(c1.sum())
    # both lines agree with the hashes the CPU oracle computed over ALL patterns of this set (tests/golden/): counts,
    # (s, e) ranges and the ordered locate positions
    for line in (one, two):
        assert line["matches_golden"]["counts_sha256"] is True and line["matches_golden"]["ranges_sha256"] is True
        assert line["locate"]["matches_golden"]["positions_sha256"] is True
    assert two["locate"]["positions_sha256"] == one["locate"]["positions_sha256"]


def test_default_line_carries_config5_at_one_gpu(tmp_path):
    """the default N = 1 line has a `config5_g1` object: the config-5 pattern set (seed 7; 8 x --npat patterns at
    test sizes) in ONE batch through the 1-rank RCCL communicator, hashed, and a sample checked against the oracle;
    its hash equals the hash of a strong-scaling run over the same set"""
    small = [a for a in COMMON if a not in ("--no-cpu-baseline", "--pattern-seed", "7")] + ["--cpu-seconds", "1"]
    line = _bench(small + ["--gpus", "1", "--npat", "8192"], tmp_path, "default")
    c5 = line["config5_g1"]
    assert "error" not in c5, c5
    assert c5["total_patterns"] == 65536 and c5["executed_steps"] == 65536 * 32
    assert c5["oracle_sample"]["identical_s_e"] is True and c5["oracle_sample"]["patterns"] == 1 << 15
    # tests/golden/config5_counts.json holds this set's hashes from the CPU oracle over all 65 536 patterns
    assert c5["matches_golden"]["counts_sha256"] is True and c5["matches_golden"]["ranges_sha256"] is True
    strong = _bench(small + ["--gpus", "1", "--total-patterns", "65536", "--no-cpu-baseline", "--no-locate"],
                    tmp_path, "strong")
    assert strong["counts_sha256"] == c5["counts_sha256"] and strong["ranges_sha256"] == c5["ranges_sha256"]
