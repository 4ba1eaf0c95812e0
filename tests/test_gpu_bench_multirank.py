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
          "--no-pmc", "--no-3b", "--no-d2h", "--no-early-exit", "--pattern-seed", "7"]


def _run(extra, tmp_path, tag):
    dump = str(tmp_path / (tag + ".npy"))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + COMMON + extra + ["--dump-counts", dump],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    line = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1]
    return json.loads(line), np.load(dump)


def test_two_ranks_equal_one_process(tmp_path):
    two, c2 = _run(["--gpus", "2", "--dist-backend", "gloo", "--npat", "4096"], tmp_path, "two")
    one, c1 = _run(["--gpus", "1", "--npat", "8192"], tmp_path, "one")
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and one["n_gpus"] == 1
    assert two["gather"]["counts_wire_dtype"] == "int32"
    assert c2.shape == c1.shape == (8192,)
    assert (c2 == c1).all()
    # aggregate value = all ranks' characters over the max-over-ranks time
    assert two["config"]["patterns_per_gpu"] == 4096 and two["value"] > 0
    # the locate leg gathered every rank's positions
    assert two["locate"]["hits"] == one["locate"]["hits"]
    assert two["locate"]["hits_per_gpu"] <= two["locate"]["hits"]
