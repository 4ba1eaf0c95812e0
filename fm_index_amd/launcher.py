"""One process per GPU: starts N ranks of a script on one node.

`python bench.py --gpus N` calls spawn_ranks() BEFORE anything touches the GPU (no torch import, no
HIP call in the parent): N fresh child processes are started with the torch.distributed
environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT) and the parent
only waits for them.  Nothing is exec-replaced; a failing rank takes the others down and its exit
code becomes the parent's.  This module imports neither torch nor the HIP library.
"""
import os
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, base=None):
    """environment of one rank.  HSA_ENABLE_IPC_MODE_LEGACY=0 is kept when the caller has it and set when
    it is missing: on this ROCm 7 image the host driver only supports dmabuf IPC, and without the variable
    RCCL's intra-node transport (and any device-buffer sharing between the rank processes) fails with
    `hipIpcGetMemHandle: invalid argument`.  The image exports it already; a rank started from a scrubbed
    environment (a test, a service manager) must still get it."""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                "LOCAL_WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                # streams that share one of HIP's hardware queues (4 by default) run in submission order: the
                # count gather would sit between two searches instead of under one
                "GPU_MAX_HW_QUEUES": env.get("GPU_MAX_HW_QUEUES", "8")})
    return env


def spawn_ranks(world, argv, timeout=None, env=None, python=None, poll_s=0.05):
    """Run `python argv...` as `world` ranks; returns the first non-zero exit code (0 if all passed).

    Children inherit stdout / stderr (rank 0 prints the result line).  When one rank fails or the
    timeout expires, the remaining children -- exactly the PIDs started here -- are terminated.
    """
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()
    cmd = [python or sys.executable] + list(argv)
    procs = [subprocess.Popen(cmd, env=rank_env(r, world, port, env)) for r in range(world)]
    deadline = None if timeout is None else time.time() + timeout
    rc = 0
    try:
        live = set(range(world))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 128 - code
            if rc != 0 or (deadline is not None and time.time() > deadline):
                if rc == 0:
                    rc = 124
                break
            if live:
                time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc
