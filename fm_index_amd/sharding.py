"""Multi-GPU layer: patterns are independent (wrapper.rs:103-124 reads
only immutable index state), so a batch shards contiguously across ranks with the index
REPLICATED on every GPU, and the only exchange is the result gather.  One process per GPU;
torch.distributed is the transport ("nccl" = RCCL over xGMI on the GPU box, "gloo" in the
CPU tests).  No data-path collective exists besides these gathers.
"""
import torch
import torch.distributed as dist


def shard_range(nitems, rank, world):
    """Contiguous shard [lo, hi) of rank: item k belongs to rank floor(k * world / nitems)."""
    lo = (nitems * rank + world - 1) // world
    hi = (nitems * (rank + 1) + world - 1) // world
    return lo, hi


def gather_counts(local, nitems, group=None):
    """All-gather per-pattern values (counts, s or e) back into input order.

    `local` is this rank's 1-D int64 tensor for its shard_range; returns the full tensor of
    `nitems` entries on every rank.  Equal shards use one all_gather_into_tensor; ragged
    shards are padded to the largest shard.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(nitems, r, world)[1] - shard_range(nitems, r, world)[0] for r in range(world)]
    assert local.numel() == sizes[rank]
    if len(set(sizes)) == 1:
        out = torch.empty(nitems, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    mx = max(sizes)
    pad = torch.zeros(mx, dtype=local.dtype, device=local.device)
    pad[:local.numel()] = local
    buf = torch.empty(mx * world, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * mx:r * mx + sizes[r]] for r in range(world)])


class PositionGatherPlan:
    """Variable-length gather of locate output, planned once and repeated without host synchronisation.

    The counts of a batch decide how many positions every rank contributes; gathering them, turning them into
    offsets and sizing the padded buffer needs ONE device-to-host synchronisation.  A caller that locates the
    same intervals again (bench.py's timed steps), or that knows an upper bound, pays it once: gather() is then
    a copy into the padded slot and one all_gather_into_tensor, with nothing for the host to wait for.
    Result layout: rank r's positions at buf[r * mx : r * mx + totals[r]], input order within a rank --
    compact() concatenates them into exactly what a single GPU produces.
    """

    def __init__(self, local_counts, nitems, group=None, pos_dtype=torch.int64):
        self.group, self.nitems = group, nitems
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        counts = gather_counts(local_counts, nitems, group)
        self.off = torch.zeros(nitems + 1, dtype=torch.int64, device=counts.device)
        self.off[1:] = torch.cumsum(counts, 0)
        # hits of every rank's shard with ONE device-to-host synchronisation
        cuts = torch.tensor([shard_range(nitems, r, self.world)[0] for r in range(self.world)] + [nitems],
                            dtype=torch.int64, device=counts.device)
        self.totals = (self.off[cuts[1:]] - self.off[cuts[:-1]]).tolist()
        self.mx = max(max(self.totals), 1)
        self.buf = torch.empty(self.mx * self.world, dtype=pos_dtype, device=counts.device)
        self.pad = torch.zeros(self.mx, dtype=pos_dtype, device=counts.device)

    def gather(self, local_pos):
        """all ranks' positions, padded layout; no host synchronisation"""
        n = self.totals[self.rank]
        src = local_pos[:n]
        if n == self.mx and src.is_contiguous():
            send = src
        else:
            self.pad[:n].copy_(src)
            send = self.pad
        dist.all_gather_into_tensor(self.buf, send, group=self.group)
        return self.buf

    def compact(self):
        if all(t == self.mx for t in self.totals):
            return self.buf
        return torch.cat([self.buf[r * self.mx:r * self.mx + self.totals[r]] for r in range(self.world)])


def compact_padded(buf, sizes, slot):
    """rank r's `sizes[r]` valid entries sit at buf[r * slot : r * slot + sizes[r]] (the layout of an
    all_gather_into_tensor of equal, padded slots); returns them concatenated = input order.  Equal full slots
    need no copy."""
    if all(sz == slot for sz in sizes):
        return buf[:slot * len(sizes)]
    return torch.cat([buf[r * slot:r * slot + sz] for r, sz in enumerate(sizes)])


def gather_positions(local_counts, local_pos, nitems, group=None):
    """Variable-length gather of locate output, one shot.

    Returns (offsets[nitems+1], positions[total]) in input order, identical on every rank and
    identical to what a single GPU produces: counts are gathered first, then the position
    lists padded to the largest shard total (PositionGatherPlan).
    """
    plan = PositionGatherPlan(local_counts, nitems, group, pos_dtype=local_pos.dtype)
    plan.gather(local_pos)
    return plan.off, plan.compact()


def wire_dtype(text_len, force=None):
    """dtype the per-pattern counts travel in: counts are <= len (wrapper.rs:132-134: e - s), so
    int32 halves the gather while len < 2^31; beyond that they stay int64.  Forcing int32 on a
    longer text is refused instead of truncating."""
    if force is not None:
        if force == torch.int32 and text_len >= (1 << 31):
            raise ValueError("counts of a text with len >= 2^31 do not fit the int32 wire format")
        return force
    return torch.int32 if text_len < (1 << 31) else torch.int64


def pick_concurrent_stream(device, other=None, tries=8, spin_cycles=400000):
    """A stream that REALLY runs concurrently with `other` (default: the current stream).

    HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default); two streams that
    land on the same queue execute in submission order, however independent they look
    (profiles/r03/rccl_overlap_probe_hwqueues.jsonl: the count gather of step k then runs BETWEEN
    searches k and k+1 instead of under k+1).  So candidates are tested: one single-thread spin kernel
    on each stream -- together they take one spin when the queues differ, two when they alias.
    Returns (stream, {"tried", "concurrent", "ratio"}); the last candidate is returned even when none
    overlapped (the pipeline stays correct, just serialised)."""
    device = torch.device(device)
    other = other or torch.cuda.current_stream(device)
    if not hasattr(torch.cuda, "_sleep"):       # no spin kernel in this torch build: take a stream untested
        return torch.cuda.Stream(device=device), {"tried": 1, "concurrent": None, "ratio": None}
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda._sleep(1000)
    torch.cuda.synchronize(device)
    # duration of one spin
    e[0].record(other)
    with torch.cuda.stream(other):
        torch.cuda._sleep(spin_cycles)
    e[1].record(other)
    torch.cuda.synchronize(device)
    one = max(e[0].elapsed_time(e[1]), 1e-3)
    info = {"tried": 0, "concurrent": False, "ratio": None}
    cand = None
    keep = []                                   # rejected candidates stay alive so that the pool moves on
    for _ in range(tries):
        cand = torch.cuda.Stream(device=device)
        info["tried"] += 1
        cand.wait_stream(other)
        e[2].record(other)
        with torch.cuda.stream(other):
            torch.cuda._sleep(spin_cycles)
        with torch.cuda.stream(cand):
            torch.cuda._sleep(spin_cycles)
        other.wait_stream(cand)
        e[3].record(other)
        torch.cuda.synchronize(device)
        ratio = e[2].elapsed_time(e[3]) / one
        info["ratio"] = round(ratio, 2)
        if ratio < 1.5:
            info["concurrent"] = True
            break
        keep.append(cand)
    return cand, info


class CountGatherPipeline:
    """The N > 1 step of bench.py (BASELINE config 5): search this rank's shard, all-gather the
    per-pattern counts of every rank.

    Two result buffers alternate.  The launch stream only ever runs the search kernels: the down-cast to
    the wire dtype and the collective of step k go to the pipeline's own communication stream (picked so
    that it does not share a hardware queue with the launch stream, pick_concurrent_stream), which waits
    for search k through an event and runs under search k+1; a buffer is reused only after its gather has
    completed (event wait on the launch stream, two steps later).  `launch(out64)` must enqueue the count
    of this rank's shard on the current stream, writing int64 counts into `out64` (the ABI's u64 counts).
    With backend "gloo" (CPU tests, single-GPU rehearsal) the counts go through host memory and the
    gather is torch.distributed's async Work.
    """

    def __init__(self, npat_local, world, text_len, device, backend="nccl", pipelined=True,
                 group=None, force_wire=None, force_collective=False, trace=False):
        self.npat, self.world, self.group = npat_local, world, group
        self.device = torch.device(device)
        self.host = backend == "gloo"
        self.wire = wire_dtype(text_len, force_wire)
        # force_collective: run the gather even on a 1-rank communicator (`bench.py --force-dist`: the
        # RCCL code path -- communicator, device-side all_gather_into_tensor, ordering against the next
        # search -- on the one GPU a test box has)
        self.collective = world > 1 or force_collective
        self.nbuf = 2 if (pipelined and self.collective) else 1
        # zeros: with ragged shards (npat_local = the largest shard) a smaller rank never writes its slot's tail
        self.local64 = [torch.zeros(npat_local, dtype=torch.int64, device=self.device) for _ in range(self.nbuf)]
        gdev = torch.device("cpu") if self.host else self.device
        self.local_w = [torch.empty(npat_local, dtype=self.wire, device=gdev) for _ in range(self.nbuf)]
        self.gathered = [torch.empty(npat_local * world, dtype=self.wire, device=gdev)
                         for _ in range(self.nbuf)] if self.collective else []
        self.pending = [None] * self.nbuf
        self.k = 0
        # device-side pipeline: own communication stream + events
        self.on_device = self.collective and not self.host and self.device.type == "cuda"
        self.comm, self.comm_info = None, None
        if self.on_device and self.nbuf > 1:
            with torch.cuda.device(self.device):
                self.comm, self.comm_info = pick_concurrent_stream(self.device)
            # one pair of events per buffer, reused (an event per step made the runtime grow its signal pool
            # in the middle of a run: one 22 ms stall of the launch stream in every 50-step measurement)
            self._searched = [torch.cuda.Event() for _ in range(self.nbuf)]
            self._done = [torch.cuda.Event() for _ in range(self.nbuf)]
        # trace: HIP events around every search launch (launch stream) and behind every gather
        # (communication stream): trace_report() turns them into the evidence that gather k ran under
        # search k+1
        self.trace = bool(trace) and self.on_device
        self.events = []

    def _wait(self, b):
        p = self.pending[b]
        if p is None:
            return
        if isinstance(p, torch.cuda.Event):
            torch.cuda.current_stream(self.device).wait_event(p)
        else:
            p.wait()
        self.pending[b] = None

    def step(self, launch):
        """one search + gather; returns the tensor that will hold all ranks' counts in input
        order (complete after drain(), or after the next step() on the same buffer)."""
        b = self.k % self.nbuf
        self.k += 1
        self._wait(b)                            # this buffer's previous gather must be done
        ks = ke = None
        if self.trace:
            ks, ke = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ks.record()
        launch(self.local64[b])
        if self.trace:
            ke.record()
        if not self.collective:
            return self.local64[b]
        if self.comm is not None:
            # searched[b] -> communication stream: down-cast + collective there, launch stream moves on
            cur = torch.cuda.current_stream(self.device)
            searched = ke if ke is not None else self._searched[b]
            if ke is None:
                searched.record(cur)
            done = torch.cuda.Event(enable_timing=True) if self.trace else self._done[b]
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(searched)
                self.local_w[b].copy_(self.local64[b])
                dist.all_gather_into_tensor(self.gathered[b], self.local_w[b], group=self.group)
                done.record(self.comm)
            self.pending[b] = done
            if self.trace:
                self.events.append((ks, ke, done))
            return self.gathered[b]
        self.local_w[b].copy_(self.local64[b])   # down-cast (and D2H for gloo)
        work = dist.all_gather_into_tensor(self.gathered[b], self.local_w[b], group=self.group,
                                           async_op=self.nbuf > 1)
        if self.nbuf > 1:
            self.pending[b] = work
        if self.trace:
            gd = torch.cuda.Event(enable_timing=True)
            gd.record()
            self.events.append((ks, ke, gd))
        return self.gathered[b]

    def drain(self):
        for b in range(self.nbuf):
            self._wait(b)

    def trace_report(self):
        """Timeline of the traced steps (call after drain() + synchronize): per step the search
        kernel's start / end and the moment its gather had completed, in microseconds from the first
        launch.  `gathers_under_next_search` counts the steps whose gather completed AFTER the next
        search had started and BEFORE it ended -- the collective and the kernel were in flight
        together; `search_gap_us` is the idle time of the launch stream between two searches (a
        serialised gather would show up here)."""
        ev = self.events
        if len(ev) < 2:
            return None
        t0 = ev[0][0]
        ks = [t0.elapsed_time(e[0]) * 1e3 for e in ev]
        ke = [t0.elapsed_time(e[1]) * 1e3 for e in ev]
        gd = [t0.elapsed_time(e[2]) * 1e3 for e in ev]
        n = len(ev)
        under = sum(1 for k in range(n - 1) if ks[k + 1] < gd[k] < ke[k + 1])
        before_end = sum(1 for k in range(n - 1) if gd[k] < ke[k + 1])
        gaps = sorted(ks[k + 1] - ke[k] for k in range(n - 1))
        glat = sorted(gd[k] - ke[k] for k in range(n))
        kern = sorted(ke[k] - ks[k] for k in range(n))
        return {"steps": n, "gathers_under_next_search": under, "gathers_done_before_next_search_ends": before_end,
                "search_gap_us_median": round(gaps[len(gaps) // 2], 2), "search_gap_us_max": round(gaps[-1], 2),
                "gather_latency_us_median": round(glat[n // 2], 2), "search_us_median": round(kern[n // 2], 2),
                "comm_stream": self.comm_info,
                "timeline_us": [[round(ks[k], 1), round(ke[k], 1), round(gd[k], 1)] for k in range(min(n, 8))]}


# ---------------------------------------------------------------------------------------------
# which physical GPU does each rank sit on?  (one process per GPU: LOCAL_RANK is only a promise)
# ---------------------------------------------------------------------------------------------
def device_identity(local):
    """{"pci_bus_id", "uuid", "name"} of cuda:`local` -- the PCI bus id straight from the HIP runtime
    this process already uses (hipDeviceGetPCIBusId), the uuid / name from torch's device properties."""
    import ctypes as C
    ident = {"pci_bus_id": None, "uuid": None, "name": None}
    try:
        props = torch.cuda.get_device_properties(local)
        ident["name"] = getattr(props, "name", None)
        u = getattr(props, "uuid", None)
        ident["uuid"] = str(u) if u is not None else None
    except Exception:  # noqa: BLE001
        pass
    try:
        import os
        hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        buf = C.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, C.c_int(local)) == 0:
            ident["pci_bus_id"] = buf.value.decode()
    except Exception:  # noqa: BLE001
        pass
    return ident


def gather_device_identities(local, group=None):
    """every rank's device_identity(), in rank order, on every rank"""
    mine = device_identity(local)
    world = dist.get_world_size(group)
    out = [None] * world
    dist.all_gather_object(out, mine, group=group)
    return out


def assert_distinct_devices(idents):
    """N ranks must sit on N distinct physical GPUs (PCI bus ids; uuids when no bus id is known)"""
    keys = [i.get("pci_bus_id") or i.get("uuid") for i in idents]
    if any(k is None for k in keys):
        raise RuntimeError("cannot identify the GPUs of all ranks: %r" % (idents,))
    if len(set(keys)) != len(keys):
        raise RuntimeError("ranks share a physical GPU: %r" % (keys,))
    return keys
