"""One-process-per-GPU plumbing for the render path (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm).

The path shards by view: every rank renders its own stereo pairs and there is NO data-path collective.
What is shared is only (a) the launch protocol bench.py and the driver use (env rendezvous, barrier, MAX-over-ranks
timing) and (b) for stage-2 training, one bucketed gradient all-reduce of the network parameters per step -- the
reference has no distributed code at all (train_stage2.py:27-55 is single process), so this is new glue around its
unmodified Trainer, not a translation of anything.  Everything here also runs on CPU with the gloo backend
(tests/test_multiproc_gloo.py, world_size 2).
"""
import os
import time

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def forced():
    """GPSGS_DIST_FORCE=1: initialise the process group and issue the exchange-step collectives even at world size 1 -- the way a 1-GPU box
    exercises RCCL itself (communicator set-up + a 20.6 MB all-reduce kernel) instead of skipping it."""
    return os.environ.get("GPSGS_DIST_FORCE") == "1"


def init(backend=None, device=None, timeout_s=None):
    """Initialise the default process group from the torchrun environment.  Returns (rank, local_rank, world).
    timeout_s: the collectives' watchdog timeout, set EXPLICITLY (default: $GPSGS_PG_TIMEOUT_S, else 1800 s) -- rank 0 of the stage-2 launcher
    evaluates a validation set while the other ranks wait for it (tools/launch_stage2.py), which must not look like a hang."""
    import datetime
    rank, local_rank, world = env_rank()
    if (world > 1 or forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        if timeout_s is None:
            timeout_s = float(os.environ.get("GPSGS_PG_TIMEOUT_S", "1800"))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s), **kw)
    return rank, local_rank, world


def _parse_cpulist(txt):
    cpus = set()
    for part in txt.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_cpus(device_index):
    """CPUs of the NUMA node the GPU hangs off (sysfs: /sys/bus/pci/devices/<bdf>/numa_node -> /sys/devices/system/node/node<n>/cpulist), or None
    when the platform does not say (numa_node = -1, no sysfs, no PCI ids from the runtime)."""
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (int(getattr(pr, "pci_domain_id", 0)), int(pr.pci_bus_id), int(pr.pci_device_id))
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip())
        if node < 0:
            return None
        return _parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read()) or None
    except Exception:  # noqa: BLE001
        return None


def set_cpu_affinity(local_rank, local_world, device_index=None, peer_device_indices=None, n_cpus=None):
    """One process per GPU, and every process is host-bound in bursts (six HIP streams of launches per rank in bench.py, the DataLoader workers of
    the trainer): pin each rank to its own slice of the CPUs -- those of its GPU's NUMA node when sysfs names one, else an even split of whatever
    this process may run on -- so that 8 ranks do not migrate across sockets or pile onto the same cores.  GPSGS_AFFINITY=0 switches it off.
    peer_device_indices: the device index of every LOCAL rank (default: rank r drives device r, which is what torchrun + LOCAL_RANK gives); pass it
    when the mapping differs.  local_world must be the number of ranks ON THIS NODE: if the launcher did not say (no LOCAL_WORLD_SIZE) and the
    caller fell back to the world size of a multi-node job, the slices would be 1 / world of the node -- so a local_world larger than the visible
    GPU count is refused (returns None, with a warning) instead of idling most cores.  After pinning, torch's intra-op pool is sized to the slice
    (it was sized for the whole machine at import time; N threads squeezed onto cores / N CPUs thrash).  n_cpus: narrow the rank's slice further, to that
    many CPUs of ONE L3 domain inside it (a rank that only drives its GPU, like bench.py: see pin_near_gpu; a trainer keeps the whole slice for
    its DataLoader workers).  Returns the CPU set chosen (or None: left alone)."""
    if os.environ.get("GPSGS_AFFINITY", "1") == "0" or not hasattr(os, "sched_setaffinity") or local_world <= 1:
        return None
    try:
        n_vis = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_vis and local_world > n_vis and "LOCAL_WORLD_SIZE" not in os.environ:
            print("gps_gaussian_amd.dist: %d ranks but %d visible GPUs and no LOCAL_WORLD_SIZE: not pinning CPUs (the local rank count is unknown)" % (local_world, n_vis))
            return None
        allowed = sorted(os.sched_getaffinity(0))
        node = gpu_numa_cpus(device_index) if device_index is not None else None
        pool = sorted(set(allowed) & node) if node else allowed
        # ranks that share a pool (same NUMA node, or no topology information: everybody) split it evenly, in local-rank order
        sharers, me = local_world, local_rank
        if node:
            peers = list(peer_device_indices) if peer_device_indices is not None else list(range(local_world))
            same = [r for r in range(local_world) if (gpu_numa_cpus(peers[r]) or set()) == node]
            sharers, me = max(1, len(same)), (same.index(local_rank) if local_rank in same else 0)
        per = max(1, len(pool) // sharers)
        mine = pool[me * per:(me + 1) * per] or pool
        if n_cpus and len(mine) > n_cpus:
            mine = _quiet_l3_cpus(mine, n_cpus)
        os.sched_setaffinity(0, mine)
        try:
            torch.set_num_threads(max(1, len(mine)))
        except Exception:  # noqa: BLE001
            pass
        return set(mine)
    except Exception:  # noqa: BLE001
        return None


def _cpu_busy_sample(interval_s=0.03):
    """Per-CPU busy jiffies over a short interval (/proc/stat), {cpu: busy} -- {} where /proc/stat is not readable."""
    def snap():
        out = {}
        for line in open("/proc/stat"):
            if line.startswith("cpu") and line[3].isdigit():
                f = line.split()
                v = [int(x) for x in f[1:9]]
                out[int(f[0][3:])] = sum(v) - v[3] - v[4]  # everything but idle and iowait
        return out
    try:
        a = snap()
        time.sleep(interval_s)
        b = snap()
        return {c: b[c] - a.get(c, 0) for c in b}
    except Exception:  # noqa: BLE001
        return {}


def _quiet_l3_cpus(pool, n_cpus):
    """n_cpus CPUs of ONE L3 domain out of `pool`: the domain that is least busy right now, one hardware thread per core first."""
    # L3 domains of the pool (sysfs); without sysfs: aligned runs of 2 * n_cpus CPU numbers
    groups = {}
    for c in sorted(pool):
        try:
            key = open("/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list" % c).read().strip()
        except Exception:  # noqa: BLE001
            key = "run%d" % (c // (2 * n_cpus))
        groups.setdefault(key, []).append(c)
    busy = _cpu_busy_sample()
    best = min(groups.values(), key=lambda cs: (sum(busy.get(c, 0) for c in cs) / len(cs), cs[0]))
    # one hardware thread per core first (the lowest-numbered sibling), the siblings only if the domain has fewer cores than asked for
    first, rest = [], []
    for c in best:
        try:
            sib = sorted(_parse_cpulist(open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read()))
        except Exception:  # noqa: BLE001
            sib = [c]
        (first if c == sib[0] or sib[0] not in best else rest).append(c)
    first.sort(key=lambda c: (busy.get(c, 0), c))
    return (first + rest)[:n_cpus]


_affinity_before_pin = None
_threads_before_pin = None
_pinned_set = None


def pin_near_gpu(device_index=0, n_cpus=8):
    """ONE process driving one GPU through the autograd API (the reference's train_stage2.py, bench.py's `value`): pin the calling thread -- and every
    thread it starts afterwards: the autograd engine's worker, the loader's -- to n_cpus CPUs of ONE L3 domain on the GPU's NUMA node, the least busy
    domain right now.  A step of the drop-in rasteriser is ~250 us of GPU work fed by ~240 us of Python / PyTorch host work spread over two threads
    (tools/host_time.py; round 5: profiles/r05_host_timeline.md); left to the scheduler on a 256-CPU box that host work migrates between cores and sockets and runs ~40 % slower (measured
    on one box, same run: forward prologue 73 -> 42 us, notification -> backward launched 82 -> 50 us, backward launched -> next forward launched
    153 -> 100 us; the step went from host-bound, 272 us, to GPU-bound, 252 us).  Pinning to the whole NUMA node does NOT do it (257 us): it is the
    shared L3 and the absence of migrations that count.  GPSGS_AFFINITY=0 switches it off.  restore_affinity() undoes it (CPU-heavy legs: an OpenMP
    baseline, DataLoader workers).  -> the CPU set chosen, or None (left alone)."""
    global _affinity_before_pin, _threads_before_pin, _pinned_set
    if os.environ.get("GPSGS_AFFINITY", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        allowed = set(os.sched_getaffinity(0))
        node = gpu_numa_cpus(device_index) if torch.cuda.is_available() else None
        pool = (allowed & node) if node and (allowed & node) else allowed
        if len(pool) <= n_cpus:
            return None
        mine = _quiet_l3_cpus(pool, n_cpus)
        if _affinity_before_pin is None:
            _affinity_before_pin = allowed
            _threads_before_pin = torch.get_num_threads()
        os.sched_setaffinity(0, mine)
        _pinned_set = set(mine)
        try:
            torch.set_num_threads(max(1, len(mine)))
        except Exception:  # noqa: BLE001
            pass
        return set(mine)
    except Exception:  # noqa: BLE001
        return None


def restore_affinity():
    """Undo pin_near_gpu(): the calling thread and every thread of the process that still carries the PINNED CPU set (= was started while pinned,
    or is the caller) get the original set back.  Threads that hold any other mask -- started before the pin, or pinned deliberately by somebody
    else (a DataLoader's pin-memory thread, RCCL's proxy threads) -- are left alone.  pin_near_gpu() pins the CALLING thread and
    what it starts afterwards: call it before the first backward / CPU tensor operation, or the autograd worker and the OpenMP pool that already
    exist stay where they were.  -> the restored set, or None."""
    global _affinity_before_pin, _threads_before_pin, _pinned_set
    if _affinity_before_pin is None or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        pinned = _pinned_set
        os.sched_setaffinity(0, _affinity_before_pin)
        # ... an OpenMP pool created by a CPU tensor operation while pinned keeps its eight CPUs otherwise: bench.py's 256-thread CPU baseline then
        # ran at 0.4x
        try:
            for tid in os.listdir("/proc/self/task"):
                try:
                    if pinned is not None and set(os.sched_getaffinity(int(tid))) == pinned:
                        os.sched_setaffinity(int(tid), _affinity_before_pin)
                except Exception:  # noqa: BLE001  (a thread that has just exited)
                    pass
        except Exception:  # noqa: BLE001
            pass
        try:
            torch.set_num_threads(max(1, _threads_before_pin or len(_affinity_before_pin)))
        except Exception:  # noqa: BLE001
            pass
        out, _affinity_before_pin, _threads_before_pin, _pinned_set = _affinity_before_pin, None, None, None
        return out
    except Exception:  # noqa: BLE001
        return None


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


def barrier(local_rank=None):
    if dist.is_initialized() and dist.get_world_size() > 1:
        if local_rank is not None and dist.get_backend() == "nccl":
            dist.barrier(device_ids=[local_rank])
        else:
            dist.barrier()


def max_over_ranks(x, device="cpu"):
    """MAX all-reduce of a python float (the timing protocol: the job is as slow as its slowest rank)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_views(n_views, rank, world):
    """Views (stereo pairs / novel cameras) are independent units: rank r renders views r, r+world, r+2*world, ..."""
    return list(range(rank, n_views, world))


class GradAllReducer:
    """Bucketed mean all-reduce of parameter gradients (the only exchange step of stage-2 training).

    The model has 5,144,408 fp32 parameters = 20.6 MB.  Two ways to run it:
      * overlap=False: everything after the backward -- `reducer()` flattens each bucket, all-reduces it, writes the mean back.  With the default
        32 MiB bucket that is ONE collective per step (xGMI is point-to-point, 7 links x ~153 GB/s, a ring all-reduce is per-link bound: fewer and
        larger messages win) -- and it is serial behind the whole backward.
      * overlap=True: the buckets are filled in REVERSE parameter order (the order autograd produces gradients in) and a post-accumulate hook on
        every parameter launches a bucket's all-reduce on a side stream the moment its last gradient exists, while the backward of the earlier
        layers is still running; `reducer()` then only launches what the hooks could not (buckets holding parameters that received no gradient:
        the reference constructs but never uses gru16 / gru32, core/update.py:105-106 -- they travel as zeros, so every rank issues identical
        collectives), waits, and writes the means back.  Smaller buckets (8 MiB: ~3 messages) give the overlap something to start early.
    Both produce the same numbers (tests/test_multiproc_gloo.py).

    Contract of overlap=True: collectives are ISSUED IN BUCKET ORDER on every rank whatever order autograd completes the buckets in
    (a bucket whose gradients are complete waits for the buckets in front of it), so ranks whose data-dependent branches finish parameters in a
    different order -- or leave some without a gradient -- still issue identical sequences (what the hooks could not start, `reducer()` starts, in
    the same order).  Which parameters a bucket waits for is LEARNT at the first `reducer()` call -- the parameters that received a gradient on
    ANY rank (one small MAX all-reduce of the mask; the reference's never-used gru16 / gru32 drop out, so their buckets do not hold the others
    back) -- and the first step therefore runs without overlap.  A bucket that receives gradients again after its all-reduce was started -- a second backward before `reducer()` (gradient
    accumulation), or a step whose `reducer()` call was skipped after an exception -- is marked dirty: `reducer()` waits for the stale collective,
    discards it and reduces the bucket again from the accumulated gradients.  Which buckets were launched by the hooks and which are dirty is
    AGREED across the ranks at the start of `reducer()` (MAX all-reduce on a control group; round 6), so accumulation combined with data-dependent
    parameter usage -- a rank that dirtied fewer buckets than another -- still issues identical sequences.  `reset()` drops all in-flight state explicitly (call it after a
    backward that raised, on EVERY rank)."""

    def __init__(self, params, bucket_bytes=None, overlap=False):
        self.params = [p for p in params if p.requires_grad]
        self.overlap = bool(overlap)
        if bucket_bytes is None:
            bucket_bytes = (8 << 20) if self.overlap else (32 << 20)
        order = list(reversed(self.params)) if self.overlap else list(self.params)
        self.buckets, cur, size = [], [], 0
        for p in order:
            n = p.numel() * p.element_size()
            if cur and size + n > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += n
        if cur:
            self.buckets.append(cur)
        self._ready = [0] * len(self.buckets)
        self._inflight = {}  # bucket index -> (flat tensor, work handle)
        self._dirty = set()  # buckets whose in-flight collective no longer holds their current gradients
        self._next = 0       # overlap: the next bucket the hooks may start (issue order = bucket order on every rank)
        self._expect = None  # overlap: per bucket, the ids of the parameters it waits for (learnt at the first call, identical on every rank)
        self._got = set()    # ids of the parameters whose gradient has arrived since the last call
        self._comm = None
        self._ctl = None     # overlap: control group on which the ranks agree on (launched, dirty) before reducer() issues anything
        self._hooks = []
        if self.overlap:
            where = {id(p): b for b, bucket in enumerate(self.buckets) for p in bucket}
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(lambda t, b=where[id(p)], k=id(p): self._on_grad(b, k)))

    def _active(self):
        return dist.is_initialized() and (dist.get_world_size() > 1 or forced())

    def _launch(self, b):
        bucket = self.buckets[b]
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        if flat.is_cuda and self.overlap:
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=flat.device)
            cur = torch.cuda.current_stream(flat.device)
            self._comm.wait_stream(cur)  # the flattened gradients are ready on the compute stream
            with torch.cuda.stream(self._comm):
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
            flat.record_stream(self._comm)
        else:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        self._inflight[b] = (flat, work)

    def _on_grad(self, b, key):
        if not self._active():
            return
        self._ready[b] += 1
        again = key in self._got
        self._got.add(key)
        if b in self._inflight:  # a gradient arrived AFTER this bucket's all-reduce was started (second backward / a skipped reducer() call /
            self._dirty.add(b)   # a parameter the bucket did not wait for)
            return
        if again or self._expect is None:  # accumulating into a bucket that has not left yet is fine; nothing starts before the first call
            return
        while self._next < len(self.buckets) and self._next not in self._inflight and self._expect[self._next] <= self._got:
            self._launch(self._next)  # in bucket order only: a completed bucket waits for the ones in front of it
            self._next += 1

    def reset(self):
        """Drop every in-flight collective and all hook state (waits for what was started, so that no rank leaves a collective half-issued)."""
        for b in sorted(self._inflight):
            self._inflight[b][1].wait()
        self._inflight.clear()
        self._dirty.clear()
        self._ready = [0] * len(self.buckets)
        self._got.clear()
        self._next = 0

    @torch.no_grad()
    def __call__(self):
        if not self._active():
            return
        world = dist.get_world_size()
        if self.overlap and self._expect is None:
            # learn what each bucket waits for: the parameters that received a gradient on ANY rank this step (identical on every rank afterwards)
            ref = self.params[0]
            mask = torch.tensor([0.0 if p.grad is None else 1.0 for p in self.params], dtype=torch.float32, device=ref.device)
            dist.all_reduce(mask, op=dist.ReduceOp.MAX)
            used = {id(p) for p, m in zip(self.params, mask.tolist()) if m > 0}
            self._expect = [{id(p) for p in bucket if id(p) in used} for bucket in self.buckets]
        if self.overlap:
            # What the hooks did is decided PER RANK (how far `_next` got, which buckets were dirtied), but every collective below must be issued by
            # every rank in the same order (a rank with a dirty bucket re-reduced it, a rank without one did not -- e.g. gradient
            # accumulation on a rank that also skipped a parameter -- and the j-th collectives of the two ranks were different buckets).  So the ranks
            # first AGREE: MAX over ranks of (buckets launched by hooks, dirty bit per bucket), on a control group of its own -- the ranks may have
            # issued different numbers of bucket collectives on the default group at this point, and collectives only pair up per communicator.
            nb = len(self.buckets)
            if self._ctl is None:
                self._ctl = dist.new_group()  # (collective: every rank reaches its first overlapped call)
            ref = self.params[0]
            state = torch.zeros(nb + 1, dtype=torch.float32, device=ref.device)
            state[0] = float(self._next)
            for b in self._dirty:
                state[1 + b] = 1.0
            dist.all_reduce(state, op=dist.ReduceOp.MAX, group=self._ctl)
            st = state.tolist()
            launched_any, dirty_any = int(st[0]), {b for b in range(nb) if st[1 + b] > 0}
            for b in range(self._next, launched_any):  # started by another rank's hooks: pair it up now (in bucket order: the same sequence everywhere)
                if b not in self._inflight:
                    self._launch(b)
            self._dirty = dirty_any
        for b in sorted(self._dirty):  # stale collectives: finish them (every rank issued them), throw the result away, reduce the bucket again
            if b in self._inflight:
                self._inflight.pop(b)[1].wait()
        self._dirty.clear()
        for b in range(len(self.buckets)):  # whatever the hooks did not start (overlap off; parameters without a gradient in the bucket; dirty ones)
            if b not in self._inflight:
                self._launch(b)
        for b, bucket in enumerate(self.buckets):
            flat, work = self._inflight.pop(b)
            work.wait()  # (CUDA: makes the current stream wait for the collective)
            flat.div_(world)
            off = 0
            for p in bucket:
                n = p.numel()
                if p.grad is None:
                    p.grad = torch.empty_like(p)
                p.grad.copy_(flat[off:off + n].view_as(p))
                off += n
        self._ready = [0] * len(self.buckets)
        self._got.clear()
        self._next = 0

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
