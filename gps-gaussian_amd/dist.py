"""One-process-per-GPU plumbing for the render path (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm).

The path shards by view: every rank renders its own stereo pairs and there is NO data-path collective.
What is shared is only (a) the launch protocol bench.py and the driver use (env rendezvous, barrier, MAX-over-ranks
timing) and (b) for stage-2 training, one bucketed gradient all-reduce of the network parameters per step -- the
reference has no distributed code at all (train_stage2.py:27-55 is single process), so this is new glue around its
unmodified Trainer, not a translation of anything.  Everything here also runs on CPU with the gloo backend
(tests/test_multiproc_gloo.py, world_size 2).
"""
import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def forced():
    """GPSGS_DIST_FORCE=1: initialise the process group and issue the exchange-step collectives even at world size 1 -- the way a 1-GPU box
    exercises RCCL itself (communicator set-up + a 20.6 MB all-reduce kernel) instead of skipping it."""
    return os.environ.get("GPSGS_DIST_FORCE") == "1"


def init(backend=None, device=None):
    """Initialise the default process group from the torchrun environment.  Returns (rank, local_rank, world)."""
    rank, local_rank, world = env_rank()
    if (world > 1 or forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


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

    The model has 5,144,408 fp32 parameters = 20.6 MB: with the default 32 MiB bucket that is ONE
    collective per step -- xGMI is point-to-point (7 links x ~153 GB/s), a ring all-reduce is per-link bound, so fewer and
    larger messages win.  Parameters that received no gradient (the reference constructs but never uses gru16/gru32,
    core/update.py:105-106) are treated as zeros so that every rank issues identical collectives."""

    def __init__(self, params, bucket_bytes=32 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.buckets, cur, size = [], [], 0
        for p in self.params:
            n = p.numel() * p.element_size()
            if cur and size + n > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += n
        if cur:
            self.buckets.append(cur)

    @torch.no_grad()
    def __call__(self):
        if not (dist.is_initialized() and (dist.get_world_size() > 1 or forced())):
            return
        world = dist.get_world_size()
        for bucket in self.buckets:
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.div_(world)
            off = 0
            for p in bucket:
                n = p.numel()
                if p.grad is None:
                    p.grad = torch.empty_like(p)
                p.grad.copy_(flat[off:off + n].view_as(p))
                off += n
