"""CPU, world_size 2, gloo: the multi-process protocol of the render path (view sharding, barrier + MAX timing as
bench.py does it, rank-0 JSON line) and the stage-2 gradient all-reduce helper."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import torch
    import gps_gaussian_amd
    from gps_gaussian_amd import dist as D

    rank, local_rank, world = D.init(backend="gloo")
    assert world == 2 and rank in (0, 1)
    # --- view sharding: disjoint, complete, balanced
    mine = D.shard_views(7, rank, world)
    import torch.distributed as dist
    gathered = [None, None]
    dist.all_gather_object(gathered, mine)
    allv = sorted(sum(gathered, []))
    assert allv == list(range(7)) and abs(len(gathered[0]) - len(gathered[1])) <= 1
    # --- timing protocol of bench.py: barrier, K steps, barrier, MAX over ranks
    D.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))          # rank 1 is the slow one
    D.barrier()
    el = D.max_over_ranks(time.perf_counter() - t0)
    assert el >= 0.1 - 1e-3
    # --- stage-2 exchange step: bucketed mean all-reduce, including a parameter that never gets a gradient
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4))
    unused = torch.nn.Linear(4, 4)         # like gru16/gru32 in the reference: constructed, never used
    params = list(net.parameters()) + list(unused.parameters())
    x = torch.full((3, 8), float(rank + 1))
    net(x).sum().backward()
    local = [p.grad.clone() for p in net.parameters()]
    red = D.GradAllReducer(params, bucket_bytes=300)   # tiny buckets -> several collectives
    assert len(red.buckets) > 1
    red()
    both = [None, None]
    dist.all_gather_object(both, [g.tolist() for g in local])
    for p, g0, g1 in zip(net.parameters(), both[0], both[1]):
        want = (torch.tensor(g0) + torch.tensor(g1)) / 2
        assert torch.allclose(p.grad, want, atol=1e-6)
    for p in unused.parameters():
        assert p.grad is not None and float(p.grad.abs().max()) == 0.0
    # --- the overlapped form: buckets in reverse parameter order, launched from post-accumulate hooks DURING the backward; same numbers
    serial = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    red2 = D.GradAllReducer(params, bucket_bytes=300, overlap=True)
    assert len(red2.buckets) > 1 and red2.buckets[0][0] is params[-1]
    net(x).sum().backward()
    assert not red2._inflight                   # the first step learns which parameters ever receive a gradient: nothing starts early
    red2()
    assert red2._expect is not None and set().union(*red2._expect) == {id(p) for p in net.parameters()}       # the unused layer is waited for by nobody
    for p, want in zip(params, serial):
        assert p.grad is not None and torch.equal(p.grad, want), (p.shape, (p.grad - want).abs().max())
    # a second iteration through the same hooks: now the collectives start DURING the backward, in bucket order
    for p in params:
        p.grad = None
    net(x).sum().backward()
    started = sorted(red2._inflight)            # collectives already in flight when backward() returns
    red2()
    assert len(started) >= 2 and started == list(range(len(started))) and not red2._inflight, started
    for p, want in zip(params, serial):
        assert torch.equal(p.grad, want)
    # --- ADVICE r04 hazards of the overlapped form ---------------------------------------------------------------------------------
    # (1) gradient accumulation: TWO backwards before reducer(): the collectives the hooks started after the first one are stale; the result must be
    #     the mean of the ACCUMULATED gradients
    for p in params:
        p.grad = None
    net(x).sum().backward()
    net(x).sum().backward()
    assert red2._dirty, "the second backward must mark the buckets already in flight"
    red2()
    for p, want in zip(params, serial):
        assert torch.allclose(p.grad, 2 * want, atol=1e-6), (p.shape, (p.grad - 2 * want).abs().max())
    assert not red2._dirty and not red2._inflight and red2._next == 0
    # (2) a step whose reducer() call was skipped (an exception between backward and the exchange): the next step must not write stale means back
    for p in params:
        p.grad = None
    net(3 * x).sum().backward()        # ... reducer() never called for this one; the caller drops the gradients and carries on
    for p in params:
        p.grad = None
    net(x).sum().backward()
    red2()
    for p, want in zip(params, serial):
        assert torch.allclose(p.grad, want, atol=1e-6)
    # explicit reset() after a failed step does the same
    for p in params:
        p.grad = None
    net(3 * x).sum().backward()
    red2.reset()
    assert not red2._inflight and red2._next == 0
    for p in params:
        p.grad = None
    net(x).sum().backward()
    red2()
    for p, want in zip(params, serial):
        assert torch.allclose(p.grad, want, atol=1e-6)
    # (3) data-dependent parameter usage: rank 1 also runs `unused` (rank 0 does not), so the ranks complete their buckets in different orders and
    #     rank 0 never completes some of them; every rank must still ISSUE the same sequence of collectives (no hang), result = mean over ranks
    for p in params:
        p.grad = None
    y = net(x)
    if rank == 1:
        y = unused(y)
    y.sum().backward()
    mine_g = [(p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for p in params]
    red2()
    both = [None, None]
    dist.all_gather_object(both, [g.tolist() for g in mine_g])
    for p, g0, g1 in zip(params, both[0], both[1]):
        assert torch.allclose(p.grad, (torch.tensor(g0) + torch.tensor(g1)) / 2, atol=1e-6)
    assert float(sum(p.grad.abs().sum() for p in unused.parameters())) > 0
    # (4) ADVICE r05: (1) and (3) COMBINED, on a reducer of its own: a trunk and two heads, every parameter EXPECTED (both ranks use both heads
    #     while the reducer learns).  Then two backwards before reducer() in which rank 0 skips head B.  Head B's parameters fill the FIRST
    #     bucket(s) (reverse parameter order), so rank 0's hooks never get past bucket 0 -- nothing launched, nothing dirty -- while rank 1's hooks
    #     launch every bucket in the first backward and dirty every one in the second.  Until round 6 each rank re-reduced ITS OWN dirty set: rank 1
    #     issued 2 N collectives, rank 0 issued N (hang, or buckets of different sizes paired up).  Now the ranks agree first.
    torch.manual_seed(1)
    trunk, head_a, head_b = torch.nn.Linear(8, 8), torch.nn.Linear(8, 4), torch.nn.Linear(8, 4)
    params3 = list(trunk.parameters()) + list(head_a.parameters()) + list(head_b.parameters())
    red3 = D.GradAllReducer(params3, bucket_bytes=150, overlap=True)
    assert len(red3.buckets) >= 3

    def step3(skip_b):
        h = trunk(x)
        out = head_a(h).sum()
        if not skip_b:
            out = out + head_b(h).sum()
        out.backward()

    step3(False); red3()                       # learns: every parameter is expected
    for p in params3:
        p.grad = None
    step3(False)
    assert len(red3._inflight) == len(red3.buckets)   # overlap is live
    red3()
    for p in params3:
        p.grad = None
    step3(rank == 0)
    step3(rank == 0)
    mine_g = [(p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for p in params3]
    mine_state = (red3._next, sorted(red3._dirty), sorted(red3._inflight))
    red3()                                     # (no other collective before it: rank 1 has bucket all-reduces in flight that rank 0 has not issued yet)
    state = [None, None]
    dist.all_gather_object(state, mine_state)
    assert state[0] != state[1], state         # the ranks really reached reducer() in different states
    assert state[0][0] == 0 and not state[0][1] and state[1][0] == len(red3.buckets) and state[1][1] == list(range(len(red3.buckets))), state
    both = [None, None]
    dist.all_gather_object(both, [g.tolist() for g in mine_g])
    for p, g0, g1 in zip(params3, both[0], both[1]):
        assert torch.allclose(p.grad, (torch.tensor(g0) + torch.tensor(g1)) / 2, atol=1e-6)
    assert not red3._dirty and not red3._inflight and red3._next == 0
    for p in params3:                          # ... and an ordinary step behind it: nothing is left half-issued on either rank
        p.grad = None
    step3(False)
    want3 = [p.grad.clone() for p in params3]  # (identical inputs on identical weights per rank; the mean over ranks follows below)
    red3()
    both = [None, None]
    dist.all_gather_object(both, [g.tolist() for g in want3])
    for p, g0, g1 in zip(params3, both[0], both[1]):
        assert torch.allclose(p.grad, (torch.tensor(g0) + torch.tensor(g1)) / 2, atol=1e-6)
    red3.remove_hooks()
    # ... and one more ordinary step behind it: nothing is left half-issued
    for p in params:
        p.grad = None
    net(x).sum().backward()
    red2()
    for p, want in zip(params, serial):
        assert torch.allclose(p.grad, want, atol=1e-6)
    red2.remove_hooks()
    # --- CPU affinity helper: every rank gets a non-empty slice of what it may run on; two ranks sharing one pool get disjoint slices when there are >= 2 CPUs
    before = sorted(os.sched_getaffinity(0))
    mine = D.set_cpu_affinity(rank, world)
    slices = [None, None]
    dist.all_gather_object(slices, sorted(mine) if mine else None)
    if len(before) >= 2:
        assert slices[0] and slices[1] and not (set(slices[0]) & set(slices[1])), slices
        assert sorted(os.sched_getaffinity(0)) == sorted(mine)
    os.sched_setaffinity(0, before)
    if rank == 0:
        print(json.dumps({"metric": "protocol", "n_gpus": world, "value": 2 / el, "views": allv}))
    D.shutdown()
""")


def test_two_rank_protocol_and_grad_allreduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(script)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # exactly one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["views"] == list(range(7))


def test_cpulist_parser_and_affinity_without_topology():
    import gps_gaussian_amd  # noqa: F401
    from gps_gaussian_amd import dist as D
    assert D._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11} and D._parse_cpulist("") == set()
    assert D.gpu_numa_cpus(0) is None or isinstance(D.gpu_numa_cpus(0), set)      # no GPU here: "the platform does not say"
    assert D.set_cpu_affinity(0, 1) is None                                        # one rank: left alone
    before = sorted(os.sched_getaffinity(0))
    try:
        env = os.environ.pop("GPSGS_AFFINITY", None)
        got = []
        for r in (0, 1):                                                            # two ranks (each its own process in real life), no topology:
            os.sched_setaffinity(0, before)                                         # an even split of what the process may run on
            got.append(D.set_cpu_affinity(r, 2))
        os.sched_setaffinity(0, before)
        if len(before) >= 2:
            assert got[0] and got[1] and not (got[0] & got[1]) and (got[0] | got[1]) <= set(before)
        if len(before) >= 8:                                                        # n_cpus: a rank that only drives its GPU narrows its slice to one L3 domain
            narrow = []
            for r in (0, 1):
                os.sched_setaffinity(0, before)
                narrow.append(D.set_cpu_affinity(r, 2, n_cpus=2))
            os.sched_setaffinity(0, before)
            assert all(n and len(n) == 2 for n in narrow) and narrow[0] <= got[0] and narrow[1] <= got[1]
        os.environ["GPSGS_AFFINITY"] = "0"
        assert D.set_cpu_affinity(0, 2) is None                                    # switched off
    finally:
        os.sched_setaffinity(0, before)
        os.environ.pop("GPSGS_AFFINITY", None)
        if env is not None:
            os.environ["GPSGS_AFFINITY"] = env


def test_pin_near_gpu_and_restore():
    """ONE rank driving one GPU through the autograd API: the calling thread is pinned to a few CPUs of one L3 domain (dist.pin_near_gpu) and can
    be given its original CPUs back (CPU-heavy legs).  Without a GPU / topology files the pool is whatever the process may run on."""
    import gps_gaussian_amd  # noqa: F401
    from gps_gaussian_amd import dist as D
    before = set(os.sched_getaffinity(0))
    env = os.environ.pop("GPSGS_AFFINITY", None)
    try:
        if len(before) >= 3:
            got = D.pin_near_gpu(0, 2)
            assert got is not None and len(got) == 2 and got <= before and set(os.sched_getaffinity(0)) == got
            assert D.pin_near_gpu(0, 2) is None                     # again, from the narrowed set: nothing left to choose from
            assert D.restore_affinity() == before and set(os.sched_getaffinity(0)) == before
            assert D.restore_affinity() is None                     # nothing to undo any more
        assert D.pin_near_gpu(0, len(before)) is None               # asks for everything there is: left alone
        os.environ["GPSGS_AFFINITY"] = "0"
        assert D.pin_near_gpu(0, 1) is None and set(os.sched_getaffinity(0)) == before
    finally:
        os.sched_setaffinity(0, before)
        os.environ.pop("GPSGS_AFFINITY", None)
        if env is not None:
            os.environ["GPSGS_AFFINITY"] = env


def test_full_pipeline_leg_respects_its_time_budget():
    """bench.py's `full_pipeline` leg (BASELINE configs 3 / 4 in child processes) must never run past its budget: with no time left every entry says it was
    skipped -- and why -- instead of starting a full-size network run; without a reference build it says so."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = bench.full_pipeline_leg(1.0)
    if "skipped" in out:                      # no reference staged in this checkout
        assert out["measured_in_this_run"] is False and "reference" in out["skipped"]
        return
    legs = [k for k in out if k.startswith("config")]
    assert len(legs) == 4
    for k in legs:
        assert out[k]["measured_in_this_run"] is False and "budget" in out[k]["skipped"], out[k]
    assert out["miopen_find_mode"] in ("FAST", os.environ.get("MIOPEN_FIND_MODE", "FAST"))
