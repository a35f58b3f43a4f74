"""N > 1 path on CPU: two gloo ranks shard a clip list exactly like eval/inference.py, never
exchange clip data, merge in rank order; and the flat-bucket gradient all-reduce (config 5)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from videotgb_amd import dist as vd
    clips = list(range(11))
    mine = vd.get_chunk(clips, world, rank)
    results = [(c, c * c) for c in mine]                 # stands for one generate() per clip
    merged = vd.gather_results(results)
    t = vd.max_over_ranks(1.0 + rank)
    torch.manual_seed(0)
    lin = torch.nn.Linear(4, 3)
    lin.weight.grad = torch.full_like(lin.weight, float(rank + 1))          # gradients that exist before the bucket: folded in
    lin.bias.grad = torch.full_like(lin.bias, 10.0 * (rank + 1))
    vd.FlatGradBucket(lin.parameters()).all_reduce(average=False)
    # the overlapped path: gradients are views of the flat buffer, segments go out from the backward hooks (last parameters first)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4))
    bucket = vd.FlatGradBucket(net.parameters(), segment_bytes=256)         # several segments
    views_ok = all(p.grad.data_ptr() == bucket.flat.data_ptr() + 4 * o for p, o in zip(bucket.params, bucket.offsets))
    x = torch.full((2, 8), float(rank + 1))
    ref = [torch.zeros_like(p) for p in net.parameters()]
    for r_ in range(world):                                                 # what the mean over ranks must be
        net.zero_grad(set_to_none=False)
        net(torch.full((2, 8), float(r_ + 1))).sum().backward()
        for g, p in zip(ref, net.parameters()):
            g += p.grad / world
    bucket.zero_()
    bucket.arm(average=True)
    net(x).sum().backward()
    launched_in_backward = sum(bucket._launched)
    bucket.all_reduce(average=True)
    overlap_ok = all(torch.allclose(p.grad, g, rtol=1e-6, atol=1e-6) for p, g in zip(net.parameters(), ref)) and views_ok \
        and all(p.grad.data_ptr() == bucket.flat.data_ptr() + 4 * o for p, o in zip(bucket.params, bucket.offsets))
    q.put((rank, list(mine), merged, t, lin.weight.grad[0, 0].item(), lin.bias.grad[0].item(), overlap_ok, launched_in_backward, len(bucket.segments)))
    dist.destroy_process_group()


def test_two_rank_clip_sharding_and_grad_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (r0, c0, m0, t0, w0, b0, ok0, l0, ns0), (r1, c1, m1, t1, w1, b1, ok1, l1, ns1) = out
    assert ok0 and ok1                                                   # views, mean over ranks, still views afterwards
    assert ns0 > 1 and l0 == ns0 and l1 == ns1                           # every segment left from a backward hook, not from all_reduce()
    assert c0 == [0, 1, 2, 3, 4, 5] and c1 == [6, 7, 8, 9, 10]          # ceil(11/2) = 6 per chunk, contiguous
    assert m0 == m1 == [(c, c * c) for c in range(11)]                  # rank-order merge == `cat` of the shell driver
    assert t0 == t1 == 2.0                                              # slowest rank
    assert w0 == w1 == 3.0 and b0 == b1 == 30.0                         # sum over ranks, one flat collective


def _uneven_worker(rank, world, port, q):
    """Rank 1's LAST layer (= the first segments in reverse order) receives no gradient; a parameter's hook fires twice on rank 0."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from videotgb_amd import dist as vd
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 16), torch.nn.Linear(16, 4))
    bucket = vd.FlatGradBucket(net.parameters(), segment_bytes=128)
    x = torch.full((2, 8), float(rank + 1))
    bucket.zero_()
    bucket.arm(average=False)
    order = []
    orig = bucket._launch
    bucket._launch = lambda si: (order.append(si), orig(si))[1]
    if rank == 0:
        net(x).sum().backward()
        bucket._on_grad(net[2].bias)                                 # a second firing of an already counted hook: must not launch anything early
    else:
        h = net[1](net[0](x))                                            # the last layer takes no part on this rank: its gradients stay zero
        h.sum().backward()
    bucket.all_reduce(average=False)
    # expected sums, computed locally
    ref = [torch.zeros_like(p) for p in net.parameters()]
    for r_ in range(world):
        net2 = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 16), torch.nn.Linear(16, 4))
        net2.load_state_dict(net.state_dict())
        xx = torch.full((2, 8), float(r_ + 1))
        (net2(xx).sum() if r_ == 0 else net2[1](net2[0](xx)).sum()).backward()
        for g, p in zip(ref, net2.parameters()):
            if p.grad is not None:
                g += p.grad
    ok = all(torch.allclose(p.grad, g, rtol=1e-6, atol=1e-6) for p, g in zip(net.parameters(), ref))
    q.put((rank, ok, order, len(bucket.segments)))
    dist.destroy_process_group()


def test_flat_bucket_launches_segments_in_index_order_on_every_rank():
    """Round-3 ADVICE: collectives pair by issue order -- a rank with a gradient-less parameter must issue the same sequence."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uneven_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (_, ok0, o0, n0), (_, ok1, o1, n1) = out
    assert ok0 and ok1
    assert n0 == n1 and n0 > 2 and o0 == o1 == list(range(n0))           # same sequence on both ranks, every segment exactly once


def _ragged_worker(rank, world, port, q):
    """world = 4, 5 clips: chunks of ceil(5 / 4) = 2 -> ranks 0, 1 hold 2 clips, rank 2 one, rank 3 NONE (get_chunk past the end).  Rank 3 runs no
    forward / backward at all -- every segment of its bucket is still empty when all_reduce() is called -- and rank 2's single clip skips the last
    layer (its last segment stays empty): both must issue the same collective sequence as the ranks whose hooks fired."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from videotgb_amd import dist as vd
    clips = list(range(5))
    mine = list(vd.get_chunk(clips, world, rank))
    merged = vd.gather_results([(c, 10 * c) for c in mine])
    torch.manual_seed(0)
    make = lambda: torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 16), torch.nn.Linear(16, 4))
    net = make()
    bucket = vd.FlatGradBucket(net.parameters(), segment_bytes=128)
    order = []
    orig = bucket._launch
    bucket._launch = lambda si: (order.append(si), orig(si))[1]

    def loss_of(model, r_, chunk):
        if not chunk:
            return None
        x = torch.stack([torch.full((8,), float(c + 1)) for c in chunk])
        return (model[1](model[0](x)) if r_ == 2 else model(x)).sum()
    bucket.zero_()
    bucket.arm(average=True)
    l = loss_of(net, rank, mine)
    if l is not None:
        l.backward()
    bucket.all_reduce(average=True)
    ref = [torch.zeros_like(p) for p in net.parameters()]
    for r_ in range(world):
        n2 = make()
        n2.load_state_dict(net.state_dict())
        l2 = loss_of(n2, r_, list(vd.get_chunk(clips, world, r_)))
        if l2 is not None:
            l2.backward()
        for g, p in zip(ref, n2.parameters()):
            if p.grad is not None:
                g += p.grad / world
    ok = all(torch.allclose(p.grad, g, rtol=1e-6, atol=1e-6) for p, g in zip(net.parameters(), ref))
    q.put((rank, mine, merged, ok, order, len(bucket.segments)))
    dist.destroy_process_group()


def test_four_ranks_ragged_clip_counts_and_empty_segments():
    """SCALE readiness without hardware: 4 gloo ranks, more ranks than full chunks, a rank with nothing to do."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert [o[1] for o in out] == [[0, 1], [2, 3], [4], []]
    n_seg = out[0][5]
    for rank, mine, merged, ok, order, ns in out:
        assert merged == [(c, 10 * c) for c in range(5)]                 # rank-order merge, the empty rank contributes nothing
        assert ok                                                        # mean over the 4 ranks, zeros from the idle rank / the skipped layer
        assert ns == n_seg and order == list(range(n_seg))               # the same collective sequence on every rank, each segment once


def test_flat_bucket_rejects_non_fp32_parameters_and_disagreeing_average():
    from videotgb_amd import dist as vd
    lin = torch.nn.Linear(4, 4).to(torch.bfloat16)
    with pytest.raises(TypeError, match="fp32"):
        vd.FlatGradBucket(lin.parameters())
    b = vd.FlatGradBucket(torch.nn.Linear(4, 4).parameters())
    b.arm(average=True)
    with pytest.raises(ValueError, match="disagree"):
        b.all_reduce(average=False)


def test_split_list_matches_reference_semantics():
    from videotgb_amd.dist import get_chunk, split_list
    assert split_list(list(range(10)), 4) == [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9]]
    assert split_list(list(range(8)), 8) == [[i] for i in range(8)]
    assert get_chunk(list(range(3)), 8, 5) == []                         # more ranks than clips


def _nccl_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from videotgb_amd import dist as vd
    dev = torch.device("cuda", rank)
    lin = torch.nn.Linear(1024, 512).to(dev)
    lin.weight.grad = torch.full_like(lin.weight, float(rank + 1))
    lin.bias.grad = torch.full_like(lin.bias, 10.0 * (rank + 1))
    vd.FlatGradBucket(lin.parameters()).all_reduce(average=True)        # DDP semantics: mean over ranks, one RCCL collective
    t = vd.max_over_ranks(1.0 + rank, device=dev)
    torch.cuda.synchronize()
    q.put((rank, lin.weight.grad[3, 7].item(), lin.bias.grad[5].item(), t))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_flat_bucket_allreduce_over_rccl_when_two_gpus_are_visible():
    """The C5 gradient exchange on the real backend ("nccl" == RCCL over xGMI): runs wherever >= 2 GPUs are visible (the
    1-GPU gpurun boxes skip it; the driver's multi-GPU node runs it)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 visible GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, w, b, t in out:
        assert w == 1.5 and b == 15.0 and t == 2.0
