"""Multi-process (gloo, world_size 2) test of the image-sharding helpers the multi-GPU benchmark uses:
disjoint balanced shards, MAX-over-ranks timing, ragged gather of per-image counts.  The data path itself
has no collective (inference shards by image)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from yolo_amd.utils import dist as D
    w, r, _ = D.init_from_env("gloo")
    assert (w, r) == (world, rank)
    b, e = D.shard_range(n_items, r, w)
    # each rank "detects" on its shard: count = image index + 1
    counts = torch.arange(b, e, dtype=torch.int32) + 1
    D.barrier()
    slowest = D.max_over_ranks(1.0 + rank)
    gathered = D.gather_counts(counts)
    if rank == 0:
        out.put((slowest, [g.tolist() for g in gathered], (b, e)))
    D.barrier()
    dist.destroy_process_group()


def test_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_items = 7
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    slowest, gathered, shard0 = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert slowest == 2.0
    assert shard0 == (0, 4)
    assert gathered == [[1, 2, 3, 4], [5, 6, 7]]  # disjoint, complete, ordered


def test_shard_range_properties():
    from yolo_amd.utils.dist import shard_range
    for n in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [e - b for b, e in parts]
            assert max(sizes) - min(sizes) <= 1


def _bucket_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from yolo_amd.utils import dist as D
    D.init_from_env("gloo")
    n = 1000
    arena = torch.arange(n, dtype=torch.float32) * (rank + 1)  # rank r holds (r+1) * i
    bk = D.GradBucketer(arena, bucket_bytes=300 * 4)
    assert bk.mode == "auto" and bk.exchange == "allreduce" and not bk.direct   # the direct mesh exchange is an opt-in
    bk.start()
    for end in (100, 250, 320, 700, 990):   # "layers" finishing during backward
        bk.ready(end)
    bk.finish()
    if rank == 0:
        out.put((arena.tolist(), bk.launched))
    D.barrier()
    dist.destroy_process_group()


def test_grad_bucketer_world2():
    """Training's only exchange step: bucketed all-reduce (mean) of the flat gradient arena, launched while
    'backward' is still producing later ranges."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    arena, launched = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert arena == [1.5 * i for i in range(1000)]               # mean of 1*i and 2*i
    assert launched == [(0, 320), (320, 700), (700, 1000)]        # >= 300 elements per bucket, tail flushed


def _direct_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from yolo_amd.utils import dist as D
    D.init_from_env("gloo")
    n = 1000
    g = torch.Generator().manual_seed(rank)
    arena = torch.randn(n, generator=g)
    mine = arena.clone()
    bk = D.GradBucketer(arena, bucket_bytes=300 * 4, mode="direct")
    assert bk.direct
    for step in range(2):   # the receive buffers are reused by the second step
        arena.copy_(mine)
        bk.start()
        for end in (100, 250, 320, 700, 990):
            bk.ready(end)
        bk.finish()
    out.put((rank, mine.tolist(), arena.tolist(), bk.launched))
    D.barrier()
    dist.destroy_process_group()


def test_grad_bucketer_direct_exchange_world3():
    """The mesh form of the exchange (every rank sends shard j to rank j, sums its own shard, sends it back to all)
    on three ranks: the mean of the ranks' arenas, bit-identical on every rank, whole shards per bucket and the
    remainder (1000 is not a multiple of 3) flushed at the end."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_direct_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = torch.tensor([g[1] for g in got]).sum(0) / world
    results = [torch.tensor(g[2]) for g in got]
    assert all(torch.equal(results[0], r) for r in results[1:])          # bit-identical on every rank
    assert float((results[0] - want).abs().max()) < 1e-6
    assert got[0][3] == [(0, 318), (318, 699), (699, 999), (999, 1000)]  # whole shards; one element left for the flush
    assert all(g[3] == got[0][3] for g in got)


def test_grad_bucketer_single_process_is_a_noop():
    from yolo_amd.utils.dist import GradBucketer
    a = torch.ones(10)
    bk = GradBucketer(a, 16)
    bk.start()
    bk.ready(4)
    bk.ready(8)
    bk.finish()
    assert torch.equal(a, torch.ones(10)) and bk.launched == [(0, 4), (4, 8), (8, 10)]


def _bcast_worker(rank, world, port, q):
    import torch.distributed as dist
    from yolo_amd.utils import dist as D
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    ts = [torch.randn(3, 5), torch.randn(7).permute(0), torch.randint(0, 9, (4,)), torch.randn(2, 3, 4).permute(2, 0, 1)]
    D.broadcast_tensors(ts, src=0)
    q.put((rank, [t.clone().contiguous() for t in ts]))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_tensors_world2():
    """Flat per-dtype broadcast: every rank ends with rank 0's values, also in non-contiguous tensors."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    torch.manual_seed(100)
    want = [torch.randn(3, 5), torch.randn(7), torch.randint(0, 9, (4,)), torch.randn(2, 3, 4).permute(2, 0, 1)]
    for a, b, w in zip(out[0], out[1], want):
        assert torch.equal(a, b) and torch.equal(a, w.contiguous())
