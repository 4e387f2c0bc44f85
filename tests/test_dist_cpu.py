"""world_size-2 gloo test of the N>1 path: shard_range partitions, max-over-ranks timing, ordered gather."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from yvhip.dist import BucketReducer, gather_objects, max_over_ranks, shard_range


def test_shard_range_partitions():
    for n in (0, 1, 7, 32, 33, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(33, rank, world)
    local = [f"img{i:03d}" for i in range(lo, hi)]           # each rank "processes" its slice independently
    t = max_over_ranks(1.0 + rank)                            # slowest rank defines the step time
    allr = gather_objects(local)
    # gradient buckets: entries become final from the end; every rank launches identical slices
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    red = BucketReducer(flat, 256)
    for low in (900, 700, 512, 100):
        red.ready(low)
    assert red.launched == [(744, 1000)] or red.launched[0] == (744, 1000)
    red.finish()
    assert [b for b in red.launched] == [(744, 1000), (488, 744), (232, 488), (0, 232)]
    assert torch.equal(flat, torch.arange(1000, dtype=torch.float32) * 3)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, t, [x for part in allr for x in part]))


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, t, merged in res:
        assert t == 2.0
        assert merged == [f"img{i:03d}" for i in range(33)]


def test_union_length():
    """Interval union used for the roofline of concurrently launched kernels."""
    from yvhip.dist import union_length
    assert union_length([]) == 0.0
    assert union_length([(0.0, 1.0)]) == 1.0
    assert union_length([(0.0, 1.0), (2.0, 3.5)]) == 2.5                       # disjoint
    assert union_length([(2.0, 3.0), (0.0, 2.5), (2.9, 4.0)]) == 4.0           # overlapping, unsorted
    assert union_length([(0.0, 5.0), (1.0, 2.0), (3.0, 4.0)]) == 5.0           # nested


def test_bench_gpus_flag_launches_ranks():
    """`python bench.py --gpus N` without a launcher starts N rank processes itself (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, same command line); YV_BENCH_DRY makes every rank report its layout instead of touching a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, YV_BENCH_DRY="1", YV_BENCH_REHEARSAL="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "2"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = sorted((json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")), key=lambda d: d["rank"])
    assert [(d["rank"], d["local"], d["world"], d["gpus"]) for d in lines] == [(0, 0, 3, 3), (1, 1, 3, 3), (2, 2, 3, 3)]
    # without the rehearsal switch a box with fewer devices refuses instead of silently running one rank
    env.pop("YV_BENCH_REHEARSAL")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "device(s) visible" in r.stderr
    # a launcher-provided world size must agree with the flag
    env.update(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "disagrees" in r.stderr
