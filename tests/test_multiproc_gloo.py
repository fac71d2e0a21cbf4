"""CPU, world_size 2 over gloo: the N>1 leg of bench.py is N independent replicas whose step time is the MAX
over ranks and whose value is the sum of the units all ranks processed / that time."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    entry.load_package()
    from cuda_learn_notes_amd import bench_utils as bu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = 0.010 * (rank + 1)  # rank 1 is the slow replica
    dist.barrier()
    t = bu.max_over_ranks(mine, dist, torch.device("cpu"))
    v = bu.aggregate_value(100.0, world, t)
    q.put((rank, t, v))
    dist.barrier()
    dist.destroy_process_group()


def test_two_replicas_time_is_max_and_value_is_aggregate():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, t, v in res:
        assert abs(t - 0.020) < 1e-12
        assert abs(v - 2 * 100.0 / 0.020) < 1e-6


def test_single_process_passthrough():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    entry.load_package()
    from cuda_learn_notes_amd import bench_utils as bu
    assert bu.max_over_ranks(0.5) == 0.5
    assert bu.aggregate_value(10.0, 1, 0.5) == 20.0
