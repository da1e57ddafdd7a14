"""N > 1 path of bench.py on CPU: two processes over gloo (127.0.0.1) run the contract's reduction -- every rank's
timings, MAX over ranks, whole-job rate = ranks x tokens / slowest time.  The GPU work itself is per-replica and has no
collective on the data path, so this is all the cross-rank logic there is."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    # rank r pretends its phases took (10 + r, 20 + 2r, 30 + 3r, 40 + 4r) ms
    local = [10.0 + rank, 20.0 + 2 * rank, 30.0 + 3 * rank, 40.0 + 4 * rank]
    red = bench.reduce_over_ranks(local, dist, "cpu")
    dist.barrier()
    if rank == 0:
        out.put(red)
    dist.destroy_process_group()


def test_rank_reduction_is_max_and_rates_are_whole_job():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, PORT, out)) for r in range(world)]
    for p in procs:
        p.start()
    red = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert red == [11.0, 22.0, 33.0, 44.0]                      # MAX over ranks of every phase
    import bench
    rates = bench.whole_job_rates(world, 2048, 128, red[1], red[2], red[3])
    assert abs(rates["prefill"] - world * 2048 / 0.022) < 1e-6
    assert abs(rates["worst"] - world * 128 / 0.033) < 1e-6
    assert abs(rates["best"] - world * 128 / 0.044) < 1e-6


def test_single_process_reduction_is_identity():
    import bench
    assert bench.reduce_over_ranks([1.0, 2.0], None, "cpu") == [1.0, 2.0]


PORT = _free_port()
