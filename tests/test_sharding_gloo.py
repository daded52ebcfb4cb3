"""N > 1 path on CPU: world_size-2 gloo run of the start-up weight broadcast (one flat buffer, state-dict order)
and of the sample sharding / result gather.  The same code runs over RCCL on GPUs (bench.py --gpus N)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from i2v_adapter_unofficial_amd.sharding import broadcast_model_weights, gather_latents, shard_items
    from i2v_adapter_unofficial_amd.i2v_adapter import I2VAdapterModule
    torch.manual_seed(100 + rank)                      # different initial weights on every rank
    m = I2VAdapterModule(2, (32, 64, 64), 4)
    before = torch.cat([p.reshape(-1) for p in m.state_dict().values()]).clone()
    nbytes = broadcast_model_weights(m, src=0)
    after = torch.cat([p.reshape(-1) for p in m.state_dict().values()])
    mine = shard_items(list(range(5)), rank, world)
    lat = torch.full((1, 2, 4, 2, 2), float(rank))
    gathered = gather_latents(lat, dst=0)
    q.put((rank, before.sum().item(), after.sum().item(), nbytes, mine,
           None if gathered is None else [float(t.mean()) for t in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_shard_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, b0, a0, n0, s0, g0), (r1, b1, a1, n1, s1, g1) = res
    assert b0 != b1, "ranks must start from different weights for the test to mean anything"
    assert a0 == a1 == b0, "after the broadcast every rank holds rank 0's weights"
    assert n0 == n1 > 0
    assert s0 == [0, 1, 2] and s1 == [3, 4]
    assert g0 == [0.0, 1.0] and g1 is None
