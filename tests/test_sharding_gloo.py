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


def _bench(*argv, env=None):
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), env=e, capture_output=True,
                       text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_bare_launch_spawns_ranks_and_shards_config4():
    """`python bench.py --gpus 2 --pairs 64 --batch 4` with NO launcher around it (how the driver invokes N = 1): bench.py
    must start `torch.distributed.run` as a child process itself, and the ranks must rendezvous on 127.0.0.1, broadcast
    the flat weight buffer, block-partition the 64 pairs, run exactly K steps for each and agree on MAX-over-ranks time.
    (--dry-run: gloo + a stub step; the same launcher / sharding / timing code runs over RCCL on GPUs.)"""
    r, d = _bench("--gpus", "2", "--pairs", "64", "--batch", "4", "--steps", "3", "--dry-run")
    assert r.returncode == 0, r.stdout + r.stderr
    assert d["dry_run"] and d["n_gpus"] == 2 and d["ranks_in_group"] == 2
    assert d["weights_equal_on_all_ranks"] and d["broadcast_bytes"] > 0
    assert d["pairs_covered_once"] and d["steps_per_pair"] == [3] and d["pairs"] == 64


def test_bench_launch_argument_errors():
    r, d = _bench("--gpus", "2", "--pairs", "6", "--batch", "4", "--steps", "1", "--dry-run")
    assert r.returncode != 0 and "multiple of ranks x batch" in (r.stdout + r.stderr)
    # a launcher that gave a different world size than --gpus is an error, not a silent N = 1 run
    r, d = _bench("--gpus", "1", "--dry-run", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stdout + r.stderr)


def test_plan_groups_matches_shard_range():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from bench import plan_groups
    seen = []
    for rank in range(8):
        g = plan_groups(64, rank, 8, 4)
        assert len(g) == 2 and all(len(x) == 4 for x in g)
        seen += [i for x in g for i in x]
    assert seen == list(range(64))


# ------------------------------------------------------------------------------------------------ adapter-gradient all-reduce
def _grad_worker(rank, world, port, q):
    """the data-parallel half of the training step (the only per-step collective of the reference is
    DDP's all-reduce of the adapter gradients): every rank fills the flat fp32 bucket with ITS gradients, one all-reduce
    sums it, the update uses the mean.  On CPU only the host logic runs (the clip + AdamW kernels need the GPU)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from i2v_adapter_unofficial_amd.i2v_adapter import I2VAdapterTransformerBlock
    from i2v_adapter_unofficial_amd.training import AdapterOptimizer

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.blocks = torch.nn.ModuleList([I2VAdapterTransformerBlock(32, 4, 8, cross_attention_dim=16) for _ in range(2)])
    torch.manual_seed(0)
    opt = AdapterOptimizer(Holder())
    g = torch.Generator().manual_seed(10 + rank)
    grads = {n: torch.randn(p.shape, generator=g) for n, p in zip(opt.names, opt.params)}
    opt.fill_gradients(grads)
    mine = opt.grad.clone()
    div = opt.reduce_gradients()
    # (numpy arrays travel by value: a tensor on the queue is a file descriptor the parent has to fetch while the child lives)
    q.put((rank, opt.names, mine.numpy().copy(), opt.grad.numpy().copy(), div, opt.master.numel()))
    dist.barrier()
    dist.destroy_process_group()


def test_adapter_gradient_allreduce_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, names0, mine0, red0, div0, n0), (_, names1, mine1, red1, div1, n1) = res
    mine0, red0, mine1, red1 = (torch.from_numpy(t) for t in (mine0, red0, mine1, red1))
    assert names0 == names1 and len(names0) == 2 * 3 and all(".i2v_adapter.to_q." in n or ".i2v_adapter.to_out." in n for n in names0)
    assert n0 == n1 == 2 * (32 * 32 + 32 * 32 + 32)            # to_q.weight, to_out.0.weight, to_out.0.bias per block
    assert div0 == div1 == 2
    assert not torch.equal(mine0, mine1)
    assert torch.equal(red0, red1) and torch.allclose(red0, mine0 + mine1)
