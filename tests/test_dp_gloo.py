"""World-size-2 `gloo` tests (CPU) of the batch-parallel machinery in nonauto_lm_amd/dp.py:
bucketed gradient all-reduce from grad hooks == single-process gradient of the full batch,
unused parameters contribute zeros, clipping matches torch's utilities, packed scalar reduction."""
import os
import socket
import sys

import pytest
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


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.body = torch.nn.Sequential(torch.nn.Linear(12, 40), torch.nn.Tanh(), torch.nn.Linear(40, 40),
                                        torch.nn.Tanh(), torch.nn.Linear(40, 3))
        self.unused = torch.nn.Linear(5, 5)   # never touched by forward (find_unused_parameters case)

    def forward(self, x):
        return self.body(x)


def _make_model():
    torch.manual_seed(13)
    return _Net()


def _data():
    g = torch.Generator().manual_seed(7)
    return torch.randn(16, 12, generator=g), torch.randn(16, 3, generator=g)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nonauto_lm_amd import dp
    torch.set_num_threads(1)
    model = _make_model()
    if rank == 1:                              # perturb, then check broadcast restores rank 0's weights
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    dp.broadcast_parameters(model)
    reducer = dp.GradReducer(model.parameters(), bucket_bytes=4096)      # several buckets
    x, y = _data()
    shard = slice(rank * 8, (rank + 1) * 8)
    out = {}
    for it in range(2):                         # two steps: state resets between steps
        loss = ((model(x[shard]) - y[shard]) ** 2).mean()
        loss.backward()
        reducer.finish()
        out[f"grads{it}"] = [p.grad.clone() for p in model.parameters()]
        norm = reducer.clip_(0.05, 0.01)
        out[f"norm{it}"] = norm.clone()
        out[f"clipped{it}"] = [p.grad.clone() for p in model.parameters()]
        reducer.zero_grad()
    sc = dp.PackedScalars(["a", "b"], "cpu")
    sc.reduce({"a": torch.tensor(float(rank + 1)), "b": 10.0 * (rank + 1)}, count=1.0)
    out["means"] = sc.means()
    out["nbuckets"] = len(reducer.buckets)
    # plain numpy payloads: torch tensors would travel as shared-memory handles that die with the child
    def plain(v):
        if isinstance(v, torch.Tensor):
            return v.numpy().copy()
        if isinstance(v, list):
            return [plain(x) for x in v]
        return v
    q.put((rank, {k: plain(v) for k, v in out.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_grad_reducer_world2_matches_full_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    def unplain(v):
        import numpy as np
        if isinstance(v, np.ndarray):
            return torch.from_numpy(v)
        if isinstance(v, list):
            return [unplain(x) for x in v]
        return v
    res = {}
    for _ in range(2):
        rank, out = q.get(timeout=100)
        res[rank] = {k: unplain(v) for k, v in out.items()}
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    # single-process reference on the full batch
    model = _make_model()
    x, y = _data()
    # mean over the full batch == average of the two shard means (equal shard sizes)
    ((model(x) - y) ** 2).mean().backward()
    ref = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in model.parameters()]
    for rank in (0, 1):
        r = res[rank]
        assert r["nbuckets"] >= 2
        for it in range(2):
            for a, b in zip(r[f"grads{it}"], ref):
                assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
        # clipping == torch utilities applied to the reference gradient
        for p, g in zip(model.parameters(), ref):
            p.grad = g.clone()
        n = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.05)
        torch.nn.utils.clip_grad_value_(model.parameters(), 0.01)
        assert torch.allclose(r["norm0"], n, rtol=1e-5)
        for a, p in zip(r["clipped0"], model.parameters()):
            assert torch.allclose(a, p.grad, rtol=1e-4, atol=1e-8)
        assert r["means"] == {"a": 1.5, "b": 15.0}
    for a, b in zip(res[0]["grads1"], res[1]["grads1"]):
        assert torch.equal(a, b)                # ranks stay bit-identical


def test_grad_reducer_single_process_is_identity():
    sys.path.insert(0, ROOT)
    from nonauto_lm_amd import dp
    model = _make_model()
    x, y = _data()
    reducer = dp.GradReducer(model.parameters(), bucket_bytes=1 << 20)
    ((model(x) - y) ** 2).mean().backward()
    reducer.finish()
    got = [p.grad.clone() for p in model.parameters()]
    model2 = _make_model()
    ((model2(x) - y) ** 2).mean().backward()
    for a, p in zip(got, model2.parameters()):
        assert torch.equal(a, p.grad if p.grad is not None else torch.zeros_like(p))
    reducer.zero_grad()
    assert all(float(p.grad.abs().max()) == 0.0 for p in model.parameters())


# ---------------------------------------------------------------------------- learned arrival counts, sink notifications
def _sink_worker(rank, world, port, q):
    """Gradient sinks (kernels that add into .grad behind autograd's back) simulated on the CPU: half of the parameters
    get their gradient through `notify_sink` instead of a grad hook.  Step 1 learns the per-bucket contribution counts,
    steps 2-3 launch every bucket as soon as its last contribution arrived (overlap), with the bucket's deferred work
    folded by `before_launch` first; fold_average leaves the SUM for the optimizer."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nonauto_lm_amd import dp
    torch.set_num_threads(1)
    model = _make_model()
    dp.broadcast_parameters(model)
    reducer = dp.GradReducer(model.parameters(), bucket_bytes=4096, flatten_params=True)
    reducer.fold_average = True
    params = [p for p in model.parameters()]
    sinked = [p for i, p in enumerate(params) if i % 2 == 1]           # "sink" parameters: biases
    x, y = _data()
    shard = slice(rank * 8, (rank + 1) * 8)
    deferred, folded, launches = [], [], []
    def before_launch(lo, hi):                                          # the bucket's deferred contributions
        launches.append((lo, hi))
        for item in [d for d in deferred if lo <= d[0].data_ptr() < hi]:
            item[0].add_(item[1])
            folded.append(item)
            deferred.remove(item)
    reducer.before_launch = before_launch
    out = {"early": []}
    for it in range(3):
        launches.clear()
        loss = ((model(x[shard]) - y[shard]) ** 2).mean()
        # autograd for the hooked half; the sinked half is computed on the side and written straight into .grad
        hooked = [p for p in params if all(p is not s for s in sinked) and p.requires_grad]
        gs = torch.autograd.grad(loss, sinked, retain_graph=True, allow_unused=True)
        loss.backward(inputs=[p for p in hooked if p is not model.unused.weight and p is not model.unused.bias])
        early = len(launches)
        for p, g in zip(sinked, gs):
            if g is None:
                continue
            if p.ndim == 1 and p.numel() == 3:                          # one contribution is deferred to the bucket flush
                deferred.append((p.grad, g))
            else:
                p.grad.add_(g)
            reducer.notify_sink(p.grad)
        out["early"].append((early, len(launches)))
        reducer.finish()
        assert not deferred
        out[f"grads{it}"] = [p.grad.clone() / world for p in params]    # fold_average: the buckets hold the SUM
        reducer.zero_grad()
    out["expected"] = list(reducer._expected)
    def plain(v):
        if isinstance(v, torch.Tensor):
            return v.numpy().copy()
        if isinstance(v, list):
            return [plain(x) for x in v]
        return v
    q.put((rank, {k: plain(v) for k, v in out.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_grad_reducer_sinks_and_learned_counts_world2():
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sink_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, out = q.get(timeout=100)
        res[rank] = out
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    model = _make_model()
    x, y = _data()
    ((model(x) - y) ** 2).mean().backward()
    ref = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in model.parameters()]
    for rank in (0, 1):
        r = res[rank]
        for it in range(3):
            for a, b in zip(r[f"grads{it}"], ref):
                assert torch.allclose(torch.from_numpy(np.asarray(a)), b, rtol=1e-5, atol=1e-7)
        assert sum(r["expected"]) > 0
        # step 1 learns (nothing before finish()); steps 2, 3 launch buckets from the arrivals
        assert r["early"][0][1] == 0
        assert r["early"][1][1] >= 1 and r["early"][2][1] >= 1
    for a, b in zip(res[0]["grads2"], res[1]["grads2"]):
        assert np.array_equal(a, b)


def _straddle_worker(rank, world, port, q):
    """a combined [dW | db] fold job whose two destinations sit in DIFFERENT buckets (bias at the end of
    bucket i, weight at the start of bucket i + 1).  The jobs are selected by the product's own
    ops.deferred_jobs_in_range; the fold itself (a HIP kernel in the product) is emulated with torch adds."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nonauto_lm_amd import dp, ops
    torch.set_num_threads(1)
    model = _make_model()
    dp.broadcast_parameters(model)
    # reversed registration order: unused.bias, unused.weight, body.4.bias, body.4.weight, body.2.bias (772 B) |
    # body.2.weight (6400 B), body.0.bias (160 B) | body.0.weight: both wide layers straddle a bucket boundary, and bucket 0
    # (holding body.2.bias) is complete BEFORE bucket 1 (body.2.weight + body.0.bias, which arrives later): with a
    # weight-only selection bucket 0 would be reduced without d b and receive it after its all-reduce
    reducer = dp.GradReducer(model.parameters(), bucket_bytes=6600)
    reducer.fold_average = True
    layers = [model.body[2], model.body[0]]          # the order backward reaches them
    out = {"straddle": [reducer._bucket_of[id(l.weight)] != reducer._bucket_of[id(l.bias)] for l in layers],
           "early": []}
    defer = {}
    launches = []

    def before_launch(lo, hi):
        launches.append((lo, hi))
        for k in ops.deferred_jobs_in_range(defer, lo, hi):
            part, g0, g1, split = defer.pop(k)
            s = part.sum(0)
            g0.view(-1).add_(s[:split])
            g1.view(-1).add_(s[split:])
    reducer.before_launch = before_launch
    x, y = _data()
    shard = slice(rank * 8, (rank + 1) * 8)
    sinked = [p for l in layers for p in (l.weight, l.bias)]
    hooked = [model.body[4].weight, model.body[4].bias]
    for it in range(3):
        launches.clear()
        loss = ((model(x[shard]) - y[shard]) ** 2).mean()
        gs = torch.autograd.grad(loss, sinked, retain_graph=True)
        loss.backward(inputs=hooked)
        for i, l in enumerate(layers):
            gw, gb = gs[2 * i], gs[2 * i + 1]
            flat = torch.cat([gw.reshape(-1), gb.reshape(-1)])
            defer[i] = (torch.stack([0.25 * flat, 0.75 * flat]), l.weight.grad, l.bias.grad, gw.numel())
            reducer.notify_sink(l.weight.grad)       # ops.sink_arrived(gw, gb) of the product
            reducer.notify_sink(l.bias.grad)
        out["early"].append(len(launches))
        reducer.finish()
        assert not defer
        out[f"grads{it}"] = [(p.grad.clone() / world).numpy().copy() for p in model.parameters()]
        reducer.zero_grad()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_combined_fold_job_straddling_two_buckets_world2():
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_straddle_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, out = q.get(timeout=100)
        res[rank] = out
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    model = _make_model()
    x, y = _data()
    ((model(x) - y) ** 2).mean().backward()
    ref = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in model.parameters()]
    for rank in (0, 1):
        r = res[rank]
        assert r["straddle"] == [True, True]
        assert r["early"][0] == 0 and r["early"][1] >= 1       # steps 2, 3 launch buckets from the arrivals
        for it in range(3):
            for a, b in zip(r[f"grads{it}"], ref):
                assert torch.allclose(torch.from_numpy(np.asarray(a)), b, rtol=1e-5, atol=1e-7)
    for a, b in zip(res[0]["grads2"], res[1]["grads2"]):
        assert np.array_equal(a, b)


def test_grad_reducer_late_contribution_raises():
    """world > 1 only matters for the exchange; the guard itself is rank-count independent: simulate world 2."""
    sys.path.insert(0, ROOT)
    from nonauto_lm_amd import dp
    model = _make_model()
    reducer = dp.GradReducer(model.parameters(), bucket_bytes=1 << 20)
    x, y = _data()
    ((model(x) - y) ** 2).mean().backward()
    reducer.finish()                                  # learns the counts
    reducer.zero_grad()
    ((model(x) - y) ** 2).mean().backward()           # bucket complete -> launched from the last hook
    assert all(reducer._launched)
    reducer.world = 2                                 # pretend the bucket went out to another rank
    with pytest.raises(RuntimeError, match="after its bucket was all-reduced"):
        reducer.notify_sink(next(model.parameters()).grad)


def test_packed_scalars_template_and_tensors():
    sys.path.insert(0, ROOT)
    from nonauto_lm_amd import dp
    sc = dp.PackedScalars(["a", "b", "c"], "cpu")
    for k in range(3):
        sc.reduce({"a": torch.tensor(2.0 + k), "b": 5.0, "c": torch.tensor(1.0)}, count=2.0)
        assert sc.means() == {"a": (2.0 + k) / 2, "b": 2.5, "c": 0.5}
    sc.reduce({"a": 1.0, "b": 7.0, "c": torch.tensor(3.0)}, count=1.0)
    assert sc.means() == {"a": 1.0, "b": 7.0, "c": 3.0}


@pytest.mark.timeout(120)
def test_spawn_local_ranks_env_and_failure(tmp_path):
    """dp.spawn_local_ranks (what `bench.py --gpus N` uses to start its ranks): every child gets RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_*; the children can rendezvous over gloo; a failing rank's exit code is returned and the
    surviving ranks are terminated."""
    sys.path.insert(0, ROOT)
    from nonauto_lm_amd import dp
    script = tmp_path / "child.py"
    script.write_text(
        "import os, sys, torch, torch.distributed as dist\n"
        "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "dist.init_process_group('gloo', rank=r, world_size=w)\n"
        "t = torch.tensor([float(r + 1)]); dist.all_reduce(t)\n"
        "open(sys.argv[1] + f'/rank{r}', 'w').write(str(t.item()))\n"
        "dist.destroy_process_group()\n")
    assert dp.spawn_local_ranks([sys.executable, str(script), str(tmp_path)], 2) == 0
    assert (tmp_path / "rank0").read_text() == "3.0" and (tmp_path / "rank1").read_text() == "3.0"
    bad = tmp_path / "bad.py"
    bad.write_text("import os, sys, time\n"
                   "if os.environ['RANK'] == '1': sys.exit(7)\n"
                   "time.sleep(60)\n")
    assert dp.spawn_local_ranks([sys.executable, str(bad)], 2) != 0
