"""One rank of tests/test_dp_nccl.py (also runnable alone with WORLD_SIZE unset: a 1-rank self-check).

Every rank trains the same small NonAutoModel (ActNorm + NICE flow prior) on ITS shard of a fixed batch through
GradReducer(flatten_params=True) + FusedAdam + TrainStep (HIP-graph replay, RCCL all-reduce captured in graph A when
RCCL allows it).  Checks, each written to <outdir>/rank<r>.json:
  * the averaged gradient of the first step == the gradient of the full batch on one rank (same injected noise),
    including the pooled ActNorm data initialisation (flows.py ActNorm._data_init);
  * after 3 steps every rank holds bit-identical parameters."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(M, D=32, H=32):
    import bench
    cfg = bench.model_config(D, H, 2)
    cfg["prior"]["flows"] = [dict(type="actnorm", input_size=D)] + cfg["prior"]["flows"]
    torch.manual_seed(13)
    model = M.VAELmModel.from_params(vocab=M.SimpleVocab(200), **cfg).cuda().train()
    model.training_outputs_only = True
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return model


def main():
    outdir = sys.argv[1]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dev = torch.device("cuda", torch.cuda.current_device())
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import bench
    import nonauto_lm_amd as M
    from nonauto_lm_amd import dp
    from nonauto_lm_amd.step import TrainStep
    B, S, D = 16 * world, 12, 32
    tokens = bench.synthetic_tokens(B, S, False, 7)
    tokens = torch.where(tokens > 0, tokens % 196 + 4, tokens).to(dev)
    eps = torch.randn(B, S, D, generator=torch.Generator().manual_seed(3)).to(dev)
    shard = slice(rank * 16, (rank + 1) * 16)
    res = {"rank": rank, "world": world}

    # ---- reference: the full batch on this rank alone, plain eager autograd (no process group involved)
    ref = build(M)
    if world > 1:                                   # keep ActNorm's pooled init out of the reference
        import nonauto_lm_amd.flows as F
        saved = F.dist.is_initialized
        F.dist.is_initialized = lambda: False
    out = ref(tokens, tokens, eps=eps)
    if world > 1:
        F.dist.is_initialized = saved
    out["loss_info"]["batch-loss"].backward()
    gref = {n: p.grad.detach().clone() for n, p in ref.named_parameters() if p.grad is not None}

    # ---- the data-parallel step
    model = build(M)
    dp.broadcast_parameters(model)
    reducer = dp.GradReducer(model.parameters(), bucket_bytes=64 << 10, flatten_params=True)
    opt = dp.FusedAdam(reducer, lr=3e-3)
    step = TrainStep(model, reducer, opt, tokens[shard].contiguous(), None, 5.0, 2.0, graph=True, warmup=3,
                     eps=eps[shard].contiguous())
    res["graph"] = step.graph_a is not None
    res["comm_in_graph"] = bool(step.comm_in_graph)
    # gradient of the first step: graph A (+ the all-reduce when it is not captured), before the update
    step._push_weights()
    if step.graph_a is not None:
        step.graph_a.replay()
        if world > 1 and not step.comm_in_graph:
            reducer.finish()
    else:
        step._fwd_bwd()
        reducer.finish()
    torch.cuda.synchronize()
    worst = 0.0
    # tensor-level max-norm error; the scale of a tensor is floored at 1e-3 of the largest gradient of the model: the
    # attention key biases have a mathematically ZERO gradient (softmax is shift invariant), what they hold is rounding
    # noise (~1e-9 of the largest gradient) that changes with the summation order of anything upstream
    gmax = max(float(g.abs().max()) for g in gref.values())
    for n, p in model.named_parameters():
        if n in gref:
            g = p.grad / world                      # fold_average: the buckets hold the rank SUM
            den = max(float(gref[n].abs().max()), 1e-3 * gmax)
            worst = max(worst, float((g - gref[n]).abs().max()) / den)
    res["grad_rel_err_vs_full_batch"] = worst
    reducer.zero_grad()
    losses = [float(step()["batch-loss"].detach()) for _ in range(3)]
    res["losses"] = losses
    flat = torch.cat([fp.detach().flatten() for fp in reducer.flat_params])
    if world > 1:
        lo, hi = flat.clone(), flat.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        res["params_identical"] = bool(torch.equal(lo, hi))
    else:
        res["params_identical"] = True
    res["finite"] = bool(torch.isfinite(flat).all())
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
