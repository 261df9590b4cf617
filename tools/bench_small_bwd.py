"""Time the fused NICE stack entry points at the C2 / C4 latent shapes through the C ABI: HIP-graph replay of 10
back-to-back launches between two events (bench._time_calls).  MIVAE_NICE_BWD_PAR=0 selects the task-serial backward
kernel of nice_stack.hip instead of k_stack_bwd_par (nice_small.hip) for an A/B.
Usage: python3 tools/bench_small_bwd.py [config]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nonauto_lm_amd as M  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
B, S, D, H, K, dtype_name, ragged = bench.CONFIGS[cfg]
dev = torch.device("cuda", 0)
torch.manual_seed(13)
model = M.VAELmModel.from_params(vocab=M.SimpleVocab(bench.VOCAB), **bench.config_model_params(cfg)).to(dev).train()
tokens = bench.synthetic_tokens(B, S, ragged, 13).to(dev)
auto = bench.FAMILY.get(cfg) == "auto"
t = bench.flow_kernel_times(model, tokens, torch.bfloat16, reps=50, graph=True, family="auto" if auto else "nonauto",
                            loss_kernels=False)
print(json.dumps({"config": cfg, "bwd_par": os.environ.get("MIVAE_NICE_BWD_PAR", "1"),
                  **{k: round(v["avg_us"], 2) for k, v in t.items() if k.startswith("nice")}}))
