"""Program run under `rocprofv3 --pmc ...` (one counter group per run): launches each dominant hand-written
kernel of the bench shape a few times through the C ABI, eagerly, with the bench's model/tokens.
Usage: python3 tools/pmc_target.py [config] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nonauto_lm_amd as M  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B, S, D, H, K, dtype_name, ragged = bench.CONFIGS[cfg]
dev = torch.device("cuda", 0)
torch.manual_seed(13)
model = M.VAELmModel.from_params(vocab=M.SimpleVocab(bench.VOCAB), **bench.config_model_params(cfg)).to(dev).train()
tokens = bench.synthetic_tokens(B, S, ragged, 13).to(dev)
auto = bench.FAMILY.get(cfg) == "auto"
bench.flow_kernel_times(model, tokens, torch.bfloat16 if dtype_name == "bf16" else torch.float32, reps=reps, graph=False,
                        family="auto" if auto else "nonauto", loss_kernels=not auto)
print("done", cfg, reps)
