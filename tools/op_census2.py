"""Which Python lines issue the small aten ops of one eager C2 training step (TorchDispatchMode + Python stacks).
python tools/op_census2.py [config]"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import nonauto_lm_amd as M
from nonauto_lm_amd import dp
from nonauto_lm_amd.step import TrainStep

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
B, S, D, H, K, dtype_name, ragged = bench.CONFIGS[cfg]
dev = torch.device("cuda", 0)
torch.manual_seed(13)
model = M.VAELmModel.from_params(vocab=M.SimpleVocab(bench.VOCAB), **bench.model_config(D, H, K)).to(dev).train()
model.training_outputs_only = True
reducer = dp.GradReducer(model.parameters(), flatten_params=True)
opt = dp.FusedAdam(reducer, lr=1e-4, bf16_shadow=True, model=model)
tokens = bench.synthetic_tokens(B, S, ragged, 13).to(dev)
step = TrainStep(model, reducer, opt, tokens, torch.bfloat16, 5.0, 2.0, graph=False)
for _ in range(4):
    step()
torch.cuda.synchronize()

SKIP = ("view", "reshape", "transpose", "t.default", "expand", "unsqueeze", "squeeze", "detach", "alias", "slice", "select",
        "as_strided", "permute", "_unsafe_view", "split", "unbind", "empty", "size", "stride", "is_", "sym_", "lift_fresh",
        "_local_scalar", "record_stream", "set_")
rows = collections.Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in SKIP):
            frame = "(autograd engine / torch internals)"
            for fr in reversed(traceback.extract_stack()):
                if "nonauto-lm_amd" in fr.filename or fr.filename.endswith("bench.py"):
                    frame = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line.strip()[:70]}"
                    break
            t = next((a for a in args if isinstance(a, torch.Tensor)), None)
            desc = "" if t is None else f"{tuple(t.shape)} {str(t.dtype).replace('torch.', '')}"
            rows[(name.replace("aten.", ""), desc, frame)] += 1
        return func(*args, **(kwargs or {}))


with Census():
    step()
torch.cuda.synchronize()
print("aten ops in one step (views excluded):", sum(rows.values()))
for (name, desc, frame), n in sorted(rows.items(), key=lambda kv: (kv[0][2], kv[0][0])):
    print(f"{n:3d} {name:34s} {desc:34s} {frame}")
