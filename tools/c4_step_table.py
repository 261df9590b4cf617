"""profiles/r03/c4_step_table.md from the rocprofv3 kernel stats of the eager C4 step (tools/collect_r03.sh):
GPU time per training step by kernel group.  Usage: python tools/c4_step_table.py <kernel_stats.csv> <bench log with the JSON line> <out.md>"""
import csv
import json
import re
import sys

stats, log, outp = sys.argv[1:4]
rows = list(csv.DictReader(open(stats)))
line = json.loads([ln for ln in open(log) if ln.startswith("{")][0])
steps = line["steps"] + line["warmup"] + 2      # timed + warm-up (+ TrainStep's own two warm-up passes)
groups = [
    ("latent path, in scope (posterior sample + log q, NICE stack fwd/bwd + weight gradients, KL/ELBO tail)",
     r"k_post_|k_stack_|k_dw_|k_tail_|k_nice_pack|k_segment_sum|k_gauss|k_reparam|k_sigma_head|k_kl_|k_elbo_"),
    ("reconstruction loss + embedding gradient (libmivae)", r"k_lce_|k_nll_|k_embed_bwd|k_fold_aug|k_colsum|k_col_"),
    ("clip + Adam + bookkeeping (libmivae)",
     r"k_adam|k_sumsq|k_clip|k_accumulate|k_fused_adam|k_scalar|k_tokens_mask|k_gather_last|k_rng|k_hbm"),
    ("encoder LSTM (MIOpen) + decoder loop + projections: stock PyTorch-ROCm (Tensile GEMMs, aten elementwise / reductions)", r".*"),
]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
acc = [[0.0, 0] for _ in groups]
detail = [[] for _ in groups]
for r in rows:
    for i, (_, pat) in enumerate(groups):
        if re.search(pat, r["Name"]):
            acc[i][0] += float(r["TotalDurationNs"])
            acc[i][1] += int(r["Calls"])
            detail[i].append(r)
            break
out = ["# C4 (AR VAE: LSTM encoder, LSTM-cell decoder, K = 4 NICE on a (256, 64) latent) -- GPU time per training step by kernel group",
       "",
       f"`rocprofv3 --kernel-trace --stats -- python3 bench.py --config c4 --no-others --no-cpu-baseline --steps {line['steps']} "
       f"--warmup {line['warmup']}` (eager step; {steps} steps traced incl. warm-up; under the profiler the step takes "
       f"{line['ms_per_step']} ms wall -- launch-bound, the tracer adds ~2x -- against 28-30 ms untraced; GPU-busy time "
       f"{tot / 1e6 / steps:.2f} ms/step).  Source: `c4_step_kernel_stats.csv`; made by `tools/c4_step_table.py`.", "",
       "| group | GPU ms / step | share of GPU time | launches / step |", "|---|---|---|---|"]
for (name, _), (t, n) in zip(groups, acc):
    out.append(f"| {name} | {t / 1e6 / steps:.3f} | {100 * t / tot:.1f} % | {n / steps:.0f} |")
out += ["", "In-scope kernels:", "", "| kernel | calls / step | avg us |", "|---|---|---|"]
for r in detail[0]:
    nm = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:90]
    out.append(f"| `{nm}` | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} |")
out += ["", "The flow / KL share of the C4 step is a few percent: the step is the 64-iteration teacher-forced LSTM-cell decoder loop "
        "and the packed-sequence LSTM encoder (out of scope, SURVEY 8f), ~1 900 small launches per step.  (The in-scope rows "
        "include the ~60 extra launches per kernel of bench.py's own kernel-timing section that follows the timed steps, so "
        "their share is an upper bound: per step the latent path is one `k_post_fwd`, one `k_stack_fwd`, one `k_tail_fwd` "
        "forward and `k_tail_bwd`, `k_stack_bwd`, `k_dw_partial`, `k_dw_reduce`, `k_post_bwd` backward, ~70 us.)"]
open(outp, "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
