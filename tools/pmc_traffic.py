"""Turn the rocprofv3 PMC passes of tools/pmc_target.py into profiles/pmc_traffic.json.

    python tools/pmc_traffic.py <config> <fetch_dir> <write_dir> [out.json]

Each directory holds one `--pmc FETCH_SIZE` or `--pmc WRITE_SIZE` run (--output-format csv).  Per kernel the
counter is averaged over its dispatches; units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md
(HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced
reads at 64 bytes, so the read side is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

import re

# device kernel (regex on the demangled name) -> logical entry point of bench.py's kernel table
NAMES = {r"k_nll_fwd<": "nll_fwd", r"k_nll_bwd<": "nll_bwd", r"k_stack_fwd<": "nice_stack_fwd",
         r"k_stack_bwd<": "nice_stack_bwd", r"k_stack_bwd_par<": "nice_stack_bwd", r"k_dw_partial[<(]": "nice_dw", r"k_dw_reduce[<(]": "nice_dw",
         r"k_lce_fwd(_s)?<": "lce_fwd", r"k_lce_combine\(": "lce_fwd", r"k_lce_bwd(_s)?<\d+, false[,>]": "lce_bwd",
         r"k_lce_bwd(_s)?<\d+, true[,>]": "lce_bwd_dh", r"k_lce_dh_combine\(": "lce_bwd_dh", r"k_lce_dh<": "lce_bwd_dh", r"k_lce_dw<": "lce_dw", r"k_lce_dwp<": "lce_dw",
         r"k_reparam_fwd<": "reparam_fwd", r"k_reparam_bwd<": "reparam_bwd",
         r"k_post_fwd<": "reparam_fwd", r"k_post_bwd<": "reparam_bwd",      # sigma head + sample + log q in one launch
         # wide-latent path (csrc/nice_big.hip): MODE 0 = forward, 2 = backward chain
         r"k_big_chain<\d+, 0, ": "nice_stack_fwd", r"k_big_chain<\d+, 2, ": "nice_stack_bwd",
         r"k_big_nets<": "nice_stack_bwd", r"k_big_dw<": "nice_stack_bwd", r"k_big_dw2<": "nice_stack_bwd", r"k_big_dw_reduce\(": "nice_stack_bwd"}


def collect(directory, counter):
    per = defaultdict(lambda: defaultdict(list))          # logical kernel -> device kernel -> values
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] != counter:
                continue
            for key, name in NAMES.items():
                if re.search(key, row["Kernel_Name"]):
                    per[name][key].append(float(row["Counter_Value"]))
    # one logical launch = one dispatch of each of its device kernels
    return {name: sum(sum(v) / len(v) for v in ks.values()) for name, ks in per.items()}


def main():
    config, fdir, wdir = sys.argv[1:4]
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             "profiles", "pmc_traffic.json")
    fetch, write = collect(fdir, "FETCH_SIZE"), collect(wdir, "WRITE_SIZE")
    table = json.load(open(out)) if os.path.exists(out) else {}
    table.setdefault("_method", "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of tools/pmc_target.py; "
                                "KiB -> bytes; FETCH_SIZE x2 (gfx950 128-byte requests tallied at 64 B); per launch")
    table[config] = {}
    for name in sorted(set(fetch) | set(write)):
        rd, wr = 2.0 * 1024.0 * fetch.get(name, 0.0), 1024.0 * write.get(name, 0.0)
        table[config][name] = {"read_bytes": round(rd), "write_bytes": round(wr), "hbm_bytes_per_launch": round(rd + wr)}
    json.dump(table, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(table[config], indent=1))


if __name__ == "__main__":
    main()
