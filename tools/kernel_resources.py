"""Print one line per kernel from hipcc -Rpass-analysis=kernel-resource-usage (VGPR/AGPR/spill/occupancy)."""
import re
import subprocess
import sys

src = sys.argv[1]
extra = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: (.*?): (.*) \[-Rpass", line)
    if not m:
        m2 = re.search(r"remark: (Function Name): (\S+)", line)
        if not m2:
            continue
        m = m2
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["name"])
    name = re.sub(r"\(.*\)$", "", name)[:90]
    print(f"{name:90s} vgpr={r.get('VGPRs')} agpr={r.get('AGPRs')} sgpr={r.get('SGPRs')} vspill={r.get('VGPRs Spill')} "
          f"scratch={r.get('ScratchSize [bytes/lane]')} occ={r.get('Occupancy [waves/SIMD]')} lds={r.get('LDS Size [bytes/block]')}")
