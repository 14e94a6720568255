"""Registers, scratch, LDS and occupancy of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py gpras_amd/csrc/sf_pass2.hip -DSF_KID=0 [name filter]
"""
import re
import subprocess
import sys

args = [a for a in sys.argv[1:] if a.endswith(".hip") or a.startswith("-")]
filt = [a for a in sys.argv[1:] if a not in args]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + args
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: (?:[^:]+:\d+:\d+: )?\s*(Function Name|SGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        m2 = re.search(r"Function Name: (\S+)", line)
        if m2:
            cur = {"name": m2.group(1)}
            rows.append(cur)
        else:
            for key in ("SGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
                m3 = re.search(re.escape(key) + r": (\d+)", line)
                if m3 and cur is not None:
                    cur[key] = int(m3.group(1))
        continue
    if m.group(1) == "Function Name":
        cur = {"name": m.group(2)}
        rows.append(cur)
    elif cur is not None:
        cur[m.group(1)] = int(m.group(2))
demangle = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>8} {'LDS':>7} {'occ':>4}  kernel")
for r, name in zip(rows, demangle):
    if filt and not any(f in name for f in filt):
        continue
    print(f"{r.get('VGPRs', -1):>5} {r.get('AGPRs', -1):>5} {r.get('SGPRs', -1):>5} {r.get('ScratchSize [bytes/lane]', -1):>8} {r.get('LDS Size [bytes/block]', -1):>7} {r.get('Occupancy [waves/SIMD]', -1):>4}  {name[:150]}")
