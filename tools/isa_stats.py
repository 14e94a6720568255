"""Instruction mix and resources of one kernel from `hipcc -S --cuda-device-only` output (development aid).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/gprx.s gpras_amd/csrc/gprx.hip
    python tools/isa_stats.py /tmp/gprx.s kmat_kernelILi0ELi0"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
name = lines[start].split(":")[0]
end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i] and lines[i].strip().endswith(":"))
body = [l.strip() for l in lines[start + 1:end] if l.strip() and not l.strip().startswith((".", ";", "_Z"))]
body = [l for l in body if not l.endswith(":")]
c = collections.Counter(l.split()[0] for l in body)
print(name, len(body), "instructions")
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
    print(f"  {k:32s} {v}")
for l in lines[end:end + 60]:
    if any(t in l for t in (".num_vgpr", ".num_agpr", ".private_seg_size", "ScratchSize", "LDSByteSize", "Occupancy", ".numbered_sgpr")):
        print(" ", l.strip())
