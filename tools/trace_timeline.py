"""Per-queue busy time and the chain's composition from a rocprofv3 kernel trace of ONE lone factorisation (development aid).
    python tools/trace_timeline.py gpurun_out/prof_<tag>/b_kernel_trace.csv [fit_index]
Splits the trace into fits at kmat_kernel launches, takes fit `fit_index` (default: last) and prints, per queue, the sum of kernel
durations, the idle gaps, and per kernel name (count, total, mean)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "kmat_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 1
lo, hi = starts[k], (starts[k + 1] if k + 1 < len(starts) else len(rows))
fit = rows[lo:hi]
t0 = int(fit[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in fit)
print(f"fit {k}: {len(fit)} launches, span {(t1 - t0) / 1e6:.3f} ms")
def short(n):
    n = n.replace("void gprx::", "").replace("gprx::", "")
    return n.split("(")[0][:60]
byq = collections.defaultdict(list)
for r in fit: byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    first, last = int(rs[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rs)
    print(f"queue {q}: {len(rs)} launches, busy {busy / 1e6:.3f} ms, first->last {(last - first) / 1e6:.3f} ms")
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rs:
        key = short(r["Kernel_Name"]); agg[key][0] += 1; agg[key][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for key, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"    {key:62s} {n:5d} x {tot / n / 1e3:9.1f} us = {tot / 1e6:8.3f} ms")
# time during which queue A (main) has nothing running
def intervals(rs): return sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rs)
for q, rs in byq.items():
    iv = intervals(rs); gaps = 0; cur_end = iv[0][1]
    for s, e in iv[1:]:
        if s > cur_end: gaps += s - cur_end
        cur_end = max(cur_end, e)
    print(f"queue {q}: idle between its own launches {gaps / 1e6:.3f} ms")
