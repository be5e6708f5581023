"""Per-stream view of one training step from a rocprofv3 kernel trace (…_kernel_trace.csv of `--kernel-trace --stats`).
usage: python profiles/critical_path.py gpurun_out/prof_r1n/runn_kernel_trace.csv > profiles/r01n_critical_path.txt"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"]) for r in rows)
adam = [i for i, e in enumerate(ev) if "adam" in e[2]]          # the optimiser kernel ends a step
# the step of MEDIAN length among the last six (all replayed from the launch tape in a `--steps 8` run): a single step can carry a
# host hiccup of the profiled run
cands = [ev[adam[i - 1] + 1:adam[i] + 1] for i in range(max(1, len(adam) - 6), len(adam))]
cands.sort(key=lambda g: max(e[1] for e in g) - g[0][0])
seg = cands[len(cands) // 2]
t0 = seg[0][0]
print("Per-stream view of ONE training step, the median-length one of the run's last six (times in us from the step's first kernel; the profiler's tracing slows the host,")
print("so host-side gaps are larger than in an unprofiled step).  Made by profiles/critical_path.py from the kernel trace.\n")
streams = collections.defaultdict(list)
for s, e, n, st in seg:
    streams[st].append((s, e, n))
names = {"0": "main", "1": "IIC branch", "2": "wgrad"}
for k, v in sorted(streams.items()):
    print(f"stream {k} ({names.get(k, '?')}): {len(v)} launches, busy {sum(e - s for s, e, _ in v) / 1e3:.0f} us, "
          f"active {(v[0][0] - t0) / 1e3:.0f} .. {(max(e for _, e, _ in v) - t0) / 1e3:.0f} us")
busy, cs, ce = 0, None, None
for s, e, _, _ in seg:
    if ce is None or s > ce:
        busy += (ce - cs) if ce is not None else 0
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"union busy {busy / 1e3:.0f} us of a {(max(e[1] for e in seg) - t0) / 1e3:.0f} us step\n")


def short(n):
    return n.split("(")[0].replace("void ", "")[:100]


print("IIC branch stream, start / duration / kernel:")
for s, e, n in streams["1"]:
    print(f"  {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {short(n)}")
a = [s for s, e, n in streams["1"] if "head_local_fwd_mfma_kernel<16>" in n][0]
b = [e for s, e, n in streams["1"] if "head_local_bwd_wave" in n][0]
print("\nmain stream while the top tap's chain runs (head_local_fwd_mfma_kernel<16> .. end of head_local_bwd_wave_kernel<16>):")
for s, e, n in streams["0"]:
    if e >= a and s <= b:
        print(f"  {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {short(n)}")
