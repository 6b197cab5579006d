"""Per-panel timeline of sy2sb from a rocprofv3 --kernel-trace CSV (last step): for a few panels, when each kernel of the chain
QR -> G -> T -> W -> symm -> K -> Z -> update starts and ends relative to the panel's QR, and the gaps between them; plus the
per-kernel sums and the stage windows of the whole step.  usage: python tools/panel_timeline.py <kernel_trace.csv> [panel ...]"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("bsp::", "").replace("void ", "").split("(")[0]))
rows.sort()
bands = [i for i, r in enumerate(rows) if r[2].startswith("band_kernel")]
start_i = bands[-1]
step = rows[start_i:]
t0 = step[0][0]
def short(n):
    if n.startswith("gemm2_kernel<128"): return "update"
    if n.startswith("gemm2_kernel<64"): return "symm"
    if n.startswith("gemm_kernel"): return "gemm64"
    return n.split("<")[0][:22]
print("last step: %d launches, %.2f ms" % (len(step), (max(r[1] for r in step) - t0) / 1e6))
per = collections.defaultdict(lambda: [0, 0])
for s, e, n in step:
    per[short(n)][0] += 1; per[short(n)][1] += e - s
for k, v in sorted(per.items(), key=lambda x: -x[1][1]):
    print("  %-24s calls %5d  sum %8.2f ms  avg %8.1f us" % (k, v[0], v[1] / 1e6, v[1] / v[0] / 1e3))
# busy time (union of all kernel intervals) and idle gaps inside the step
ev = sorted([(s, 1) for s, e, n in step] + [(e, -1) for s, e, n in step])
c = 0; last = t0; idle = 0
for t, d in ev:
    if c == 0: idle += t - last
    c += d; last = t
print("  GPU idle (no kernel running) inside the step: %.2f ms" % (idle / 1e6))
qr = [i for i, r in enumerate(step) if r[2].startswith("panel_qr") or r[2].startswith("tsqr_tree")]
print("panel factorisation kernel, duration by panel (us):", " ".join("%.0f" % ((step[i][1] - step[i][0]) / 1e3) for i in qr))
want = [int(x) for x in sys.argv[2:]] or [0, 1, 8, 24, 40, 56, 62]
print("panels in trace: %d" % len(qr))
for p in want:
    if p + 1 >= len(qr): continue
    a, b = qr[p], qr[p + 1]
    base = step[a][0]
    print("panel %d (next QR starts %.1f us after this one):" % (p, (step[b][0] - base) / 1e3))
    prev_end = None
    for s, e, n in step[a:b]:
        gap = "" if prev_end is None else " gap %+7.1f" % ((s - prev_end) / 1e3)
        print("    %-10s start %8.1f  dur %8.1f%s" % (short(n), (s - base) / 1e3, (e - s) / 1e3, gap))
        prev_end = max(prev_end or e, e)
