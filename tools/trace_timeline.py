"""Timeline analysis of a rocprofv3 --kernel-trace CSV: inside the sy2sb window of the last step, how long is the GPU
running (a) two or more big GEMMs, (b) exactly one, (c) only latency-bound kernels (panel QR, T, small products), (d) nothing.
usage: python tools/trace_timeline.py <kernel_trace.csv>"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# last sb2st launch marks the end of the last step's sy2sb; the std_form before it marks its start
sb = [i for i, r in enumerate(rows) if "sb2st_kernel" in r[2]]
end = rows[sb[-1]][0]
stds = [i for i, r in enumerate(rows) if "std_form_kernel" in r[2] and r[1] <= end]
start = rows[stds[-1]][1]
win = [r for r in rows if r[0] >= start and r[1] <= end + 1]
def kind(n):
    if "gemm2_kernel" in n: return "big"
    if "gemm_kernel<64, 128" in n: return "big"
    return "small"
ev = []
for s, e, n in win:
    k = kind(n)
    ev.append((s, 1, k)); ev.append((e, -1, k))
ev.sort()
cnt = collections.Counter(); t0 = start; acc = collections.Counter()
for t, d, k in ev:
    state = ("big>=2" if cnt["big"] >= 2 else "big=1" if cnt["big"] == 1 else "small only" if cnt["small"] > 0 else "idle")
    acc[state] += t - t0; t0 = t
    cnt[k] += d
tot = end - start
print("sy2sb window %.1f ms" % (tot / 1e6))
for k, v in acc.most_common():
    print("  %-10s %7.1f ms  %5.1f %%" % (k, v / 1e6, 100.0 * v / tot))
per = collections.Counter()
for s, e, n in win: per[n.replace("(anonymous namespace)::", "").split("(")[0][:60]] += e - s
for k, v in per.most_common(8): print("  sum %-60s %7.1f ms" % (k, v / 1e6))
