"""What runs after the bulge-chasing kernel of the last step (kernel trace CSV): batched bisection on the main stream,
the consumed eigenvector's path on the side stream."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split('(')[0][-30:], r["Queue_Id"]))
rows.sort()
sb = [i for i, r in enumerate(rows) if 'sb2st' in r[2]][-1]
t0 = rows[sb][1]
for s, e, n, q in rows[sb:sb + 14]:
    print("%-32s q%s start %+8.2f ms end %+8.2f ms  dur %7.2f ms" % (n, q, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
