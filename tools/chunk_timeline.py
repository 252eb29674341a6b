"""One chunk of the mk_search pipeline as a kernel timeline (both HIP streams), from a rocprofv3 --kernel-trace csv.
   python tools/chunk_timeline.py <kernel_trace.csv> [chunk number within the last step]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("mk::", ""), r.get("Queue_Id", "")) for r in rows)
starts = [e for e in ev if e[2].startswith("kmer_count_kernel")]
per_step = len(starts) // 2 if len(starts) >= 2 else len(starts)
a = starts[len(starts) - per_step + which]
b = starts[len(starts) - per_step + which + 1]
t0, t1 = a[0], b[0]
print("chunk window %.2f ms" % ((t1 - t0) / 1e6))
tiny = {}
for s, e, n, q in ev:
    if e < t0 or s > t1: continue
    d = (e - s) / 1e6
    short = ("rocprim" if "rocprim" in n else n[:44])
    if d < 0.15:
        k = (q, short); tiny.setdefault(k, [0, 0.0, (s - t0) / 1e6]); tiny[k][0] += 1; tiny[k][1] += d
        continue
    print("q%-3s %8.2f +%7.2f  %s" % (q, (s - t0) / 1e6, d, short))
for (q, n), v in sorted(tiny.items(), key=lambda x: x[1][2]): print("q%-3s tiny x%-4d sum %6.2f ms first at %7.2f  %s" % (q, v[0], v[1], v[2], n))
