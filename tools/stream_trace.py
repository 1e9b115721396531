"""One training step cut out of a rocprofv3 kernel trace, per HIP stream: who runs when, alone or beside whom, and where the caller's stream waits.
usage: rocprofv3 --kernel-trace -d DIR -o NAME --output-format csv -- python3 tools/prof_config.py 32 18 1 128 416 f32 10
       python3 tools/stream_trace.py DIR/.../NAME_kernel_trace.csv [step_index_from_end=2]
A step is the span between two consecutive adam_kernel launches.  Streams are told apart by the trace's Stream_Id column (Queue_Id when it has none)."""
import csv, sys, collections

path = sys.argv[1]; back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = list(csv.DictReader(open(path)))
key = "Stream_Id" if "Stream_Id" in rows[0] and len({r["Stream_Id"] for r in rows}) > 1 else "Queue_Id"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[key], r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[3]]
if len(adam) < back + 1:
    sys.exit("not enough steps in the trace")
lo, hi = adam[-back - 1], adam[-back]
step = ev[lo + 1:hi + 1]                       # kernels after one ADAM up to and including the next
t0 = ev[lo][1]; t1 = step[-1][0]               # end of the previous ADAM .. start of this step's ADAM
print("step: %.3f ms between two adam_kernel launches, %d kernels, streams told apart by %s" % ((t1 - t0) / 1e6, len(step) - 1, key))
by = collections.defaultdict(list)
for s, e, q, n in step[:-1]:
    by[q].append((s, e, n))
order = sorted(by, key=lambda q: -len(by[q]))
main = order[0]
for q in order:
    k = by[q]; busy = sum(e - s for s, e, _ in k)
    print("  stream %-6s %4d kernels, busy %.3f ms, first %.3f, last end %.3f ms%s" % (q, len(k), busy / 1e6, (k[0][0] - t0) / 1e6, (max(e for _, e, _ in k) - t0) / 1e6, "   <- caller's stream (most kernels)" if q == main else ""))
# sweep: time with 0 / 1 / 2+ streams active, and per stream the time it runs ALONE
pts = []
for q in by:
    for s, e, _ in by[q]:
        pts.append((s, 1, q)); pts.append((e, -1, q))
pts.sort()
active = collections.Counter(); last = t0; none = 0; alone = collections.Counter(); multi = 0
for t, d, q in pts:
    live = [x for x in active if active[x] > 0]
    dt = t - last
    if dt > 0:
        if not live: none += dt
        elif len(live) == 1: alone[live[0]] += dt
        else: multi += dt
    active[q] += d; last = t
none += max(0, t1 - last)
print("  no kernel running: %.3f ms; two or more streams active: %.3f ms; alone: %s" % (none / 1e6, multi / 1e6, ", ".join("%s %.3f ms" % (q, alone[q] / 1e6) for q in order)))
# the caller's stream: its gaps, and what the other streams did during the large ones
m = sorted(by[main]); gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(m, m[1:]):
    if s1 - e0 > 0: gaps.append((s1 - e0, e0, s1, n0, n1))
tail_gap = t1 - max(e for _, e, _ in m)
print("  gaps on the caller's stream: %d, sum %.3f ms; after its last kernel until ADAM: %.1f us" % (len(gaps), sum(g[0] for g in gaps) / 1e6, tail_gap / 1e3))
for g, a, b, n0, n1 in sorted(gaps, reverse=True)[:6]:
    others = [(q, n) for q in by if q != main for s, e, n in by[q] if s < b and e > a]
    print("    %.1f us after %s before %s; meanwhile: %s" % (g / 1e3, n0.split("(")[0][:50], n1.split("(")[0][:50], ", ".join("%s:%s" % (q, n.split("(")[0][5:40]) for q, n in others[:4]) or "nothing"))
lastm = max(e for _, e, _ in m)
late = [(q, s, e, n) for q in by if q != main for s, e, n in by[q] if e > lastm]
print("  kernels of other streams still running after the caller's last kernel: %s" % (", ".join("%s %s (%.0f us, ends +%.0f us)" % (q, n.split("(")[0][5:45], (e - s) / 1e3, (e - lastm) / 1e3) for q, s, e, n in late) or "none"))
