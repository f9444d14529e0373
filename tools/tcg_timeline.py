"""the repeating launch sequence of a tCG loop from a rocprofv3 kernel trace: mean duration and preceding gap per
position of the period anchored on a kernel name: python tools/tcg_timeline.py <dir> <anchor substring>"""
import csv, glob, sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
anchor = sys.argv[2]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("void ", "").replace("dcora::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:34]


names = [short(r["Kernel_Name"]) for r in rows]
st = [int(r["Start_Timestamp"]) for r in rows]
en = [int(r["End_Timestamp"]) for r in rows]
anchors = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
periods = defaultdict(list)
for a, b in zip(anchors[:-1], anchors[1:]):
    periods[tuple(names[a:b])].append(a)
sig, starts = max(periods.items(), key=lambda kv: len(kv[1]))
print("%d periods of %d launches (of %d anchored)" % (len(starts), len(sig), len(anchors) - 1))
L = len(sig)
dur = [0.0] * L
gap = [0.0] * L
for a in starts:
    for j in range(L):
        dur[j] += (en[a + j] - st[a + j]) / 1e3
        gap[j] += (st[a + j] - en[a + j - 1]) / 1e3 if a + j > 0 else 0.0
n = len(starts)
for j in range(L):
    print("%-36s dur %6.2f us  gap before %5.2f us" % (sig[j], dur[j] / n, gap[j] / n))
print("period: %.2f us (busy %.2f)" % (sum((st[a + L] - st[a]) for a in starts) / 1e3 / n, sum(dur) / n))
