"""GPU idle time between consecutive kernels of a rocprofv3 --kernel-trace csv, attributed to the kernel that FOLLOWS the gap
(tools/kernel_gaps.py <kernel_trace.csv> [skip_first_n]): where a loop that synchronises every step leaves the GPU waiting."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[skip:]
gap, busy, cnt = collections.Counter(), collections.Counter(), collections.Counter()
prev_end = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-60:]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None:
        gap[name] += max(0, s - prev_end)
    busy[name] += e - s
    cnt[name] += 1
    prev_end = max(prev_end or 0, e)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"{len(rows)} kernels over {span / 1e3:.1f} us; busy {sum(busy.values()) / 1e3:.1f} us, idle {sum(gap.values()) / 1e3:.1f} us")
print(f"{'kernel':62s} {'calls':>6s} {'avg us':>8s} {'avg gap before, us':>20s}")
for name, _ in sorted(busy.items(), key=lambda kv: -kv[1]):
    print(f"{name:62s} {cnt[name]:6d} {busy[name] / cnt[name] / 1e3:8.2f} {gap[name] / cnt[name] / 1e3:20.2f}")
