"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short committed summary.

    python tools/summarize_rocprof.py gpurun_out/prof_r1/bench_kernel_stats.csv profiles/r01_bench_kernel_stats.md "command line"
"""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
cmd = sys.argv[3] if len(sys.argv) > 3 else ""
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(dst, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\n")
    f.write("| kernel | calls | total ms | avg ms | % | min ms | max ms |\n|---|---|---|---|---|---|---|\n")
    for r in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 12]:
        name = r["Name"].split("(")[0]
        if len(name) > 70:
            name = name[:67] + "..."
        f.write(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e6:.4f} | "
                f"{float(r['Percentage']):.3f} | {float(r['MinNs'])/1e6:.4f} | {float(r['MaxNs'])/1e6:.4f} |\n")
print(open(dst).read())
