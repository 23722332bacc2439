"""Per-kernel summary (calls, average / total duration, share of GPU time) of a rocprofv3 run stored as a rocpd SQLite database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- python3 ...` writes DIR/NAME_results.db on ROCm 7), as JSON for profiles/.
    python scripts/rocpd_summary.py gpurun_out/prof_x/x_results.db profiles/r2_x.json "command line that was profiled" """
import collections
import json
import sqlite3
import sys

db, out, cmd = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
con = sqlite3.connect(db)
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
rows = cur.execute(f"select d.start, d.end, s.kernel_name from {disp} d join {sym} s on d.kernel_id = s.id").fetchall()
agg = collections.defaultdict(list)
for st, en, name in rows:
    agg[name].append((en - st) / 1e3)
total = sum(sum(v) for v in agg.values())
kern = [{"name": k[:160], "calls": len(v), "avg_us": sum(v) / len(v), "total_ms": sum(v) / 1e3, "share": sum(v) / total}
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:16]]
json.dump({"command": cmd, "dispatches": len(rows), "gpu_time_ms": total / 1e3, "kernels": kern}, open(out, "w"), indent=1)
for k in kern[:10]:
    print(f"{k['total_ms']:9.2f} ms  n={k['calls']:6d}  avg {k['avg_us']:8.2f} us  {k['share'] * 100:5.1f} %  {k['name'][:90]}")
