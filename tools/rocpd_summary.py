"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel table: calls, total / avg / min / max
duration and share of GPU kernel time.  Usage: python tools/rocpd_summary.py results.db > profiles/xxx.md"""
import sqlite3
import sys

db = sys.argv[1]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in con.execute("pragma table_info(%s)" % disp)]
scol = [r[1] for r in con.execute("pragma table_info(%s)" % sym)]
name_col = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else "name")
q = ("select s.%s, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) from %s d "
     "join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, disp, sym, name_col))
rows = list(con.execute(q))
tot = sum(r[2] for r in rows)
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|")
for name, n, t, mn, mx in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    nm = name if len(name) < 110 else name[:107] + "..."
    print("| `%s` | %d | %.2f | %.1f | %.1f | %.1f | %.2f |" % (nm, n, t / 1e6, t / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
print("\ntotal GPU kernel time %.1f ms over %d kernels (%d distinct)" % (tot / 1e6, sum(r[1] for r in rows), len(rows)))
