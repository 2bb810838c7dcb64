"""scratch: per-dispatch durations of the bfs_* kernels of the LAST BFS run in a rocprofv3 kernel-trace db, in launch order"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
order = "start" if "start" in cols else ("start_timestamp" if "start_timestamp" in cols else "rowid")
rows = list(c.execute(f"select name, grid_x, duration from kernels where name like '%bfs_%' or name like '%scan_%' order by {order}"))
# the last run: from the last bfs_discover_kernel whose predecessor chain restarts (7 levels per run)
idx = [k for k, r in enumerate(rows) if r[0].startswith("bfs_discover_kernel")]
start = idx[-7] if len(idx) >= 7 else 0
for name, g, d in rows[start:]:
    print(f"{name.split('(')[0]:28s} grid_x={g:>9} {d / 1000:9.1f} us")
