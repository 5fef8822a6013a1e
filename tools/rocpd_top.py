"""Per-kernel statistics of a rocprofv3 --kernel-trace run (its rocpd SQLite output) as text.
usage: python tools/rocpd_top.py RESULTS.db [rows=40]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    cur = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit ?", (rows,))
    print("kernel,calls,total_us,avg_us,percent")
    for name, calls, total, avg, pct in cur:
        print(f"\"{name[:150]}\",{calls},{total:.1f},{avg:.2f},{pct:.2f}")


if __name__ == "__main__":
    main()
