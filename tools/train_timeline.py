"""Timeline of a rocprofv3 --kernel-trace run of the training bench (rocpd SQLite output): how much of the wall time
of the last steps the GPU is busy (union of the kernels' intervals), the idle gaps, and kernel time per category.
usage: python tools/train_timeline.py RESULTS.db [steps=4]"""
import sqlite3
import sys


def category(name: str) -> str:
    for key, cat in (("wgrad_reduce", "wgrad reduce"), ("wgrad", "wgrad"), ("conv_igemm", "conv fwd/dgrad"),
                     ("bn_reduce", "bn bwd reduce"), ("bn_apply", "bn apply"), ("bn_finalize", "bn finalize"),
                     ("bn_", "bn other"), ("loss_", "loss"), ("sgd_step", "sgd"), ("pack_train", "pack"),
                     ("stem_down", "frozen prefix"), ("resblock_c64", "frozen prefix"), ("at::native", "torch"),
                     ("rocclr", "copies/fills")):
        if key in name:
            return cat
    return "other"


def main():
    db = sqlite3.connect(sys.argv[1])
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    views = [r[0] for r in db.execute("select name from sqlite_master where type in ('view','table')")]
    if "kernels" not in views:
        print("no 'kernels' view; have:", views)
        return
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    # the steps are delimited by the optimizer's launch
    ends = [e for n, s, e in rows if "sgd_step" in n]
    if len(ends) < nsteps + 1:
        print("too few steps in the trace:", len(ends), cols)
        return
    t0, t1 = ends[-nsteps - 1], ends[-1]
    win = [(n, max(s, t0), min(e, t1)) for n, s, e in rows if e > t0 and s < t1]
    wall = (t1 - t0) / nsteps
    busy, cur_s, cur_e = 0, None, None
    gaps, where, last = [], [], None
    for n, s, e in win:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
                gaps.append(s - cur_e)
                where.append((s - cur_e, last, n))
            cur_s, cur_e = s, e
            last = n
        else:
            if e > cur_e:
                cur_e, last = e, n
    busy += cur_e - cur_s
    print(f"steps {nsteps}: wall {wall / 1e6:.3f} ms per step, GPU busy {busy / nsteps / 1e6:.3f} ms "
          f"({100 * busy / (t1 - t0):.1f} %), idle gaps {sum(gaps) / nsteps / 1e6:.3f} ms per step in "
          f"{len(gaps) / nsteps:.0f} gaps, largest {max(gaps) / 1e3:.1f} us")
    print("largest gaps (us, kernel before -> kernel after):")
    for g, a, b in sorted(where, reverse=True)[:12 * nsteps]:
        print(f"  {g / 1e3:8.1f}  {a[:60]} -> {b[:60]}")
    cat = {}
    for n, s, e in win:
        c = cat.setdefault(category(n), [0, 0])
        c[0] += e - s
        c[1] += 1
    print("category, ms per step (sum of kernel durations), launches per step")
    for k, (t, c) in sorted(cat.items(), key=lambda kv: -kv[1][0]):
        print(f"{k:18s} {t / nsteps / 1e6:8.3f} {c / nsteps:8.1f}")


if __name__ == "__main__":
    main()
