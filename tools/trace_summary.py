"""Summarise a rocprofv3 --kernel-trace CSV in dispatch order: consecutive runs of the same kernel sequence are
folded, so a micro-benchmark that repeats (kernel A, kernel B) x R per shape prints one line per kernel and shape."""
import csv
import sys


def main(path, reps=12):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    seq = [(r["Kernel_Name"].split("(")[0][-60:], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows
           if "wgrad" in r["Kernel_Name"] or "conv_igemm" in r["Kernel_Name"]]
    i = 0
    while i < len(seq):
        # a group = `reps` repetitions of a short pattern starting here
        for plen in (1, 2, 3):
            pat = [n for n, _ in seq[i:i + plen]]
            if len(seq) >= i + plen * reps and all(seq[i + j][0] == pat[j % plen] for j in range(plen * reps)):
                for j in range(plen):
                    d = sorted(seq[i + j + plen * r][1] for r in range(2, reps))
                    print(f"{pat[j]:>62s}  median {d[len(d)//2]/1e3:8.1f} us")
                print()
                i += plen * reps
                break
        else:
            i += 1


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12)
