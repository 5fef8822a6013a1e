"""Per-kernel means of every counter in one or more rocprofv3 --pmc passes:

    python tools/pmc_counters.py <counter_collection.csv> [...]

One row per kernel (ours only), one column per counter: the counter's sum over a launch, averaged over launches.
rocprofv3 reports SQ counters summed over all XCDs/SEs; GRBM_GUI_ACTIVE is the sum of the 8 XCDs' active cycles."""
import csv
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_summary import short  # noqa: E402


def main():
    agg = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(lambda: defaultdict(set))
    for path in sys.argv[1:]:
        with open(path) as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"]
                if "yolo" not in k:
                    continue
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[k][r["Counter_Name"]].add((path, r["Dispatch_Id"]))
    counters = sorted({c for k in agg for c in agg[k]})
    for k in sorted(agg, key=lambda k: -agg[k].get("GRBM_GUI_ACTIVE", 0.0)):
        print(short(k))
        for c in counters:
            if c in agg[k]:
                n = len(launches[k][c])
                print("    %-34s %16.0f  (%d launches)" % (c, agg[k][c] / n, n))


if __name__ == "__main__":
    main()
