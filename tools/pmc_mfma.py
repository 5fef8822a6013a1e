"""MFMA utilisation per kernel from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE):

    python tools/pmc_mfma.py <counter_collection.csv>

utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 256 CUs * 4 SIMDs), cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3
reports the sum over the 8 XCDs).  MOPS_BF16 counts units of 512 FLOPs: FLOP/cycle/CU = 512 * MOPS / cycles / 256,
against 4096 dense bf16 FLOP/cycle/CU on the matrix cores."""
import csv
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_summary import short  # noqa: E402


def main():
    vals = defaultdict(lambda: defaultdict(float))
    name_of = {}
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            vals[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            name_of[r["Dispatch_Id"]] = r["Kernel_Name"]
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for d, v in vals.items():
        k = name_of[d]
        if "yolo" not in k:
            continue
        cnt[k] += 1
        for c, x in v.items():
            agg[k][c] += x
    print("%-46s %8s %12s %10s %14s %10s" % ("kernel", "launches", "cycles/launch", "MFMA busy", "FLOP/cyc/CU", "of 4096"))
    for k in sorted(agg, key=lambda k: -agg[k]["GRBM_GUI_ACTIVE"]):
        a, n = agg[k], cnt[k]
        cyc = a["GRBM_GUI_ACTIVE"] / 8
        if cyc <= 0:
            continue
        busy = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4)
        fl = 512.0 * a["SQ_INSTS_VALU_MFMA_MOPS_BF16"] / cyc / 256
        print("%-46s %8d %12.0f %9.1f%% %14.0f %9.1f%%" % (short(k), n, cyc / n, 100 * busy, fl, 100 * fl / 4096))


if __name__ == "__main__":
    main()
