"""HBM-side traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command.

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> [out.json]

Counter unit: KiB.  gfx950 correction: FETCH_SIZE counts 64 B per 128-B request of 16-byte-per-lane loads, so
read bytes = 2 * FETCH_SIZE * 1024 (exact for the conv kernels' LDS-DMA loads); WRITE_SIZE is exact.  Infinity-
Cache hits are counted, so this is traffic leaving L2: an upper bound on DRAM bytes.  Prints a table; the JSON
(read by bench.py) holds the figures of the kernel named by --kernel-substring (default: the conv instance with
the most launches x bytes) and, under "postproc_kernels", those of the decode / NMS / gather kernels."""
import csv
import json
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    by_dispatch = defaultdict(float)
    name_of = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            d = r["Dispatch_Id"]
            by_dispatch[d] += float(r["Counter_Value"])
            name_of[d] = r["Kernel_Name"]
    tot, cnt = defaultdict(float), defaultdict(int)
    for d, v in by_dispatch.items():
        tot[name_of[d]] += v
        cnt[name_of[d]] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def short(name):
    """The name yolo_conv_kernel_name() gives the instance (what bench.py looks up), from the profiler's mangled or
    demangled kernel name; template arguments: element type, tile channels, tile pixels, wave layout (2), two taps per
    k-step, waves, BatchNorm sums, stride-2 data gradient, strip form (and, not part of the name, the shared-column
    form of tiles with an odd number of pixel fragment columns: it is implied by 176)."""
    m = re.search(r"conv_igemm_kernelI(DF16b|f)Li(\d+)ELi(\d+)ELi(\d)ELi(\d)ELb(\d)ELi\d+ELb(\d)ELb(\d)ELb(\d)", name)
    if m:
        g = list(m.groups())
    else:
        m = re.search(r"conv_igemm_kernel<([^,]+), *(\d+), *(\d+), *(\d), *(\d), *(\w+), *\d+, *(\w+), *(\w+), *(\w+)(?:, *\w+)?>", name)
        if not m:
            return re.sub(r"^void |\(.*$", "", name).replace("yolo::", "")[:44]
        g = list(m.groups())
        g[0] = "DF16b" if "bfloat" in g[0] else "f"
        for i in (5, 6, 7, 8):
            g[i] = "1" if g[i] in ("true", "1") else "0"
    form = "strip" if g[8] == "1" else ("tap2" if g[5] == "1" else "plain")
    extra = ("+bnsums" if g[6] == "1" else "") + ("+s2dgrad" if g[7] == "1" else "")
    return "conv_igemm_kernel<%s,%s,%s,%s,%s,%s>%s" % ("bf16" if g[0] == "DF16b" else "float", g[1], g[2], g[3], g[4], form, extra)


def kernel_source_sha():
    """The same hash bench.py computes: which kernel sources the record belongs to."""
    import glob
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha1()
    for path in sorted(glob.glob(os.path.join(root, "yolo_amd", "csrc", "conv.hip")) +
                       glob.glob(os.path.join(root, "yolo_amd", "csrc", "*.h"))):
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def postproc_source_sha():
    """Hash of the decode / NMS kernel sources: which sources the post-processing rows of the record belong to."""
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha1()
    for name in ("decode.hip", "nms.hip", "common.h"):
        with open(os.path.join(root, "yolo_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k in fetch:
        if k in write and ("yolo" in k):
            f, n = fetch[k]
            w, _ = write[k]
            rows.append((short(k), n, f, w, (2 * f + w) * 1024))
    rows.sort(key=lambda r: -r[1] * r[4])
    print("%-46s %8s %15s %15s %18s" % ("kernel", "launches", "FETCH_SIZE KiB", "WRITE_SIZE KiB", "HBM bytes/launch"))
    for r in rows:
        print("%-46s %8d %15.1f %15.1f %18d" % r)
    if len(sys.argv) > 3:
        convs = [r for r in rows if r[0].startswith("conv_igemm_kernel<bf16,128")]
        recs = {r[0]: dict(kernel=r[0], hbm_bytes_per_launch=round(r[4]), launches_sampled=r[1],
                           fetch_size_kib_per_launch=round(r[2], 1), write_size_kib_per_launch=round(r[3], 1)) for r in convs}
        # decode / NMS / gather kernels: counter bytes per launch, for bench.py's post-processing block
        post = {r[0]: dict(hbm_bytes_per_launch=round(r[4]), launches_sampled=r[1]) for r in rows
                if r[0].startswith(("decode_", "nms_", "gather_"))}
        with open(sys.argv[3], "w") as f:
            json.dump(dict(kernels=recs, kernel_source_sha=kernel_source_sha(), postproc_kernels=post,
                           postproc_source_sha=postproc_source_sha(), source="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of bench.py; "
                                                "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024"), f, indent=1)


if __name__ == "__main__":
    main()
