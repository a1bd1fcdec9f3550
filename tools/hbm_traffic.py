"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> profiles/<name>.json.
usage: python tools/hbm_traffic.py <dir_with_fetch_pass> <dir_with_write_pass> <out.json> [commit]
FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reads half the bytes a 16-B/lane coalesced stream fetches
(MI355X_MICROARCH.md, HBM section), hence the x2 on the read side."""
import collections
import csv
import glob
import json
import sys


def read(d, counter):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].replace(" ", "")
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    return acc


def main():
    fe, wr = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in fe:
        n = fe[k][0]
        fb = fe[k][1] * 1024 * 2 / n
        wb = wr.get(k, [n, 0.0])[1] * 1024 / n
        out[k] = {"launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
    tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in out.values())
    json.dump({"doc": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 1 --warmup 1` (2 steps incl. "
                      "warm-up); FETCH_SIZE x2 (gfx950 correction for 16-B/lane coalesced reads), KB->bytes",
               "commit": sys.argv[4] if len(sys.argv) > 4 else "unrecorded",
               "total_hbm_bytes_both_steps": tot, "kernels": out}, open(sys.argv[3], "w"), indent=1)
    print(f"total {tot / 2e9:.1f} GB per step")


if __name__ == "__main__":
    main()
