"""Turn two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same
bench.py command, each with --kernel-trace only) into profiles/<tag>_pmc_hbm_traffic.json,
which bench.py reads for roofline.traffic.

  python profiles/make_pmc_traffic.py <dir with the passes' csv files> <tag> <batch per launch>

gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE counts half of
the bytes of wide coalesced reads, both counters are in KiB:
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    """sha256 over the HIP sources of the library: bench.py reports a traffic figure only while the kernels it
    was measured on are the kernels it runs."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "vcnf_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def main(root, tag, batch):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    kernels = {}
    for k, c in acc.items():
        if "vcnf::" not in k or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        # launches that do (almost) nothing are left out of the mean: since round 3 every split-half launch of the
        # fused RQS layer is followed by a launch of the exact fp32 kernel that only re-evaluates flagged tiles
        # (normally none) - same kernel name as the real fp32-path launches of the bench's second timed region
        def mean_of_real(v):
            top = max(v)
            real = [x for x in v if x >= 0.5 * top] if top > 0 else v
            return sum(real) / len(real), len(real)
        fe, nfe = mean_of_real(c["FETCH_SIZE"])
        wr, nwr = mean_of_real(c["WRITE_SIZE"])
        kernels[k.replace("void ", "")] = {"FETCH_SIZE_KB_mean": fe, "WRITE_SIZE_KB_mean": wr,
                                           "dispatches": len(c["FETCH_SIZE"]), "dispatches_in_mean": min(nfe, nwr),
                                           "hbm_bytes_per_launch": int((2 * fe + wr) * 1024)}
    out = {"command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- "
                      "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline",
           "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reads 1/2 of wide "
                         "coalesced streams; MI355X_MICROARCH.md, section HBM)",
           "batch_per_launch": batch, "kernel_source_sha256": kernel_source_hash(), "kernels": kernels}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "%s_pmc_hbm_traffic.json" % tag)
    json.dump(out, open(path, "w"), indent=1)
    print(path, len(kernels), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]))
