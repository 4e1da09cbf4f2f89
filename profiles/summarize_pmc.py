#!/usr/bin/env python
"""Aggregate rocprofv3 counter_collection CSVs into a small per-kernel summary (run on the GPU box).
usage: summarize_pmc.py <prof_dir> <out.json> [n_rows dim [sparse_docs vocab]]
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB-like
units of 1024 B in rocprofv3's derived metric... we report the RAW counter and the corrected bytes:
  read_bytes  = FETCH_SIZE * 1024 * 2   (gfx950: FETCH_SIZE reports exactly 1/2 of a wide coalesced stream)
  write_bytes = WRITE_SIZE * 1024       (uncalibrated, small here)
"""
import csv, glob, json, os, sys
from collections import defaultdict

prof, out = sys.argv[1], sys.argv[2]
res = {}
for sub in sorted(os.listdir(prof)):
    for f in glob.glob(os.path.join(prof, sub, "*counter_collection.csv")):
        agg = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?").split("(")[0][:60]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in agg.items():
            for c, vals in cs.items():
                # a kernel may also be launched on a sliver of the data in the same run (the sparse scan's pre-pass over a short
                # prefix of the corpus): "mean" is over the FULL-SIZE launches (>= half the largest value), "mean_all" over all
                big = [v for v in vals if v >= 0.5 * max(vals)] or vals
                res.setdefault(k, {})[c] = {"launches": len(big), "mean": sum(big) / len(big), "min": min(vals), "max": max(vals),
                                            "launches_all": len(vals), "mean_all": sum(vals) / len(vals)}
summary = {"per_kernel": res}
# The streaming kernels of the run: the dense scans (bench.py times the default kernel as the headline, the earlier ones as
# `other_kernels`, the d = 1024 instantiation in its config5 leg) and the sparse (SPLADE) scan.  hbm_traffic.json is keyed by
# "<kernel>@<dim>" so that a roofline.traffic in bench.py's line is the entry of exactly the kernel + geometry it names, and
# every entry carries the sha256 of the kernel's source file at collection time: bench.py reports null for an entry whose
# source has changed since (a kernel edit without a re-profile must not keep the old number).
import hashlib
import re

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "bergen_amd", "csrc")
SOURCES = {"bh_scan_topk256_kernel": "scan_topk256.hip", "bh_scan_topk192_kernel": "scan_topk192.hip", "bh_scan_topk_kernel": "scan_topk.hip",
           "bh_csr_scan_mfma_kernel": "csr_mfma.hip"}


def base_name(k):
    for name in SOURCES:
        if name in k:
            return name
    return None


def dim_of(name, k, default_dim):
    """The padded dim from the kernel's first template argument (NK32 * 32 for the 16x16x32 kernels, NK * 16 for the
    32x32x16 kernel); the sparse kernel streams CSR entries: its 'dim' is the vocabulary given on the command line."""
    m = re.search(r"<\s*(\d+)", k)
    if name == "bh_csr_scan_mfma_kernel" or not m:
        return default_dim
    return int(m.group(1)) * (16 if name == "bh_scan_topk_kernel" else 32)


def sha16(name):
    try:
        return hashlib.sha256(open(os.path.join(CSRC, SOURCES[name]), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


n_rows = int(sys.argv[3]) if len(sys.argv) > 4 else None
dim_arg = int(sys.argv[4]) if len(sys.argv) > 4 else None
sparse_docs = int(sys.argv[5]) if len(sys.argv) > 6 else None
sparse_vocab = int(sys.argv[6]) if len(sys.argv) > 6 else None
scan = {k: cs for k, cs in res.items() if base_name(k) and "FETCH_SIZE" in cs}
traffic = {}
for k, cs in scan.items():
    rd = cs["FETCH_SIZE"]["mean"] * 1024 * 2
    wr = cs.get("WRITE_SIZE", {"mean": 0})["mean"] * 1024
    name = base_name(k)
    sparse = name == "bh_csr_scan_mfma_kernel"
    dim = dim_of(name, k, sparse_vocab if sparse else dim_arg)
    # the paired launch of the 256-query kernel (option pair256: two passes per launch, template argument ABL = 128) has its own entry
    tpl = k.split("<", 1)[1].split(",") if "<" in k else []
    paired = name == "bh_scan_topk256_kernel" and len(tpl) > 6 and tpl[6].strip() == "128"
    key = f"{name}/paired@{dim}" if paired else f"{name}@{dim}"
    # a run may hold several instantiations of one kernel at one dim (ablations, the sparse pre-pass): keep the one that moves
    # the most bytes per launch
    if key in traffic and traffic[key]["hbm_bytes_per_launch"] >= rd + wr:
        continue
    traffic[key] = {"kernel": name, "kernel_full": k, "launches": cs["FETCH_SIZE"]["launches"], "read_corrected_x2": rd, "write": wr,
                    "hbm_bytes_per_launch": rd + wr, "fetch_size_raw": cs["FETCH_SIZE"]["mean"], "dim": dim,
                    "n_rows": sparse_docs if sparse else n_rows, "source_sha16": sha16(name)}
summary["scan_hbm_bytes_per_launch"] = traffic
if traffic and n_rows is not None:
    json.dump({"kernels": traffic,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), read side x2 per MI355X_MICROARCH.md; "
                         "source_sha16 = sha256 of the kernel's .hip file when the counters were collected"},
              open(os.path.join(os.path.dirname(out), "hbm_traffic.json"), "w"), indent=1)
if res or not os.path.exists(out):  # (a trace-only run has no counter files: keep the summary of the last counter run)
    json.dump(summary, open(out, "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
