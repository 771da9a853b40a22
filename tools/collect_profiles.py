#!/usr/bin/env python3
"""Condense rocprofv3 output under gpurun_out/ into the small, committed summaries under profiles/.

    python tools/collect_profiles.py <tag> <stats_dir> <pmc_fetch_dir> <pmc_write_dir> [pairs]

stats_dir : rocprofv3 --kernel-trace --stats --output-format csv   around  bench.py --steps K
pmc_*_dir : rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) around bench.py --steps 1 --warmup 0
Writes profiles/<tag>_kernel_stats.csv (our kernels + the top torch kernels) and profiles/traffic.json
(per-launch HBM bytes per stage, read by bench.py for roofline.traffic).
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# kernel (as it appears in rocprofv3 names) -> (stage, FETCH_SIZE correction).  gfx950's FETCH_SIZE counts 64 B per 128-B
# request of a wide coalesced stream (MI355X_MICROARCH.md, HBM section), i.e. half the bytes; the factor is calibrated per
# kernel on byte counts known exactly for this workload:
#   scatter_records<unsigned long>  reads 2.6e9 records x 8 B = 20.8 GB, reports 10.4 GB            -> x2
#   bucket_count / bucket_lookup    read the same 20.8 GB (+ 4.3 GB of slices), report 10.4 / 12.6   -> x2
#   scatter_records<unsigned int>, row_hist read 4-byte words with 4 B/lane coalesced loads          -> x2 (same pattern class)
#   bucket_hist / features<NONE>    read the 1.13 GB stream with 8 B + 4 B per lane, report 0.6 GB   -> x2
#   scatter_stream                  two lanes share each 8-byte word; reports the 1.13 GB as it is   -> x1
#   features<HASH> / kmer_count     one random 8-byte slot per lane = one 64-B request each          -> x1
KERNELS = [
    # the super-k-mer pipeline (one GPU).  Factors by the same calibration: mini_scatter2 reads n_records x 12 B (8-B and 4-B
    # coalesced loads: the 8-B plane is the bulk) and mini_count streams the same planes plus the 4-B provisional words
    ("mini_plan_kernel", "plan (every step, for the next batch, side stream)", 2.0), ("round_rows_kernel", "plan (every step, for the next batch, side stream)", 1.0),
    ("mini_total_kernel", "plan (every step, for the next batch, side stream)", 1.0), ("distinct_sketch_kernel", "plan (every step, for the next batch, side stream)", 2.0),
    ("mini_scatter_kernel", "kmer_count+lookup", 2.0), ("mini_scatter2_kernel", "kmer_count+lookup", 2.0),
    ("mini_count_kernel", "kmer_count+lookup", 2.0),
    # N > 1 ranks (bench.py --rehearse-dist N): the count half keeps the stage name of the count, the rest is the exchange / features
    ("mini_lookup_half_kernel", "features", 2.0), ("mini_lookup_half_merge_kernel", "features", 2.0), ("mini_merge_bins_kernel", "exchange", 2.0), ("mini_gather_entries_kernel", "exchange", 2.0),
    ("bucket_hist_kernel", "kmer_count", 2.0), ("scan_kernel", "kmer_count", 1.0), ("digit_scan_kernel", "kmer_count", 1.0),
    ("tile_rows_kernel", "kmer_count", 1.0), ("scatter_stream_kernel", "kmer_count", 1.0),
    ("scatter_records_kernel<unsigned long", "kmer_count", 2.0), ("bucket_count_kernel", "kmer_count", 2.0),
    ("bucket_count32_kernel", "kmer_count", 2.0), ("bucket_count_emit32_kernel", "kmer_count+lookup", 2.0), ("bucket_count_compact32_kernel", "kmer_count", 2.0),
    ("bucket_count_compact_kernel", "kmer_count", 2.0),
    ("kmer_count_kernel", "kmer_count", 1.0),
    ("bucket_lookup_kernel", "features", 2.0), ("bucket_lookup32_kernel", "features", 2.0), ("scatter_records_kernel<unsigned int", "features", 2.0),
    ("row_hist_kernel", "features", 2.0), ("group_caps_kernel", "features", 1.0),
    ("features_kernel<unsigned int, 0", "features", 2.0), ("features_kernel", "features", 1.0),
    ("normalize_rows_kernel", "normalise", 2.0),
]


def short(name):
    for key, _, _ in KERNELS:
        if key in name:
            return key
    return None


def info(key):
    for k, stage, corr in KERNELS:
        if k == key:
            return stage, corr
    raise KeyError(key)


def rehearse(tag, stats_dir):
    """kernel statistics of `bench.py --rehearse-dist 8` -> profiles/<tag>_rehearse_n8_kernel_stats.csv (our kernels only)"""
    rows = list(csv.DictReader(open(sorted(glob.glob(os.path.join(stats_dir, "*", "*_kernel_stats.csv")))[-1])))
    rows = [r for r in rows if short(r["Name"])]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(os.path.join(ROOT, "profiles", f"{tag}_rehearse_n8_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs"])
        for r in rows:
            w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"]])


def main():
    if sys.argv[1] == "rehearse":
        return rehearse(sys.argv[2], sys.argv[3])
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    pairs = int(sys.argv[5]) if len(sys.argv) > 5 else 10_000_000
    pipeline = sys.argv[6] if len(sys.argv) > 6 else "mini"
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    rows = list(csv.DictReader(open(sorted(glob.glob(os.path.join(stats_dir, "*", "*_kernel_stats.csv")))[-1])))
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        ours = [r for r in rows if short(r["Name"])]
        rest = [r for r in rows if not short(r["Name"])][:6]
        for r in ours + rest:
            w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    detail = {}
    for d, ctr in ((fetch_dir, "FETCH_SIZE"), (write_dir, "WRITE_SIZE")):
        f = sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")))[-1]
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k and r["Counter_Name"] == ctr:
                e = detail.setdefault(k, {"FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0, "launches": 0})
                e[ctr + "_KB"] += float(r["Counter_Value"])
                if ctr == "FETCH_SIZE":
                    e["launches"] += 1
    stages = {}
    for k, e in detail.items():
        stage, corr = info(k)
        e["fetch_correction"] = corr
        e["raw_bytes"] = (e["FETCH_SIZE_KB"] + e["WRITE_SIZE_KB"]) * 1024          # the counters as they are
        e["hbm_bytes"] = (e["FETCH_SIZE_KB"] * corr + e["WRITE_SIZE_KB"]) * 1024    # an ESTIMATE: see the note
        e["stage"] = stage
        stages[stage] = stages.get(stage, 0.0) + e["hbm_bytes"]
    if "kmer_count+lookup" in stages:                  # fused run: the stage is the whole K2 pipeline with the lookups inside
        stages["kmer_count+lookup"] += stages.pop("kmer_count", 0.0)
    raw_stages = {}
    for e in detail.values():
        raw_stages[e["stage"]] = raw_stages.get(e["stage"], 0.0) + e["raw_bytes"]
    if "kmer_count+lookup" in raw_stages and "kmer_count" in raw_stages:
        raw_stages["kmer_count+lookup"] += raw_stages.pop("kmer_count")
    json.dump({"pairs": pairs, "tag": tag, "pipeline": pipeline, "kernels_raw": raw_stages,
               "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) around bench.py --steps 1 --warmup 0",
               "note": "counters are KiB; per kernel: raw_bytes = (FETCH_SIZE + WRITE_SIZE) x 1024 as counted; hbm_bytes = (FETCH_SIZE x "
                       "fetch_correction + WRITE_SIZE) x 1024 is an ESTIMATE: fetch_correction = 2 for kernels whose reads are "
                       "coalesced streams (MI355X_MICROARCH.md: gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced "
                       "read), 1 otherwise; checked against byte counts that are known exactly -- mini_scatter2 reads n_records x 12 B, "
                       "scatter_records<unsigned long> 2.6e9 x 8 B -- see tools/collect_profiles.py.  'kernels' = estimate per stage "
                       "(what bench.py reports as roofline.traffic), 'kernels_raw' = the raw sum; stages sum their kernels' launches of one step",
               "detail": detail, "kernels": stages}, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    for k, e in sorted(detail.items()):
        print(f"{k:40s} fetch {e['FETCH_SIZE_KB']*1024/1e9:8.2f} GB (x{e['fetch_correction']:.0f})  write {e['WRITE_SIZE_KB']*1024/1e9:8.2f} GB")
    print(stages)


if __name__ == "__main__":
    main()
