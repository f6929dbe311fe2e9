#!/usr/bin/env python3
"""Builds rNN_pmc_traffic.{json,txt} from two rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; separate
passes, as MI355X_MICROARCH.md prescribes).  usage: make_pmc_traffic.py <fetch_dir> <write_dir> <out_prefix> <cmd>"""
import csv, glob, json, os, re, sys, collections

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import kernel_key
from bench import kernel_sources_hash, library_hash  # what these counters were measured on (bench.py refuses a stale file)


def per_kernel(d, counter):
    acc, n = collections.defaultdict(float), collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = kernel_key(r["Kernel_Name"])
            acc[k] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
    return {k: (acc[k] / len(n[k]), len(n[k])) for k in acc}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), " + sys.argv[4] +
                ", workload synth_unbounded(seed=2001,P=6131954) 1920x1080, one MI355X; counter unit KB, averaged per "
                "launch. fetch_bytes_x2 applies the gfx950 FETCH_SIZE correction for wide streaming reads "
                "(MI355X_MICROARCH.md HBM section); it is uncalibrated for the narrow gathers of expand/render.",
       "kernel_sources_sha256": kernel_sources_hash(), "library_sha256": library_hash(), "kernels": {}}
lines = ["# HBM traffic per launch from PMC counters (see the .json _note)",
         "%-24s %8s %14s %14s %14s" % ("kernel", "launches", "fetch_raw_MB", "fetch_x2_MB", "write_MB")]
for k in fetch:
    if not k.startswith("k_"):
        continue
    fb, n = fetch[k][0] * 1024, fetch[k][1]
    wb = write.get(k, (0, 0))[0] * 1024
    out["kernels"][k] = {"launches": n, "fetch_bytes_raw": round(fb), "fetch_bytes_x2": round(2 * fb), "write_bytes": round(wb)}
    lines.append("%-24s %8d %14.1f %14.1f %14.1f" % (k, n, fb / 1e6, 2 * fb / 1e6, wb / 1e6))
# launches per forward frame (k_cull_compact runs once per frame, forward-only and keep_state frames alike); the backward's
# and the optimiser's kernels are not part of the forward frame
frames = out["kernels"].get("k_cull_compact", {}).get("launches", 0)
not_forward = ("k_render_backward", "k_preprocess_backward", "k_zero_grads2d", "k_adam", "k_slice_bounds", "k_tile_order")
if frames:
    out["forward_frame_launches"] = {k: round(v["launches"] / frames) for k, v in out["kernels"].items()
                                     if not k.startswith(not_forward) and round(v["launches"] / frames) >= 1}
json.dump(out, open(sys.argv[3] + ".json", "w"), indent=1)
open(sys.argv[3] + ".txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
