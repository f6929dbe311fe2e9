#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 --pmc counters (counter_collection.csv), one line per kernel."""
import csv, glob, os, re, sys, collections

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench import kernel_sources_hash, library_hash
from kernel_names import kernel_key

print("# kernel_sources_sha256", kernel_sources_hash())
print("# library_sha256", library_hash())
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = kernel_key(r["Kernel_Name"])
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[n].add(r["Dispatch_Id"])
for n in acc:
    k = max(1, len(cnt[n]))
    print("%-24s launches=%d %s" % (n, k, {c: round(v / k) for c, v in sorted(acc[n].items())}))
