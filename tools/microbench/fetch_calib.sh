# FETCH_SIZE calibration for this library's access patterns: gpurun -- bash tools/microbench/fetch_calib.sh
# -> gpurun_out/fetch_calib.txt (copy to profiles/rNN_fetch_calibration.txt)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
set -o pipefail
hipcc -O3 --offload-arch=gfx950 tools/microbench/fetch_calib.hip -o tools/microbench/fetch_calib || exit 1
O=gpurun_out/fetch_calib
rm -rf $O && mkdir -p $O
./tools/microbench/fetch_calib > $O/plain.txt || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc -- ./tools/microbench/fetch_calib > $O/pmc_run.txt 2>&1 || exit 1
python3 - <<'PY' > gpurun_out/fetch_calib.txt
import csv, glob, re, collections
req = {}
for line in open("gpurun_out/fetch_calib/plain.txt"):
    m = re.match(r"(\S+)\s+requested_bytes (\d+)\s+best ([\d.]+) us\s+([\d.]+) TB/s", line)
    if m: req[m.group(1)] = (float(m.group(2)), float(m.group(3)), float(m.group(4)))
acc, n = collections.defaultdict(float), collections.defaultdict(set)
for f in glob.glob("gpurun_out/fetch_calib/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "FETCH_SIZE": continue
        k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        acc[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
print("# FETCH_SIZE (rocprofv3, KB per dispatch x 1024) against the bytes each pattern REQUESTS, 1 GiB table, one MI355X")
print("# factor = requested / FETCH_SIZE: what the counter has to be multiplied by to give requested bytes;")
print("# for gathers the memory system moves whole sectors, so FETCH_SIZE / accesses = bytes counted per access")
print("%-36s %14s %14s %8s %12s %10s" % ("pattern", "requested_MB", "FETCH_SIZE_MB", "factor", "us (no pmc)", "TB/s req"))
for k, (rb, us, tbs) in req.items():
    kk = [x for x in acc if x.replace(" ", "") == k.replace(" ", "")]
    fs = acc[kk[0]] / len(n[kk[0]]) * 1024 if kk else float("nan")
    print("%-36s %14.1f %14.1f %8.3f %12.1f %10.2f" % (k, rb / 1e6, fs / 1e6, rb / fs if fs else float("nan"), us, tbs))
PY
cat gpurun_out/fetch_calib.txt
