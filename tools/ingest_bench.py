"""Time-to-scene-resident for a bicycle-sized PLY (6.13 M splats, 1.52 GB): host reader + upload vs device ingest.
Tuning / measurement aid for SURVEY 8f rank 1; prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import luisacomputegaussiansplatting_amd as L

P = int(sys.argv[1]) if len(sys.argv) > 1 else 6131954
path = "/tmp/lcgs_ingest_bench.ply"
rng = np.random.default_rng(1)
rot = rng.normal(size=(P, 4)).astype(np.float32)
L.write_ply_raw(path, rng.normal(0, 2, (P, 3)).astype(np.float32), rng.normal(0.3, 0.8, (P, 3)).astype(np.float32),
                rng.normal(0, 0.1, (P, 45)).astype(np.float32), rng.normal(0, 2.5, P).astype(np.float32),
                rng.normal(-4.3, 1.1, (P, 3)).astype(np.float32), rot)
size = os.path.getsize(path)
r = L.Renderer(L.Context(0))
r.load_ply(path)  # warm: page cache, allocations, code objects
out = {"splats": P, "file_bytes": size}
for name, fn in (("device_ingest", lambda: r.load_ply(path)),
                 ("host_read_plus_upload", lambda: r.upload_scene(L.read_gs_ply(path)))):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    out[name + "_s"] = round(min(ts), 4)
    out[name + "_GBps"] = round(size / min(ts) / 1e9, 2)
os.remove(path)
print(json.dumps(out))
