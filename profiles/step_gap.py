"""Where the host-side time of a headline step goes: python profiles/step_gap.py"""
import os, sys, time, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, bergen_amd
from bergen_amd import _lib
_lib.init(0)
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 21_000_000
q = bench.make_queries(2837, 768, dev)
ix = bergen_amd.FlatIndex(n, 768, metric="ip", device=0)
bench.fill_shard(ix, 0, n, 768, q, n, dev)
ix.finalize()
for _ in range(3):
    ix.search(q, 50, host=True)
rows = []
for _ in range(10):
    t0 = time.perf_counter(); s, i = ix.search(q, 50, host=True); t1 = time.perf_counter()
    s2, i2 = s.clone(), i.clone(); t2 = time.perf_counter()
    c = ix.counters(); t3 = time.perf_counter()
    rows.append(((t1 - t0) * 1e3, c["total_ms"], (t2 - t1) * 1e3, (t3 - t2) * 1e3))
for name, j in (("search() wall ms", 0), ("kernels total_ms", 1), ("clone ms", 2), ("counters ms", 3)):
    v = [r[j] for r in rows]
    print(f"{name:18s} median {statistics.median(v):8.3f}  min {min(v):8.3f}  max {max(v):8.3f}")
print("pinned clone is_pinned:", s2.is_pinned())
# the same through the stage (Retrieve.search_rows: what bench.py times), and with paired launches off
stage = bench.HipEnv(0).make_stage(0, 1)
key = "bench://synthetic-corpus"
stage.adopt_resident_index(key, ix, n, "ip", rows=None)
for pair in (1, 0, 1):
    ix.set_option("pair256", pair)
    for _ in range(2):
        stage.search_rows(q, key, 50, "ip", n)
    wall, kern = [], []
    for _ in range(8):
        t0 = time.perf_counter(); r = stage.search_rows(q, key, 50, "ip", n); t1 = time.perf_counter()
        wall.append((t1 - t0) * 1e3); kern.append(ix.counters()["total_ms"])
    print(f"stage.search_rows pair256={pair}: wall median {statistics.median(wall):8.3f} min {min(wall):8.3f} max {max(wall):8.3f}   kernels median {statistics.median(kern):8.3f}")
# what the 1.7 MB of result copies cost by themselves: fresh tensors (malloc / free every time) vs one preallocated pair
s, i = ix.search(q, 50, host=True)
pre = (torch.empty_like(s, pin_memory=False), torch.empty_like(i, pin_memory=False))
for label in ("clone + free", "copy_ into preallocated"):
    ts = []
    keep = None
    for _ in range(200):
        t0 = time.perf_counter()
        if label.startswith("clone"):
            keep = (s.clone(), i.clone())
        else:
            pre[0].copy_(s); pre[1].copy_(i)
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(f"{label:26s} median {ts[100]:7.3f}  p90 {ts[180]:7.3f}  p99 {ts[198]:7.3f}  max {ts[-1]:7.3f} ms")
def _read(p):
    try:
        return open(p).read().strip().replace("\n", " | ")
    except OSError as e:
        return f"({e.__class__.__name__})"
print("loadavg", open("/proc/loadavg").read().strip(), "cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
print("cgroup cpu.max", _read("/sys/fs/cgroup/cpu.max"), "| cpu.stat", _read("/sys/fs/cgroup/cpu.stat"))
print("THP", _read("/sys/kernel/mm/transparent_hugepage/enabled"), "| defrag", _read("/sys/kernel/mm/transparent_hugepage/defrag"))
print("MALLOC env", {k: v for k, v in os.environ.items() if k.startswith("MALLOC_")})
