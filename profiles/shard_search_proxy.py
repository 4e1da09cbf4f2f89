"""One-GPU proxy of the WHOLE 8-GPU ShardedSearcher.search (round-5 review item 5b), not of the scan alone: what rank 0 does per search at
world size 8, each piece measured on this GPU —
  (1) the local search over an eighth of the corpus, results written straight into the packed send buffer (ShardedSearcher over a
      one-rank RCCL group: the real all_gather_into_tensor / broadcast code path, on device buffers);
  (2) the all-gather itself: measured at world size 1 (launch + RCCL's fixed cost), plus the payload of 7 peers over ONE xGMI link at
      153 GB/s as the transfer estimate (the gather is latency-bound: 12 Q k + 8 bytes per rank);
  (3) the unpacking copies + bh_merge_topk over EIGHT partial lists (synthetic peers: this shard's lists with other id offsets and
      perturbed scores, so that the merge really interleaves) + the broadcast.
Projected step = (1) + (2) + (3); speed-up = the full-corpus one-GPU search / projected step.
    python profiles/shard_search_proxy.py [dim k queries] out=profiles/r06_shard_search_proxy.json"""
import json, os, statistics, sys, time
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, bergen_amd
from bergen_amd import _lib
from bergen_amd.index import merge_topk

av = [int(x) for x in sys.argv[1:] if "=" not in x]
OUT_PATH = next((x.split("=", 1)[1] for x in sys.argv[1:] if x.startswith("out=")), None)  # (RCCL prints its banner on stdout)
dim, k, nq = (av + [768, 50, 2837][len(av):])[:3]
G, n_total = 8, 21_000_000
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
_lib.init(0)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
q = bench.make_queries(nq, dim, dev)


def timed(fn, reps=7):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts), min(ts)


out = {"workload": f"{nq} queries x 21 M x {dim} fp16, top-{k}; world size {G} projected from one GPU", "pieces_ms": {}}
# full corpus on one GPU (the baseline of the speed-up)
ix = bergen_amd.FlatIndex(n_total, dim, metric="ip", device=0)
bench.fill_shard(ix, 0, n_total, dim, q, n_total, dev)
ix.finalize()
full_ms, _ = timed(lambda: ix.search(q, k), reps=5)
ix.close()
out["full_corpus_one_gpu_ms"] = full_ms
# (1) + (2, fixed part) + (3, one list): the real ShardedSearcher path on a one-rank RCCL group
lo, hi = bergen_amd.shard_range(n_total, 0, G)
sh = bergen_amd.FlatIndex(hi - lo, dim, metric="ip", device=0)
bench.fill_shard(sh, lo, hi, dim, q, n_total, dev)
sh.finalize()
ss = bergen_amd.ShardedSearcher(sh, lo, rank=0, world_size=1, exercise_collective=True)
whole1_ms, _ = timed(lambda: ss.search(q, k, broadcast=True))
local_ms, _ = timed(lambda: sh.search(q, k, id_offset=lo))
out["pieces_ms"]["local_search_eighth"] = local_ms
out["pieces_ms"]["sharded_search_world1_gather_merge1_broadcast"] = whole1_ms
fixed_collectives_ms = max(0.0, whole1_ms - local_ms)
# (3) the merge over EIGHT lists + the unpacking copies, as ShardedSearcher does them
s1, i1 = sh.search(q, k, id_offset=lo)
s1, i1 = torch.as_tensor(s1).to(dev), torch.as_tensor(i1).to(dev)
gen = torch.Generator(device=dev).manual_seed(5)
# (canonical order inside every synthetic list, like a real shard's: score descending, ties by ascending id — a list that is not makes the
# merge kernel count all pairs for that query, which a real search never triggers)
all_s = torch.stack([s1 + 1e-3 * torch.randn(s1.shape, generator=gen, device=dev) for _ in range(G)])
all_i = torch.stack([i1 + r * (hi - lo) for r in range(G)])
order = torch.argsort(all_i, dim=2, stable=True)
all_s, all_i = torch.gather(all_s, 2, order), torch.gather(all_i, 2, order)
order = torch.argsort(all_s, dim=2, descending=True, stable=True)
all_s, all_i = torch.gather(all_s, 2, order).contiguous(), torch.gather(all_i, 2, order).contiguous()
per = nq * k * 12 + 8
flat = torch.empty(G * per + 64, dtype=torch.uint8, device=dev)
buf_s = torch.empty_like(all_s)
buf_i = torch.empty_like(all_i)
res = (torch.empty((nq, k), dtype=torch.float32, device=dev), torch.empty((nq, k), dtype=torch.int64, device=dev))


def merge8():
    buf_s.copy_(all_s)  # (the unpacking of the gathered byte buffer into the dense [G, Q, k] forms)
    buf_i.copy_(all_i)
    merge_topk(buf_s, buf_i, out=res)


merge8_ms, _ = timed(merge8)
merge1_ms, _ = timed(lambda: merge_topk(buf_s[:1], buf_i[:1], out=res))
out["pieces_ms"]["unpack_and_merge_8_lists"] = merge8_ms
out["pieces_ms"]["unpack_and_merge_1_list"] = merge1_ms
payload = 7 * per
out["pieces_ms"]["gather_payload_7_peers_one_xgmi_link_estimate"] = payload / 153e9 * 1e3
projected = local_ms + fixed_collectives_ms + (merge8_ms - merge1_ms) + payload / 153e9 * 1e3
out["projected_world8_step_ms"] = projected
out["projected_speedup_whole_search"] = full_ms / projected
out["scan_only_speedup_for_comparison"] = full_ms / local_ms
out["note"] = ("fixed cost of the two collectives and the one-list merge = world-1 ShardedSearcher.search minus the bare local search; the "
               "8-list merge replaces the 1-list merge; RCCL's per-peer latency at world 8 is NOT measurable on one GPU and is not in the figure")
if OUT_PATH:
    json.dump(out, open(OUT_PATH, "w"), indent=1)
else:
    print(json.dumps(out, indent=1))
dist.destroy_process_group()
