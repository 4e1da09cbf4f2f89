"""CPU: bench.py's multi-rank path (`--gpus 2`, launched exactly as the driver launches it, but through
tests/bench_standin.py: gloo instead of RCCL, an oracle-backed index instead of the HIP one).  Checks the contract of the
JSON line and that the sharded search reproduces the single-rank result bit for bit."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, extra, launcher=True):
    flags = ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--n-rows", "30011", "--queries", "37", "--no-encoder",
             "--no-cpu-baseline", "--no-other-kernels", "--no-larger-k", "--no-config5", "--no-real-size", "--no-stage", "--no-certificate-leg", "--no-splade"] + extra
    script = os.path.join(ROOT, "tests", "bench_standin.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    if world == 1 or not launcher:
        cmd = [sys.executable, script] + flags
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), script] + flags
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 3])
def test_bench_line_of_a_multi_rank_run(world):
    r = _run(world, [])
    assert r["n_gpus"] == world and r["steps"] == 2 and r["warmup"] == 1
    assert r["unit"] == "queries/s" and r["higher_is_better"] is True and r["scaling"] == "strong"
    assert r["value"] > 0 and abs(r["value"] - 37 / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]
    assert r["parity_check"] == "pass"          # planted positives on top of the merged lists, sorted scores
    assert r["config"]["rows_per_gpu"] == -(-30011 // world)
    assert f"row-shard x{world}" in r["config"]["parallelism"]
    for key in ("roofline", "kernel_ms_per_step", "uncertified_queries", "vs_baseline", "dtype", "data", "metric"):
        assert key in r


def test_gpus_n_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` typed by hand (no torchrun, no WORLD_SIZE): bench.launch_ranks_if_needed starts the ranks with
    the driver's own command instead of dying on an assert."""
    r = _run(2, [], launcher=False)
    assert r["n_gpus"] == 2 and r["parity_check"] == "pass" and "row-shard x2" in r["config"]["parallelism"]


def test_single_rank_standin_agrees():
    r = _run(1, [])
    assert r["n_gpus"] == 1 and r["parity_check"] == "pass"
    # the streaming full-list gate (a float64 GEMM per block, no oracle) agrees with the oracle-backed index on complete lists
    assert r["full_list_gate"]["ids_and_fp32_scores_bit_exact"] is True and r["full_list_gate"]["queries"] == 32


def test_power_sampler_without_a_sensor_reports_nothing_and_does_not_raise():
    """bench.py samples the board's power sensor behind its timed region; in a container without an AMD hwmon node (here) the
    sampler must come back empty-handed, never fail the bench line."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    smp = bench.PowerSampler(period=0.01)
    smp.start()
    time.sleep(0.05)
    got = smp.stop(skip_s=0.0)
    assert isinstance(got, dict) and got.get("samples", 0) >= 0 and "error" not in got
    assert smp.cap_watts() is None or smp.cap_watts() > 0


def test_committed_pmc_traffic_is_reported_only_for_the_kernel_source_it_was_measured_on(tmp_path):
    """roofline.traffic comes from profiles/hbm_traffic.json (rocprofv3 --pmc, separate passes).  bench.pmc_traffic hands an
    entry out only for exactly the kernel + geometry asked for and only while the kernel's source file still has the sha256
    the entry was collected under — a kernel edit without a re-profile must show null, not the old number."""
    import hashlib
    sys.path.insert(0, ROOT)
    import bench
    committed = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    table = json.load(open(committed))["kernels"]
    assert "bh_scan_topk256_kernel@768" in table and "bh_csr_scan_mfma_kernel@30522" in table
    for key, ent in table.items():
        src = os.path.join(ROOT, "bergen_amd", "csrc", bench.KERNEL_SOURCES[ent["kernel"]])
        fresh = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16] == ent["source_sha16"]
        variant = "paired" if "/paired@" in key else None  # (the paired launch of the 256-query kernel has an entry of its own)
        got = bench.pmc_traffic(committed, ent["kernel"], ent["n_rows"], ent["dim"], variant=variant)
        assert (got == ent["hbm_bytes_per_launch"]) if fresh else (got is None), key
        assert bench.pmc_traffic(committed, ent["kernel"], ent["n_rows"] + 1, ent["dim"], variant=variant) is None      # another corpus size
        assert bench.pmc_traffic(committed, ent["kernel"], ent["n_rows"], ent["dim"] + 64, variant=variant) is None     # another geometry
    # a stale hash
    k0 = "bh_scan_topk256_kernel@768"
    stale = {"kernels": {k0: dict(table[k0], source_sha16="0" * 16)}}
    p = tmp_path / "t.json"
    p.write_text(json.dumps(stale))
    assert bench.pmc_traffic(str(p), "bh_scan_topk256_kernel", table[k0]["n_rows"], 768) is None
    assert bench.pmc_traffic(str(tmp_path / "missing.json"), "bh_scan_topk256_kernel", 1, 768) is None


def test_scan_roofline_accounts_for_paired_unpaired_and_tail_launches():
    """bench.scan_roofline: the roofline object names the DOMINANT launch of a step.  With paired launches (index.hip option
    pair256: two query-tile passes per launch) the units per launch are 2 passes, so SURVEY §8d's per-pass bytes count twice per
    launch while the corpus leaves HBM once; the unpaired launch of an odd pass count and the 128-query tail pass are reported
    beside it, each from its own HIP-event time; and the gate's query choice covers both halves of a paired launch."""
    sys.path.insert(0, ROOT)
    import bench
    n, d, k, steps = 21_000_000, 768, 50, 4
    per_pass = n * d * 2.0 + 256 * d * 2.0 + 256 * k * 12.0
    c = {"query_tile": 256, "n_passes": 12, "paired_launches": 5, "tail_query_tile": 128, "shader_mhz": 1500.0}
    acc = {"scan_ms": steps * (5 * 13.4 + 7.3 + 5.0), "paired_scan_ms": steps * 5 * 13.4, "tail_scan_ms": steps * 5.0}
    r = bench.scan_roofline(acc, c, steps, n, d, k, os.path.join(ROOT, "profiles", "missing.json"))
    assert r["passes_per_launch"] == 2 and r["launches"] == 5 * steps and abs(r["avg_launch_ms"] - 13.4) < 1e-9
    assert r["algorithmic_bytes_per_launch"] == 2 * per_pass and r["algorithmic_bytes_per_pass"] == per_pass
    assert abs(r["achieved"] - 2 * per_pass / 13.4e-3 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert r["hbm_bytes_needed_per_launch"] == n * d * 2.0 + 2 * (256 * d * 2.0 + 256 * k * 12.0) and r["traffic"] is None
    assert r["unpaired_launch"]["launches"] == steps and abs(r["unpaired_launch"]["avg_launch_ms"] - 7.3) < 1e-9
    assert abs(r["tail_pass"]["avg_launch_ms"] - 5.0) < 1e-9
    # the two rooflines side by side, each self-consistent; `bound` names the one that binds (round-5 review: "hbm" stood beside binding "mfma")
    assert r["hbm"]["unit"] == "GB/s" and r["hbm"]["frac"] == r["frac"] and r["hbm"]["peak"] == 8000.0
    assert r["mfma"]["unit"] == "TFLOP/s" and r["mfma"]["frac"] == r["mfma_frac"] and abs(r["mfma"]["achieved"] - 2 * 2.0 * 256 * n * d / 13.4e-3 / 1e12) < 1e-6
    assert r["bound"] == r["binding_resource"] == ("mfma" if r["mfma_frac"] >= r["hbm_frac_of_needed_bytes"] else "hbm")
    assert r["mfma_busy_frac"] is None or 0.0 < r["mfma_busy_frac"] < 1.0  # (from profiles/sq_counters.json when the kernel source is unchanged)
    # the figure that cannot be misread: the larger of (bytes the launch must move) / time / peak and flops / time / peak
    assert r["algorithmic_frac"] == r["frac"] and r["binding_resource"] == "mfma"
    assert abs(r["frac_binding"] - 2.0 * 2 * 256 * n * d / 13.4e-3 / 1e12 / 2500.0) < 1e-9 and r["frac_binding"] < r["frac"]
    # the balanced remainder (index.hip option balance_tail): six paired launches, the sixth with 277 queries — reported beside the dominant
    # launch and kept out of its average
    cb = {"query_tile": 256, "n_passes": 12, "paired_launches": 6, "tail_query_tile": 0, "shader_mhz": 1500.0, "balanced_queries": 277}
    rb = bench.scan_roofline({"scan_ms": steps * (5 * 13.4 + 8.0), "paired_scan_ms": steps * (5 * 13.4 + 8.0), "tail_scan_ms": 0.0,
                              "balanced_scan_ms": steps * 8.0}, cb, steps, n, d, k, "missing.json")
    assert rb["launches"] == 5 * steps and abs(rb["avg_launch_ms"] - 13.4) < 1e-9 and rb["tail_pass"] is None and "unpaired_launch" not in rb
    assert rb["balanced_launch"]["queries"] == 277 and abs(rb["balanced_launch"]["avg_launch_ms"] - 8.0) < 1e-9
    assert rb["launches_of_this_kernel"] == 6 * steps and abs(rb["avg_launch_ms_over_all_launches_of_this_kernel"] - (5 * 13.4 + 8.0) / 6) < 1e-9
    # pairing off (or a library that does not report it): one pass per launch, the tail pass taken out of the average
    c1 = {"query_tile": 256, "n_passes": 12, "tail_query_tile": 128, "shader_mhz": 1500.0}
    r1 = bench.scan_roofline({"scan_ms": steps * (11 * 7.3 + 5.0), "tail_scan_ms": steps * 5.0}, c1, steps, n, d, k, "missing.json")
    assert r1["passes_per_launch"] == 1 and r1["launches"] == 11 * steps and abs(r1["avg_launch_ms"] - 7.3) < 1e-9 and "unpaired_launch" not in r1
    assert r1["algorithmic_bytes_per_launch"] == per_pass
    # d = 1024: the 256-query kernel's tile is 128 queries (no tail routing), four paired launches for 1 000 queries
    c5 = {"query_tile": 128, "n_passes": 8, "paired_launches": 4, "tail_query_tile": 0, "shader_mhz": 1400.0}
    r5 = bench.scan_roofline({"scan_ms": 4 * 13.0, "paired_scan_ms": 4 * 13.0, "tail_scan_ms": 0.0}, c5, 1, n, 1024, 200, "missing.json")
    assert r5["kernel"] == "bh_scan_topk256_kernel" and r5["launches"] == 4 and r5["tail_pass"] is None and "unpaired_launch" not in r5
    # the gate looks into both halves of the first and the last paired launch, the unpaired pass and the tail pass
    idx = bench.gate_queries(2837, c, 32)
    passes = {i // 256 for i in idx}
    assert {0, 1, 9, 10, 11} <= passes and len(idx) <= 32
    assert {i // 256 for i in bench.gate_queries(2837, c1, 32)} == {0, 6, 10, 11}


def test_secondary_summary_flattens_the_other_legs_into_the_roofline_object():
    """The driver's record keeps `roofline` in full and only the key names of the other legs: bench.secondary_summary puts their
    headline scalars inside it; a leg that did not run (or failed) shows up as None, never as a KeyError."""
    sys.path.insert(0, ROOT)
    import bench
    out = {"passages_per_s": 35000.123456, "encoder_roofline": {"frac": 0.33}, "encoder": {"ms_per_step_kernels": 14.2, "steps": 10},
           "splade_search": {"queries_per_s": 16000.0, "queries": 2837, "roofline": {"frac": 0.52},
                             "full_list_gate": {"ids_and_fp32_scores_bit_exact": True}},
           "config5": {"error": "boom"}, "parity_check": "pass",
           "rerank": {"bert_large_shape": {"pairs_per_s": 5000.0, "roofline": {"frac": 0.1}}}}
    sec = bench.secondary_summary(out)
    assert sec["passages_per_s"] == 35000.1235 and sec["encoder_frac"] == 0.33 and sec["splade_full_list_gate"] is True
    assert sec["config5_queries_per_s"] is None and sec["e5_large_frac"] is None and sec["rerank_bert_frac"] == 0.1
    assert sec["rerank_deberta_pairs_per_s"] is None and sec["parity_check_all_legs"] == "pass" and sec["splade_queries"] == 2837
    import json
    assert len(json.dumps(sec)) < 2000
