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
        got = bench.pmc_traffic(committed, ent["kernel"], ent["n_rows"], ent["dim"])
        assert (got == ent["hbm_bytes_per_launch"]) if fresh else (got is None), key
        assert bench.pmc_traffic(committed, ent["kernel"], ent["n_rows"] + 1, ent["dim"]) is None      # another corpus size
        assert bench.pmc_traffic(committed, ent["kernel"], ent["n_rows"], ent["dim"] + 64) is None     # another geometry
    # a stale hash
    k0 = "bh_scan_topk256_kernel@768"
    stale = {"kernels": {k0: dict(table[k0], source_sha16="0" * 16)}}
    p = tmp_path / "t.json"
    p.write_text(json.dumps(stale))
    assert bench.pmc_traffic(str(p), "bh_scan_topk256_kernel", table[k0]["n_rows"], 768) is None
    assert bench.pmc_traffic(str(tmp_path / "missing.json"), "bh_scan_topk256_kernel", 1, 768) is None
