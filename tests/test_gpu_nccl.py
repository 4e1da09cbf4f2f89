"""GPU: the RCCL transport itself, on the ONE GPU a test box has.  RCCL accepts a one-rank group per device, so the `nccl` backend is
initialised with world size 1 and the sharded search runs its whole collective path — `all_gather_into_tensor` of the packed
(scores | ids | status) byte buffer, the HIP merge of the gathered lists, the `broadcast` of the merged lists — on DEVICE buffers,
the way every rank of the 8-GPU run does (bergen_amd/sharded.py; reference decomposition: modules/retrieve.py:152-177).  What the
gloo tests cannot see and this does: `device_id=` initialisation, dmabuf IPC set-up, alignment of the uint8 views RCCL is handed,
stream order between the search's stream, torch's current stream and RCCL's.  What only the 8-GPU run adds: more than one peer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def nccl_world_1():
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("n,d,nq,k,metric", [(40_001, 768, 300, 50, "ip"), (9_000, 1024, 70, 200, "cos"), (20_000, 768, 2837, 50, "ip")])
def test_sharded_search_through_rccl_world_size_one(nccl_world_1, n, d, nq, k, metric):
    from bergen_amd import FlatIndex
    from bergen_amd.sharded import ShardedSearcher
    from oracle import c_oracle
    from oracle.compare import assert_bit_exact
    rng = np.random.default_rng(n + k)
    x = rng.standard_normal((n, d)).astype(np.float16)
    q = rng.standard_normal((nq, d)).astype(np.float16)
    x[n // 2] = x[7]
    ix = FlatIndex(n, d, metric=metric, device=0)
    ix.upload(x)
    ix.finalize()
    row_lo = 1_000_000  # global row ids of a shard that does not start at 0
    s = ShardedSearcher(ix, row_lo, exercise_collective=True)
    assert s.world_size == 1 and not s._host_collective
    qd = torch.from_numpy(q).cuda()
    for broadcast in (False, True):
        for rep in range(2):  # second round: the buffers are reused
            got_s, got_i = s.search(qd, k, broadcast=broadcast)
            assert got_s.is_cuda and got_i.is_cuda
            xs, qs = (c_oracle.l2_normalize_rows(x), c_oracle.l2_normalize_rows(q)) if metric == "cos" else (x, q)
            want_s, want_i = c_oracle.canonical_search(qs, xs, k, id_offset=row_lo)
            assert_bit_exact(got_s.cpu().numpy(), got_i.cpu().numpy(), want_s, want_i, f"rccl world 1, broadcast={broadcast}, round {rep}")
    # asynchronous form: no status read inside search(); the check rides on the copy to the host
    got = s.search(qd, k, broadcast=True, check=False)
    s.check_last()
    assert torch.equal(got[1].cpu(), torch.from_numpy(want_i))
    # host queries: the collective's buffers still live on the device (RCCL moves device memory only)
    s2 = ShardedSearcher(ix, row_lo, exercise_collective=True, device="cuda:0")
    hs, hi = s2.search(q, k, broadcast=True)
    assert hs.is_cuda
    assert_bit_exact(hs.cpu().numpy(), hi.cpu().numpy(), want_s, want_i, "host queries")
    ix.close()


def test_stage_search_rows_under_rccl(nccl_world_1, tmp_path):
    """The stage's own multi-rank code path (Retrieve.search_rows with search_world > 1 is what bench.py --gpus N steps through) needs
    world > 1 to engage; at world 1 under an initialised nccl group `search_world="auto"` must resolve to the single-rank path and
    still answer — the N = 1 line of a torchrun launch."""
    import bergen_amd
    from oracle import c_oracle
    from oracle.compare import assert_bit_exact
    from tests.test_retrieve_sharded_gloo import _Plug
    rng = np.random.default_rng(3)
    n, d, nq, k = 30_000, 768, 100, 50
    x = rng.standard_normal((n, d)).astype(np.float16)
    q = rng.standard_normal((nq, d)).astype(np.float16)
    stage = bergen_amd.Retrieve(init_args=_Plug("ip"), batch_size=64, num_workers=0, search_world="auto", device=0)
    assert stage.search_world == 1 and stage.search_rank == 0
    ix = bergen_amd.FlatIndex(n, d, metric="ip", device=0)
    ix.upload(x)
    ix.finalize()
    stage.adopt_resident_index("mem://docs", ix, n, "ip")
    s, i = stage.search_rows(torch.from_numpy(q).cuda(), "mem://docs", k, "ip", n)
    want_s, want_i = c_oracle.canonical_search(q, x, k)
    assert_bit_exact(s.numpy(), i.numpy(), want_s, want_i, "stage under an initialised nccl group")
    stage.close()
