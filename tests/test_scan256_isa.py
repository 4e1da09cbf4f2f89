"""CPU: instruction-level checks of the 256-query scan kernel's generated code (cross-compiled for gfx950, no GPU needed).

scan_topk256.hip issues its LDS fragment reads as inline asm and tracks their completion by hand (`s_waitcnt lgkmcnt(n)`);
the compiler does not know that the destination registers are still in flight.  These tests read the ISA hipcc produced
and check what the source can only promise:
  * no instruction touches the destination of a fragment read before the wait that covers it (a register copy or spill
    inserted by the register allocator in that window would move stale bytes);
  * the tile loop of every production instantiation is free of scratch traffic (a reload in there also drains the whole
    LDS-DMA pipeline through vmcnt) and holds exactly the expected number of MFMAs;
  * the wait states between the last MFMAs and the first VALU read of their results are still in front of that read."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "bergen_amd", "csrc", "scan_topk256.hip")
HIPCC = "/opt/rocm/bin/hipcc"

RESOURCE_LOG = ""

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


@pytest.fixture(scope="module")
def kernels():
    from hipcc_cache import scan256_asm_and_remarks  # (one compilation shared with test_build_resources.py)
    out, remarks = scan256_asm_and_remarks()
    global RESOURCE_LOG
    RESOURCE_LOG = remarks
    res, name, body = {}, None, []
    for line in open(out):
        m = re.match(r"^(_Z22bh_scan_topk256_kernel\S+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(line.rstrip("\n"))
            if line.startswith(".Lfunc_end"):  # (not the first s_endpgm: a kernel may have several exits)
                res[name] = body
                name = None
    assert res
    return res


def _regs(txt):
    u = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", txt):
        u.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", txt):
        u.add(int(m.group(1)))
    return u


def _code(lines):
    for i, l in enumerate(lines):
        c = l.split(";")[0].strip()
        if c and not c.startswith(".") and not c.endswith(":"):
            yield i, c


def production(name):
    # template arguments: NK32, KP, LS, R, PD, NT, ABL, LM, SCHED, NBUF, NB; production = ABL 0, LM 1, SCHED 1, NBUF = PD
    m = re.search(r"kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name)
    assert m, name
    nk32, kp, ls, r, pd, nt, abl, lm, sched, nbuf, nb = (int(x) for x in m.groups())
    # (LM 1 at d = 768 — NK32 24 — is one of the bench-only schedule variants of the headline geometry, option ring_variant 1: the
    # production kernel of that geometry is LM 0, checked by test_headline_kernel_uses_the_whole_register_file_without_scratch and below)
    return abl == 0 and sched == 1 and nbuf == pd and (lm == 1 if nk32 != 24 else lm == 0), nk32, kp, ls, nb


def test_no_instruction_touches_a_fragment_register_in_flight(kernels):
    for name, lines in kernels.items():
        pending = []  # (line, destination registers) of LDS operations not yet covered by a wait, oldest first
        for i, c in _code(lines):
            m = re.match(r"s_waitcnt.*lgkmcnt\((\d+)\)", c)
            if m:
                del pending[:max(0, len(pending) - int(m.group(1)))]
                continue
            if c.startswith("s_waitcnt"):
                continue
            if c.startswith("ds_read_b128"):
                pending.append((i, _regs(c.split(",")[0])))
                continue
            if c.startswith("ds_"):
                pending.append((i, set()))
                continue
            used = _regs(c)
            for j, dest in pending:
                assert not (used & dest), f"{name}: line {i + 1} `{c}` touches the destination of the LDS read at line {j + 1}"


def test_tile_loop_of_the_production_kernels(kernels):
    seen = 0
    for name, lines in kernels.items():
        prod, nk32, kp, ls, nb = production(name)
        if not prod:
            continue
        seen += 1
        # the SCANNING tile loop: the depth-1 loop in front of the filter's wait states (since round 5 the kernel also has an idle
        # tile loop — waves none of whose queries exists: rendezvous and refill only — and the cold paths hold loops of their own)
        end = next(i for i in range(len(lines)) if re.search(r"\bs_nop 15\b", lines[i]))
        start = max(i for i, l in enumerate(lines[:end]) if "Loop Header: Depth=1" in l)
        hot = [c for i, c in _code(lines[start:end])]
        assert not any("scratch_" in c for c in hot), f"{name}: scratch traffic inside the tile loop"
        n_mfma = sum(1 for c in hot if c.startswith("v_mfma_f32_16x16x32_f16"))
        assert n_mfma == 2 * nb * nk32, f"{name}: {n_mfma} MFMAs per tile, expected {2 * nb * nk32}"   # 2 row blocks x NB query blocks x k-steps
        n_read = sum(1 for c in hot if c.startswith("ds_read_b128"))
        assert n_read == 2 * nk32, f"{name}: {n_read} fragment reads per tile"
        # the wait states must precede the first VALU instruction that reads an accumulator
        tail = [c for i, c in _code(lines[end:end + 12])]
        assert tail[0].startswith("s_nop 15") and tail[1].startswith("s_nop 7"), f"{name}: {tail[:3]}"
        assert not any(c.startswith("v_") for c in hot[-2:] if not c.startswith("v_mfma")), name
    assert seen >= 24  # dims 384 / 512 / 768 / 1024 x candidate lists 64 / 128 / 256 x nt / default policy (d = 768: the LM 0 kernels)


def test_headline_kernel_uses_the_whole_register_file_without_scratch(kernels):
    cur, usage = None, {}
    for line in RESOURCE_LOG.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            usage[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            usage[cur][m.group(1).strip()] = int(m.group(2))
    head = [u for n, u in usage.items() if re.search(r"kernelILi24ELi64ELi12ELi3ELi4ELb[01]ELi0ELi0ELi1ELi4ELi2E", n)]
    assert len(head) == 2
    for u in head:
        assert u["ScratchSize"] == 0 and u["Occupancy"] == 2 and u["VGPRs"] <= 128 and u["AGPRs"] <= 128, u
