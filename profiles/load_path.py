"""The index load path (SURVEY H-4): a reference-layout folder of embedding_chunk_*.pt files -> resident HBM index, timed end
to end through the same code Retrieve.retrieve runs (utils.load_chunk on a prefetch thread -> FlatIndex.upload through two
pinned staging buffers -> finalize), with its parts timed on their own.  Page cache warm (the files were just written).
Run on the GPU box:  python profiles/load_path.py > profiles/r02_load_path.json
                     python profiles/load_path.py --full > profiles/r03_load_path_full.json   (32 GB folder, cold + warm)"""
import json
import os
import shutil
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bergen_amd  # noqa: E402
from bergen_amd import _lib, utils  # noqa: E402


def build_folder(path, n_rows, dim, dtype, chunk_rows=150_000):
    shutil.rmtree(path, ignore_errors=True)
    os.makedirs(path)
    g = torch.Generator().manual_seed(5)
    block = torch.nn.functional.normalize(torch.randn(chunk_rows, dim, generator=g), dim=1).to(dtype)
    done, i = 0, 0
    while done < n_rows:
        m = min(chunk_rows, n_rows - done)
        i += m // 128 + 1
        torch.save(block[:m].clone(), os.path.join(path, f"embedding_chunk_{i}.pt"))  # named after its last batch, like the reference
        done += m
    return utils.sorted_chunk_files(path)


def timed_load(files, n_rows, dim, prefetch, mmap):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ix = bergen_amd.FlatIndex(n_rows, dim, metric="ip", device=0)
    t_alloc = time.perf_counter() - t0
    row = 0
    t_load = t_up = 0.0
    src = utils.prefetched(files, lambda f: utils.load_chunk(f, mmap=mmap), depth=1) if prefetch else (utils.load_chunk(f, mmap=mmap) for f in files)
    ta = time.perf_counter()
    for emb in src:
        tb = time.perf_counter()
        t_load += tb - ta
        ix.upload(emb, row0=row)
        row += emb.shape[0]
        ta = time.perf_counter()
        t_up += ta - tb
    ix.finalize()
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    ix.close()
    return {"seconds": total, "index_alloc_s": t_alloc, "waiting_for_chunk_s": t_load, "upload_calls_s": t_up}


def evict_page_cache(files):
    """Drop the files' pages from the page cache: the global knob where the container allows it, else per file
    (fsync + POSIX_FADV_DONTNEED works for the owner of the files).  Returns what was done."""
    os.sync()
    try:
        with open("/proc/sys/vm/drop_caches", "w") as f:
            f.write("3\n")
        return "drop_caches"
    except OSError:
        pass
    for path in files:
        fd = os.open(path, os.O_RDONLY)
        try:
            os.fsync(fd)
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
        finally:
            os.close(fd)
    return "posix_fadvise(DONTNEED) per file"


def full_size_cold(tmp, n_rows=21_000_000, dim=768):
    """The headline corpus as a reference-layout folder (141 chunk files, 32 GB), read with a COLD page cache through the
    stage's own load path (Retrieve._resident_index: mapped files on a prefetch thread -> pinned staging -> HBM), then again
    warm.  The first-touch figure is what a user sees after a reboot or on a box with less RAM than the folder."""
    files = build_folder(tmp, n_rows, dim, torch.float16)
    file_bytes = sum(os.path.getsize(f) for f in files)
    how = evict_page_cache(files)

    class _Plug:
        model_name = "bench/precomputed"
        similarity = bergen_amd.DotProduct()
        model = torch.nn.Identity()

    out = {"rows": n_rows, "dim": dim, "files": len(files), "file_bytes": file_bytes, "page_cache_eviction": how,
           "folder_on": os.popen(f"df -T {tmp} | tail -1").read().split()[:2]}
    for name in ("cold", "warm"):
        stage = bergen_amd.Retrieve(init_args=_Plug(), batch_size=512, num_workers=0, device=0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stage._resident_index(tmp, n_rows, "ip")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[name] = {"seconds": dt, "file_GB_per_s": file_bytes / dt / 1e9}
        stage.close()
        print(name, out[name], file=sys.stderr, flush=True)
    return out


def main():
    if "--full" in sys.argv:
        _lib.init(0)
        tmp = sys.argv[sys.argv.index("--dir") + 1] if "--dir" in sys.argv else "/tmp/bergen_load_path_full"
        try:
            print(json.dumps({"what": "reference-layout folder of the headline corpus -> resident HBM index, cold and warm page cache",
                              "full_size": full_size_cold(tmp)}, indent=1))
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        return
    _lib.init(0)
    out = {"what": "reference-layout chunk folder -> resident HBM index (page cache warm)", "cases": []}
    tmp = "/tmp/bergen_load_path"
    for n_rows, dim, dtype in ((2_100_000, 768, torch.float16), (1_050_000, 768, torch.float32)):
        files = build_folder(tmp, n_rows, dim, dtype)
        file_bytes = sum(os.path.getsize(f) for f in files)
        case = {"rows": n_rows, "dim": dim, "dtype": str(dtype).replace("torch.", ""), "files": len(files), "file_bytes": file_bytes}
        timed_load(files[:2], 300_000, dim, True, True)  # warm-up: library, pinned buffers
        for name, prefetch, mmap in (("read_then_upload", False, False), ("mmap_then_upload", False, True), ("prefetch_thread_mmap", True, True)):
            r = timed_load(files, n_rows, dim, prefetch, mmap)
            r["file_GB_per_s"] = file_bytes / r["seconds"] / 1e9
            case[name] = r
        # the PCIe leg alone: one chunk already in pinned memory
        x = utils.load_chunk(files[0]).pin_memory()
        ix = bergen_amd.FlatIndex(x.shape[0], dim, metric="ip", device=0)
        ix.upload(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ix.upload(x, row0=0)
        torch.cuda.synchronize()
        case["upload_of_one_chunk_GB_per_s"] = 5 * x.numel() * x.element_size() / (time.perf_counter() - t0) / 1e9
        ix.close()
        out["cases"].append(case)
        print(case, file=sys.stderr, flush=True)
    shutil.rmtree(tmp, ignore_errors=True)
    # extrapolation to the headline corpus
    c = out["cases"][0]
    out["kilt_100w_21M_rows_estimate_s"] = c["prefetch_thread_mmap"]["seconds"] * 21_000_000 / c["rows"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
