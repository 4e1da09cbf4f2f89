"""One cross-compilation of scan_topk256.hip (the slowest translation unit: ~80 s) shared by the CPU tests that read its
generated code (test_scan256_isa.py) and its resource remarks (test_build_resources.py): the assembly listing and hipcc's
-Rpass-analysis=kernel-resource-usage output, cached under the system temp directory by the sha256 of the source and headers."""
import glob
import hashlib
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bergen_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def scan256_asm_and_remarks():
    """-> (path of the device assembly of scan_topk256.hip, hipcc's stderr with the resource-usage remarks)."""
    src = os.path.join(CSRC, "scan_topk256.hip")
    h = hashlib.sha256()
    for f in [src] + sorted(glob.glob(os.path.join(CSRC, "*.h"))):
        h.update(open(f, "rb").read())
    base = os.path.join(tempfile.gettempdir(), f"bergen_amd_scan256_{h.hexdigest()[:16]}")
    asm, log = base + ".s", base + ".log"
    if not (os.path.exists(asm) and os.path.exists(log)):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", asm + ".tmp",
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd=CSRC)
        assert r.returncode == 0, r.stderr[-2000:]
        with open(log + ".tmp", "w") as f:
            f.write(r.stderr)
        os.replace(asm + ".tmp", asm)
        os.replace(log + ".tmp", log)
    return asm, open(log).read()
