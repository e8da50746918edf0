"""Device assembly of csrc/api.hip for the scanners (store_hazard_scan.py, pinned_mfma_audit.py): compiled once into build/ and reused while
no source under csrc/ or include/ is newer than it (each compile is ~45 s; the CPU test suite runs three scans)."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assembly_path(variants: bool = False) -> str:
    out = os.path.join(ROOT, "build", "api_variants.s" if variants else "api_product.s")
    srcs = glob.glob(os.path.join(ROOT, "contrastiveprosthetics_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*")) + \
        glob.glob(os.path.join(ROOT, "tools", "variants", "*"))
    newest = max(os.path.getmtime(f) for f in srcs)
    if os.path.exists(out) and os.path.getmtime(out) > newest and os.path.getsize(out) > 0:
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    tmp = out + ".tmp"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-w"]
                   + (["-DCP_VARIANTS"] if variants else []) + [os.path.join(ROOT, "contrastiveprosthetics_amd", "csrc", "api.hip"), "-o", tmp],
                   check=True, cwd=os.path.dirname(out), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    os.replace(tmp, out)
    return out
