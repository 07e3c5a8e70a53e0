"""sha256 over the sources the fused spectrum kernel is compiled from: the provenance stamp that ties
profiles/pmc_traffic.json (written by tools/pmc_summary.py from a --pmc pass) to the kernel bench.py is timing.
bench.py refuses to quote the file's HBM traffic when this hash differs from the one recorded in it."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ("fft_lds.hh", "fft_quad.hh", "fft_kernels.hip", "fft_side.hip", "fft_radix.hh", "device_math.hh", "libm_float.hh",
                  "kernels.hh", "spectrogram_body.hh")


def kernel_sources_sha256() -> str:
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        path = os.path.join(ROOT, "cyberether_amd", "csrc", "kernels", name)
        h.update(name.encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


if __name__ == "__main__":
    print(kernel_sources_sha256())
