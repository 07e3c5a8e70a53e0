"""examples/flowgraphs/multi-fm.yml (tests/golden/reference_flowgraphs/multi-fm.yml): 8 x 8000 CF32 -> Filter (51 taps, 2 heads, convolution
length 8050 = 2*5*5*7*23) -> spectrum engines + FM.  Times whole cycles (HIP graph replay) and prints the unit list, so the
same command under `rocprofv3 --kernel-trace --stats` shows the Filter's kernels.  JST_TILED_GENERIC=0 puts the 8050- and
805-point transforms back on the one-launch-per-pass kernels (fft_global.hip) for the comparison."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cyberether_amd import jetstream as js            # noqa: E402
from cyberether_amd.flowgraph import Flowgraph        # noqa: E402


def main():
    cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    fixture = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "reference_flowgraphs", "multi-fm.yml")
    fg = Flowgraph(fixture, ring_slots=1)
    rng = np.random.default_rng(31)
    x = (rng.standard_normal((8, 8000)) + 1j * rng.standard_normal((8, 8000))).astype(np.complex64)
    fg.feed("soapy", x)
    rt = fg.runtime(graph=True, fuse=True)
    rt.compute(20)
    t0 = time.perf_counter()
    rt.compute(cycles)
    dt = time.perf_counter() - t0
    print(json.dumps({"flowgraph": "multi-fm.yml", "cycles": cycles, "us_per_cycle": 1e6 * dt / cycles,
                      "fft_path_8050": js.fft_path(8050), "fft_path_805": js.fft_path(805),
                      "generic_radix_tiles": os.environ.get("JST_TILED_GENERIC", "1") != "0", "branches": rt.branches, "units": list(rt.units)}))
    rt.destroy()


if __name__ == "__main__":
    main()
