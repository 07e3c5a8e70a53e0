"""cyberether_amd -- MI355X (gfx950) compute backend for CyberEther's Jetstream DSP module graph.

The product is the native library ``cyberether_amd/lib/libjetstream_hip.so`` (hand-written HIP
kernels + the C++ module/runtime layer, C ABI in ``include/jetstream_hip.h``).  This package is
the thin Python host mirror of the reference's Tensor / Module / Runtime / Block interface on top
of that ABI (``cyberether_amd.jetstream``).  There is no CPU or PyTorch fallback: importing
``cyberether_amd.jetstream`` raises if the library has not been built
(``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C cyberether_amd/csrc``).
"""

__version__ = "0.1.0"
