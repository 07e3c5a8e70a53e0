"""CPU: INTEGRATION.md section 3 is real, LINKED code.  integration/device_hip/core_hip_device.patch (the closed switches of the
reference's core: DeviceType enum + name maps, MakeBackend, the Runtime factory, the CPU mirror of a host-accessible HIP buffer,
duplicate's HasBufferBackend) applies to the reference tree, and oracle/ref_jetstream_build.sh links the patched core with the
reference-side units beside it (buffer_hip.cc, runtime_native_hip_impl.cc, modules/*.cc) and libjetstream_hip.so into
oracle/_ref/libref_jetstream_devhip.so.  Here, without a GPU: the patch applies, the library exists, exports the new factories,
loads, knows the device name and holds every module of the spectrum chain under (DeviceType::HIP, NATIVE) -- the RUN is
tests/test_gpu_reference_device_hip.py.  Also: the integration/mi355x_provider/ units compile against the UNPATCHED reference.
Skipped where the reference tree is absent (the GPU box)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
DEV = os.path.join(ROOT, "integration", "device_hip")
TOUCHED = ["include/jetstream/memory/types.hh", "src/memory/types.cc", "src/memory/buffer_backend.hh", "src/memory/buffer.cc",
           "src/memory/buffer_cpu.cc", "src/runtime/runtime.cc", "src/domains/core/duplicate/module_impl.cc"]
DEVHIP = os.path.join(ROOT, "oracle", "_ref", "libref_jetstream_devhip.so")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "include", "jetstream", "registry.hh")),
                                reason="reference tree not present")


def torch_include():
    import torch
    return os.path.join(os.path.dirname(torch.__file__), "include")


@pytest.fixture(scope="module")
def patched(tmp_path_factory):
    tree = str(tmp_path_factory.mktemp("ref_hip"))
    for rel in TOUCHED:
        os.makedirs(os.path.dirname(os.path.join(tree, rel)), exist_ok=True)
        shutil.copy(os.path.join(REF, rel), os.path.join(tree, rel))
    r = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "-i", os.path.join(DEV, "core_hip_device.patch")],
                       cwd=tree, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return tree


def test_the_patch_adds_the_device(patched):
    text = open(os.path.join(patched, "include/jetstream/memory/types.hh")).read()
    assert "HIP     = 1 << 6" in text            # the value include/jetstream_hip.h calls JST_DEVICE_HIP
    header = open(os.path.join(ROOT, "include", "jetstream_hip.h")).read()
    assert "JST_DEVICE_HIP = 1 << 6" in header
    assert "CreateHipBackend" in open(os.path.join(patched, "src/memory/buffer.cc")).read()
    assert "NativeHipRuntimeFactory" in open(os.path.join(patched, "src/runtime/runtime.cc")).read()
    assert "Mirroring HIP buffer" in open(os.path.join(patched, "src/memory/buffer_cpu.cc")).read()
    assert "case DeviceType::HIP" in open(os.path.join(patched, "src/domains/core/duplicate/module_impl.cc")).read()


@pytest.fixture(scope="module")
def devhip():
    if not os.path.exists(os.path.join(ROOT, "cyberether_amd", "lib", "libjetstream_hip.so")):
        import __graft_entry__
        __graft_entry__.build()
    if not os.path.exists(DEVHIP):
        subprocess.check_call([os.path.join(ROOT, "oracle", "ref_jetstream_build.sh"), "-j", "8"], stdout=subprocess.DEVNULL)
    assert os.path.exists(DEVHIP), "oracle/ref_jetstream_build.sh did not link the DeviceType::HIP build of the reference"
    return DEVHIP


def test_the_patched_reference_links_with_the_hip_units(devhip):
    syms = subprocess.run(["nm", "-DC", "--defined-only", devhip], capture_output=True, text=True).stdout
    for want in ("Jetstream::detail::CreateHipBackend()", "Jetstream::NativeHipRuntimeFactory()", "jetstream_hip_runtime_configure",
                 "jetstream_hip_runtime_flush", "jetstream_hip_runtime_units"):
        assert want in syms, f"{want} not defined in libref_jetstream_devhip.so"
    needed = subprocess.run(["readelf", "-d", devhip], capture_output=True, text=True).stdout
    assert "libjetstream_hip.so" in needed and "libamdhip64" in needed


def test_the_registry_of_the_patched_reference_holds_the_hip_modules(devhip):
    """A separate process: one build of the reference per process (the CPU suite's checker is libref_jetstream.so)."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from oracle import ref_jetstream as rj\n"
            "rj.use_device_hip_library()\n"
            "l = rj.lib()\n"
            "assert l.ref_device_known(b'hip') == 1\n"
            "for m in ('cast','window','invert','reshape','multiply','fft','amplitude','range','spectrogram','ring_source'):\n"
            "    assert rj.registry_has(m, 'generic', device='hip'), m\n"
            "assert rj.registry_has('amplitude', 'fast', device='hip') and rj.registry_has('fft', 'mi355x')\n"
            "assert not rj.registry_has('fft', 'generic', device='cuda')\n"
            "print('ok')\n" % ROOT)
    r = subprocess.run(["python3", "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


PROVIDER = {"fft": "dsp/fft", "amplitude": "dsp/amplitude", "range": "core/range", "multiply": "core/multiply", "invert": "dsp/invert",
            "window": "dsp/window", "reshape": "core/reshape", "cast": "core/cast", "spectrogram": "visualization/spectrogram"}


@pytest.mark.parametrize("unit", sorted(PROVIDER))
def test_provider_units_compile_against_the_unpatched_reference(tmp_path, unit):
    cmd = ["g++", "-std=c++20", "-fsyntax-only", "-w", "-DFMT_HEADER_ONLY=1", "-I" + os.path.join(ROOT, "oracle", "ref_shim"),
           "-I" + os.path.join(REF, "include"), "-I" + os.path.join(REF, "src"), "-I" + os.path.join(REF, "src/domains", PROVIDER[unit]),
           "-I" + torch_include(), "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "integration", "mi355x_provider"),
           os.path.join(ROOT, "integration", "mi355x_provider", unit + ".cc")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
