"""CPU: INTEGRATION.md section 3 is real code.  integration/device_hip/core_hip_device.patch (the closed switches of the
reference's core: DeviceType enum + name maps, MakeBackend, the Runtime factory) is applied with `patch -p1` to a TEMPORARY copy
of the five reference files it touches, and every touched unit plus the new reference-side units beside it (buffer_hip.cc,
runtime_native_hip_impl.cc, fft_module_impl_native_hip.cc, runtime_context_native_hip.hh) is compiled to an object against
the reference's real headers, the HIP runtime's headers and this repo's include/ with -DJETSTREAM_BACKEND_HIP_AVAILABLE.
Also: the integration/mi355x_provider/ units compile (they are linked and RUN by tests/test_gpu_reference_drives_library.py).
Skipped where the reference tree is absent (the GPU box)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
DEV = os.path.join(ROOT, "integration", "device_hip")
TOUCHED = ["include/jetstream/memory/types.hh", "src/memory/types.cc", "src/memory/buffer_backend.hh", "src/memory/buffer.cc",
           "src/runtime/runtime.cc"]

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "include", "jetstream", "registry.hh")),
                                reason="reference tree not present")


def torch_include():
    import torch
    return os.path.join(os.path.dirname(torch.__file__), "include")


def flags(tree):
    return ["g++", "-std=c++20", "-c", "-O0", "-w", "-fPIC", "-DFMT_HEADER_ONLY=1", "-DJETSTREAM_BACKEND_HIP_AVAILABLE",
            "-D__HIP_PLATFORM_AMD__",
            "-I" + os.path.join(tree, "include"), "-I" + os.path.join(tree, "src", "memory"),   # the PATCHED headers first
            "-I" + os.path.join(ROOT, "oracle", "ref_shim"), "-I" + os.path.join(REF, "include"), "-I" + os.path.join(REF, "src"),
            "-I" + torch_include(), "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include"]


@pytest.fixture(scope="module")
def patched(tmp_path_factory):
    tree = str(tmp_path_factory.mktemp("ref_hip"))
    for rel in TOUCHED:
        os.makedirs(os.path.dirname(os.path.join(tree, rel)), exist_ok=True)
        shutil.copy(os.path.join(REF, rel), os.path.join(tree, rel))
    r = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "-i", os.path.join(DEV, "core_hip_device.patch")],
                       cwd=tree, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    # the new header goes where the patch says it lives
    shutil.copy(os.path.join(DEV, "runtime_context_native_hip.hh"), os.path.join(tree, "include", "jetstream"))
    return tree


def test_the_patch_adds_the_device(patched):
    text = open(os.path.join(patched, "include/jetstream/memory/types.hh")).read()
    assert "HIP     = 1 << 6" in text            # the value include/jetstream_hip.h calls JST_DEVICE_HIP
    header = open(os.path.join(ROOT, "include", "jetstream_hip.h")).read()
    assert "JST_DEVICE_HIP = 1 << 6" in header
    assert "CreateHipBackend" in open(os.path.join(patched, "src/memory/buffer.cc")).read()
    assert "NativeHipRuntimeFactory" in open(os.path.join(patched, "src/runtime/runtime.cc")).read()


@pytest.mark.parametrize("unit", ["src/memory/types.cc", "src/memory/buffer.cc", "src/runtime/runtime.cc"])
def test_touched_units_compile(patched, tmp_path, unit):
    r = subprocess.run(flags(patched) + [os.path.join(patched, unit), "-o", str(tmp_path / "unit.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.parametrize("unit,extra", [("buffer_hip.cc", []), ("runtime_native_hip_impl.cc", []),
                                        ("fft_module_impl_native_hip.cc", ["-I" + os.path.join(REF, "src/domains/dsp/fft")])])
def test_new_units_compile(patched, tmp_path, unit, extra):
    obj = str(tmp_path / "unit.o")
    r = subprocess.run(flags(patched) + extra + [os.path.join(DEV, unit), "-o", obj], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    syms = subprocess.run(["nm", "-C", obj], capture_output=True, text=True).stdout
    want = {"buffer_hip.cc": "Jetstream::detail::CreateHipBackend()", "runtime_native_hip_impl.cc": "Jetstream::NativeHipRuntimeFactory()",
            "fft_module_impl_native_hip.cc": "FftImplNativeHip"}[unit]
    assert want in syms, f"{unit}: {want} not defined (is the unit compiled out?)"


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
