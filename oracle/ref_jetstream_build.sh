#!/bin/bash
# oracle/ref_jetstream_build.sh -- TEST INFRASTRUCTURE: compiles the REFERENCE's own core (Tensor / Module / Block /
# Flowgraph / Registry / native-CPU runtime / synchronous scheduler) and the native-CPU module and block
# implementations of the hot path IN PLACE from /root/reference into oracle/_ref/libref_jetstream.so (git-ignored;
# travels to the GPU box like the other built libraries).  Nothing is copied: every translation unit is compiled
# where it lies, with oracle/ref_shim first on the include path (a stub config.hh; jst::fmt forwarded to the fmt
# headers torch ships, header-only).  The reference's own build (meson + network wraps) is NOT run.  What is left out:
# render / compositor / viewport / remote instance / YAML parser (rapidyaml is a network wrap) / python runtime /
# non-CPU devices -- none of them is on the path; oracle/ref_jetstream.cc provides the three symbols the core still
# asks for (PythonRuntimeFactory and two Platform helpers) and the extern "C" harness the tests drive.
# Round 5: the visualization modules (spectrogram / waterfall / lineplot) are compiled in place too -- their COMPUTE halves
# are on the path (a10, a11, f2).  Their translation units also hold the present halves: oracle/ref_shim
# carries empty stand-ins for the shader tables the reference's build generates (resources/shaders/*_shaders.hh) and a
# few glm types; oracle/ref_render_stubs.cc the Render:: symbols those halves reference (aborting if ever reached).
#   usage: oracle/ref_jetstream_build.sh [-j N]   (REF=/root/reference by default)
set -u
HERE=$(cd "$(dirname "$0")" && pwd)
REF=${REF:-/root/reference}
JOBS=8
[ "${1:-}" = "-j" ] && JOBS=$2
if [ ! -f "$REF/src/module.cc" ]; then echo "reference tree absent: keeping prebuilt _ref/ (if any)"; exit 0; fi
TORCH_INC=$(python3 -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'include'))" 2>/dev/null)
if [ -z "$TORCH_INC" ] || [ ! -f "$TORCH_INC/fmt/format.h" ]; then echo "fmt headers not found: libref_jetstream.so not built"; exit 0; fi
OUT=$HERE/_ref
OBJ=$OUT/obj_jetstream
mkdir -p "$OBJ"
# -O3, no -march (meson_options.txt:1, meson.build:8); -ffp-contract=off is a no-op on baseline x86-64 (no FMA) and
# keeps it that way if someone adds -march flags.
CXXFLAGS="-O3 -std=c++20 -fPIC -ffp-contract=off -w -DFMT_HEADER_ONLY=1 -I$HERE/ref_shim -I$REF/include -I$REF/src -I$TORCH_INC"

CORE="logger module module_impl module_context module_interface module_surface block block_impl block_context
      block_interface registry parser_map parser_decode parser_encode tensor_link testing scheduler scheduler_context
      scheduler_synchronous flowgraph flowgraph_metadata flowgraph_environment flowgraph_view runtime/runtime runtime/native/cpu/context
      runtime/native/cpu/impl memory/tensor memory/buffer memory/buffer_cpu memory/axis memory/types memory/token
      platform/terminal platform/process platform/paths"
MODULES="core/add core/arithmetic core/cast core/duplicate core/expand_dims core/flatten core/multiply
         core/multiply_constant core/ones_tensor core/pad core/permutation core/range core/reshape core/signal_axes
         core/slice core/squeeze_dims core/unpad
         dsp/agc dsp/am dsp/amplitude dsp/fft dsp/filter_taps dsp/fm dsp/fold dsp/invert dsp/overlap_add
         dsp/phase_correction dsp/signal_generator dsp/squelch dsp/window
         visualization/spectrogram visualization/waterfall visualization/lineplot
"
BLOCKS="$MODULES dsp/filter dsp/filter_engine dsp/spectrum_engine dsp/decimator"

LIST=$OBJ/units.txt
: > "$LIST"
for c in $CORE; do echo "src/$c.cc" >> "$LIST"; done
for m in $MODULES; do
  for f in module_impl.cc module_impl_native_cpu.cc; do
    [ -f "$REF/src/domains/$m/$f" ] && echo "src/domains/$m/$f" >> "$LIST"
  done
done
for b in $BLOCKS; do [ -f "$REF/src/domains/$b/block_impl.cc" ] && echo "src/domains/$b/block_impl.cc" >> "$LIST"; done
[ -n "${EXTRA_UNITS:-}" ] && for u in $EXTRA_UNITS; do echo "$u" >> "$LIST"; done

compile_one() {
  u=$1
  o=$OBJ/$(echo "$u" | tr '/' '_' | sed 's/\.cc$//').o
  if [ ! -f "$o" ] || [ "$REF/$u" -nt "$o" ]; then
    g++ $CXXFLAGS -I"$(dirname "$REF/$u")" -c "$REF/$u" -o "$o" 2> "$o.err" || { echo "FAILED $u"; head -5 "$o.err"; rm -f "$o"; return 1; }
  fi
  rm -f "$o.err"
}
export -f compile_one; export OBJ REF CXXFLAGS
xargs -a "$LIST" -P "$JOBS" -I{} bash -c 'compile_one {}' || { echo "libref_jetstream.so: some units failed"; exit 1; }
g++ $CXXFLAGS -c "$HERE/ref_jetstream.cc" -o "$OBJ/harness.o" || exit 1
g++ $CXXFLAGS -c "$HERE/ref_render_stubs.cc" -o "$OBJ/render_stubs.o" || exit 1
OBJS=$(sed 's|/|_|g; s|\.cc$|.o|' "$LIST" | sed "s|^|$OBJ/|")
g++ -shared -fPIC -o "$OUT/libref_jetstream.so" $OBJS "$OBJ/harness.o" "$OBJ/render_stubs.o" -Wl,--no-undefined -lpthread ${EXTRA_LINK:-} && echo "built _ref/libref_jetstream.so from $REF"

# ---- the reference DRIVING the library (round 5) --------------------------------------------------------------------------
# A second library, oracle/_ref/libref_jetstream_hip.so: the same reference objects plus the `provider: mi355x` modules of
# integration/mi355x_provider/ (reference-side code: the reference's own Impl classes with computeSubmit() forwarded to
# libjetstream_hip.so), linked against cyberether_amd/lib/libjetstream_hip.so.  The reference's registry, Module::create,
# runtime, scheduler and the spectrum_engine BLOCK then run the HIP kernels (tests/test_gpu_reference_drives_library.py).
# Kept apart from libref_jetstream.so so that the CPU checker above never depends on the product.
REPO=$(cd "$HERE/.." && pwd)
HIPLIB=$REPO/cyberether_amd/lib/libjetstream_hip.so
if [ -f "$HIPLIB" ]; then
  SHIM=$REPO/integration/mi355x_provider
  SHIM_OBJS=""
  for pair in fft:dsp/fft amplitude:dsp/amplitude range:core/range multiply:core/multiply invert:dsp/invert window:dsp/window \
              reshape:core/reshape cast:core/cast spectrogram:visualization/spectrogram; do
    f=${pair%%:*}; d=${pair##*:}
    o=$OBJ/mi355x_$f.o
    if [ ! -f "$o" ] || [ "$SHIM/$f.cc" -nt "$o" ] || [ "$SHIM/mi355x_bridge.hh" -nt "$o" ]; then
      g++ $CXXFLAGS -I"$REF/src/domains/$d" -I"$SHIM" -I"$REPO/include" -c "$SHIM/$f.cc" -o "$o" || { echo "FAILED integration/mi355x_provider/$f.cc"; exit 1; }
    fi
    SHIM_OBJS="$SHIM_OBJS $o"
  done
  g++ -shared -fPIC -o "$OUT/libref_jetstream_hip.so" $OBJS "$OBJ/harness.o" "$OBJ/render_stubs.o" $SHIM_OBJS -L"$REPO/cyberether_amd/lib" -l:libjetstream_hip.so \
      -Wl,--no-undefined -Wl,-rpath,'$ORIGIN/../../cyberether_amd/lib' -lpthread && echo "built _ref/libref_jetstream_hip.so (reference + mi355x provider -> libjetstream_hip.so)"
fi

# ---- the reference DEVICE-RESIDENT on the library: DeviceType::HIP (round 6) ---------------------------------------------
# A third library, oracle/_ref/libref_jetstream_devhip.so: the reference's core compiled with integration/device_hip/
# core_hip_device.patch (the DeviceType bit, MakeBackend, the Runtime factory, the CPU mirror of a host-accessible HIP buffer,
# duplicate's HasBufferBackend) + the reference-side units beside the patch: buffer_hip.cc (HBM tensor store),
# runtime_native_hip_impl.cc (the segment runtime: module by module, or handed to ONE jst_runtime with graph capture and
# fusion) and modules/*.cc (one module_impl_native_hip.cc per module of the spectrum chain: the reference's own Impl on device
# tensors, its compute hooks on the library module).  The patch is applied to a TEMPORARY copy of the seven files it touches
# (mktemp, removed afterwards: no reference source enters the repository or travels to the GPU box); every translation unit
# is recompiled with -DJETSTREAM_BACKEND_HIP_AVAILABLE and the patched headers first on the include path, so the whole
# library agrees on the enum; everything else is compiled where it lies under /root/reference, as above.
# tests/test_gpu_reference_device_hip.py runs the reference's Flowgraph / scheduler / Runtime on it: tensors stay in HBM
# between modules, outputs and Spectrogram bins bit-equal to the reference's CPU modules.
DEV=$REPO/integration/device_hip
if [ -f "$HIPLIB" ] && [ -f "$DEV/core_hip_device.patch" ] && [ -f /opt/rocm/include/hip/hip_runtime_api.h ]; then
  TREE=$(mktemp -d /tmp/ref_hip_tree.XXXXXX)
  trap 'rm -rf "$TREE"' EXIT
  TOUCHED="include/jetstream/memory/types.hh src/memory/types.cc src/memory/buffer_backend.hh src/memory/buffer.cc src/memory/buffer_cpu.cc
           src/runtime/runtime.cc src/domains/core/duplicate/module_impl.cc"
  for f in $TOUCHED; do mkdir -p "$TREE/$(dirname $f)"; cp "$REF/$f" "$TREE/$f"; done
  (cd "$TREE" && patch -s -p1 --no-backup-if-mismatch -i "$DEV/core_hip_device.patch") || { echo "core_hip_device.patch does not apply"; exit 1; }
  cp "$DEV/runtime_context_native_hip.hh" "$TREE/include/jetstream/"
  OBJD=$OUT/obj_jetstream_devhip
  mkdir -p "$OBJD"
  DEVFLAGS="-O3 -std=c++20 -fPIC -ffp-contract=off -w -DFMT_HEADER_ONLY=1 -DJETSTREAM_BACKEND_HIP_AVAILABLE -D__HIP_PLATFORM_AMD__
            -I$TREE/include -I$TREE/src -I$TREE/src/memory -I$HERE/ref_shim -I$REF/include -I$REF/src -I$TORCH_INC -I$REPO/include -I$DEV -I/opt/rocm/include"
  STAMP="$OBJD/.patch_stamp"
  # a changed patch (or header of the HIP glue) invalidates every object: the enum is in everybody's headers
  if [ ! -f "$STAMP" ] || [ "$DEV/core_hip_device.patch" -nt "$STAMP" ] || [ "$DEV/runtime_context_native_hip.hh" -nt "$STAMP" ]; then
    rm -f "$OBJD"/*.o; touch "$STAMP"
  fi
  compile_dev() {
    u=$1
    o=$OBJD/$(echo "$u" | tr '/' '_' | sed 's/\.cc$//').o
    src=$REF/$u
    [ -f "$TREE/$u" ] && src=$TREE/$u            # a patched unit
    if [ ! -f "$o" ] || [ "$REF/$u" -nt "$o" ]; then
      g++ $DEVFLAGS -I"$(dirname "$REF/$u")" -c "$src" -o "$o" 2> "$o.err" || { echo "FAILED (devhip) $u"; head -5 "$o.err"; rm -f "$o"; return 1; }
    fi
    rm -f "$o.err"
  }
  export -f compile_dev; export OBJD TREE DEVFLAGS
  xargs -a "$LIST" -P "$JOBS" -I{} bash -c 'compile_dev {}' || { echo "libref_jetstream_devhip.so: some units failed"; exit 1; }
  GLUE_OBJS=""
  for pair in buffer_hip:. runtime_native_hip_impl:. cast:core/cast window:dsp/window invert:dsp/invert reshape:core/reshape multiply:core/multiply \
              fft:dsp/fft amplitude:dsp/amplitude range:core/range spectrogram:visualization/spectrogram ring_source:. side_chains:. ${DEVHIP_EXTRA_MODULES:-}; do
    f=${pair%%:*}; d=${pair##*:}
    src=$DEV/modules/$f.cc; [ -f "$DEV/$f.cc" ] && src=$DEV/$f.cc
    o=$OBJD/glue_$f.o
    if [ ! -f "$o" ] || [ "$src" -nt "$o" ] || [ "$DEV/hip_library_module.hh" -nt "$o" ] || [ "$DEV/native_hip_module.hh" -nt "$o" ] || [ "$REPO/include/jetstream_hip.h" -nt "$o" ]; then
      g++ $DEVFLAGS -I"$REF/src/domains/$d" -c "$src" -o "$o" || { echo "FAILED integration/device_hip/$f.cc"; exit 1; }
    fi
    GLUE_OBJS="$GLUE_OBJS $o"
  done
  # the `provider: mi355x` modules of the second library too (CPU-device modules: nothing of theirs depends on the patch), so
  # that ONE reference build serves a whole pytest process: host-staged provider AND device-resident HIP modules
  for pair in fft:dsp/fft amplitude:dsp/amplitude range:core/range multiply:core/multiply invert:dsp/invert window:dsp/window \
              reshape:core/reshape cast:core/cast spectrogram:visualization/spectrogram ${MI355X_EXTRA_MODULES:-}; do
    f=${pair%%:*}; d=${pair##*:}
    o=$OBJD/mi355x_$f.o
    if [ ! -f "$o" ] || [ "$SHIM/$f.cc" -nt "$o" ] || [ "$SHIM/mi355x_bridge.hh" -nt "$o" ]; then
      g++ $DEVFLAGS -I"$REF/src/domains/$d" -I"$SHIM" -c "$SHIM/$f.cc" -o "$o" || { echo "FAILED (devhip) integration/mi355x_provider/$f.cc"; exit 1; }
    fi
    GLUE_OBJS="$GLUE_OBJS $o"
  done
  if [ ! -f "$OBJD/harness.o" ] || [ "$HERE/ref_jetstream.cc" -nt "$OBJD/harness.o" ] || [ "$DEV/hip_library_module.hh" -nt "$OBJD/harness.o" ]; then
    g++ $DEVFLAGS -c "$HERE/ref_jetstream.cc" -o "$OBJD/harness.o" || exit 1
  fi
  g++ $DEVFLAGS -c "$HERE/ref_render_stubs.cc" -o "$OBJD/render_stubs.o" || exit 1
  DOBJS=$(sed 's|/|_|g; s|\.cc$|.o|' "$LIST" | sed "s|^|$OBJD/|")
  g++ -shared -fPIC -o "$OUT/libref_jetstream_devhip.so" $DOBJS "$OBJD/harness.o" "$OBJD/render_stubs.o" $GLUE_OBJS -L"$REPO/cyberether_amd/lib" -l:libjetstream_hip.so \
      -L/opt/rocm/lib -lamdhip64 -Wl,--no-undefined -Wl,-rpath,'$ORIGIN/../../cyberether_amd/lib' -Wl,-rpath,/opt/rocm/lib -lpthread \
    && echo "built _ref/libref_jetstream_devhip.so (reference core + core_hip_device.patch + HIP backend / runtime / modules -> libjetstream_hip.so)"
fi
