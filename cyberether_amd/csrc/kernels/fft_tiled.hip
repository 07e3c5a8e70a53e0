// fft_tiled.hip -- LDS-tiled mixed-radix FFT for lengths the register kernels (fft_lds.hh) do not
// cover: any n whose cfftp plan (pocketfft.hh:1476-1497) uses radices <= 11 or generic odd primes up to
// kMaxGenericRadix (passg, pocketfft.hh:1314-1421: tile_pass_generic below), in ONE kernel when a
// transform fits an LDS tile, otherwise in TWO kernels (instead of one launch per pass through HBM,
// fft_global.hip):
//
//   plan factors f0..f(nf-1), split at g:  R1 = f0*..*f(g-1),  S = n / R1
//   kernel A ("columns"): passes 0..g-1.  For these passes the low part of the index,
//       i mod S, never changes (ido_p is a multiple of S), so the n-point array is S independent
//       columns of R1 elements at stride S; a workgroup keeps CA adjacent columns in LDS and runs
//       the g passes on them with local ido' = ido_p / S and twiddle index i = column + S*i'.
//   kernel B ("blocks"): passes g..nf-1.  After pass g-1 everything derived from the contiguous
//       block k = [k*S, (k+1)*S) stays together (Stockham: k'' = k + l1*c), so a workgroup keeps CB
//       adjacent blocks in LDS, runs the remaining passes with local l1' = l1 / R1, and writes
//       result q of block k to its autosorted position k + R1*q (CB adjacent k -> contiguous).
//   g = 0 (single kernel): R1 = 1, a "block" is a whole transform read from the strided input.
//
// Same butterflies, same twiddle table W[k] = exp(2 pi j k / n) and the same `i == 0 skips the
// twiddle` rule as every other FFT path here, so results stay bit-identical to pocketfft.  The
// window multiply (prologue) and amplitude / range (epilogue) functors of fft_lds.hh fuse in, which
// gives BASELINE config 5 (65536 points) the same one-pass-over-HBM shape as config 2:
// 8 B read + 8+8 B scratch + 4 B write per sample instead of 6 passes x 16 B + 3 elementwise passes.
#include "fft_lds.hh"
#include "fft_radix.hh"
#include "kernels.hh"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace jst::kernels {

using namespace jst::dev;

namespace {

#ifdef JST_TILED_TIMELINE  // tools/ubench/tiled_timeline.hip: wall-clock stamps of workgroup phases
__device__ unsigned long long* jst_tiled_tl = nullptr;
// stamps go to LDS and reach memory in one piece at the end: a global store per stamp would sit in front of the next
// __syncthreads (which waits for vmcnt(0)) and stretch every phase by a store round trip
__shared__ unsigned long long jst_tiled_stamps[16];
#define JST_TSTAMP(i) do { if (threadIdx.x == 0) jst_tiled_stamps[(i)] = wall_clock64(); } while (0)
#define JST_TSTAMP_FLUSH() do { if (threadIdx.x == 0) { jst_tiled_stamps[15] = wall_clock64(); \
    for (int q_ = 0; q_ < 16; ++q_) jst_tiled_tl[(size_t)blockIdx.x * 16 + q_] = jst_tiled_stamps[q_]; } } while (0)
#else
#define JST_TSTAMP(i) do {} while (0)
#define JST_TSTAMP_FLUSH() do {} while (0)
#endif

#ifndef JST_TILED_MIN_WAVES
#define JST_TILED_MIN_WAVES 6  // wavefronts per SIMD the register budget must allow (80 VGPRs)
#endif
constexpr int kMaxThreads = 1024;  // workgroup size follows the tile: about 4 elements per thread
constexpr uint32_t kTileElems = 8192;  // upper bound of a tile (64 KiB of LDS, one buffer: passes run in place)
constexpr uint64_t kWantGroups = 1024; // enough workgroups to cover 256 CUs several times
constexpr uint32_t kMaxGenericRadix = 127;  // largest prime passg runs on an LDS tile (its wal[] table lives in LDS)
constexpr bool is_generic_radix(uint32_t ip) { return ip > 11; }
// tile_pass_generic: a wave owns at most kGenericSlots tasks = (pair of l, chunk of 64 butterflies x lanes)
constexpr int kGenericSlots = 4;
constexpr uint32_t generic_pass_tasks(uint32_t ip, uint64_t butterflies_x_lanes) {
    return (uint32_t)((((ip + 1u) / 2u) / 2u) * ((butterflies_x_lanes + 63u) / 64u));
}
constexpr uint64_t generic_pass_threads(uint32_t ip, uint64_t tile_elems) {  // workgroup size the pass needs on this tile
    return 64ull * ((generic_pass_tasks(ip, tile_elems / ip) + kGenericSlots - 1) / kGenericSlots);
}
// entries of pass (ip, global ido) in the per-pass twiddle table: the output twiddles, then -- generic radix only --
// passg's wal[0..ip-1] = W[m * n/ip] (comp_twiddle's csarr, pocketfft.hh:1526-1531)
constexpr uint64_t pass_table_entries(uint64_t ip, uint64_t ido) { return (ip - 1) * ido + (is_generic_radix((uint32_t)ip) ? ip : 0); }

// Pad fused into the first load (core/pad/module_impl_native_cpu.cc:75-140): positions at or beyond
// `valid` along the transform axis read as zero, everything else comes from the unpadded tensor.
// A load of data that is read ONCE (the transform's input, the scratch image behind the columns kernel): `nt`, so that it does
// not displace what comes back -- config 3 moves 128 MB of input and 128 MB of scratch through a 256 MB Infinity Cache.
// JST_TILED_STREAM_LOADS=0: A/B switch (plain loads).
#ifndef JST_TILED_STREAM_LOADS
#define JST_TILED_STREAM_LOADS 1
#endif
__device__ __forceinline__ float2 stream_load(const float2* p) {
#if JST_TILED_STREAM_LOADS
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f v = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(p));
    return mk(v.x, v.y);
#else
    return *p;
#endif
}

struct LoadCF32Padded {
    const float2* in;
    uint32_t valid;
    static constexpr bool kHasOperand = false;
    __device__ __forceinline__ const void* row(int64_t base) const { return in + base; }
    __device__ __forceinline__ const void* operand_row() const { return in; }
    __device__ __forceinline__ float2 apply(float2 v, float2) const { return v; }
    template <bool CONTIG>
    __device__ __forceinline__ float2 load(int64_t base, int64_t axis_stride, int pos) const {
        // branch-free: a conditional load is a branch, and hipcc drains vmcnt at every such branch when eight of them
        // are unrolled back to back (one HBM round trip per element); load a clamped position, select afterwards
        const uint32_t p = (uint32_t)pos < valid ? (uint32_t)pos : (valid ? valid - 1u : 0u);
        const float2 v = stream_load(in + (base + (int64_t)p * axis_stride));
        return (uint32_t)pos < valid ? v : mk(0.0f, 0.0f);
    }
};

struct TiledPlan {
    uint32_t n, nf, g, R1, S;
    uint32_t CA, CB;        // columns / blocks per workgroup: powers of two (ragged last tile)
    uint32_t ca_shift, cb_shift;
    // kernel B's lanes as `CB / grp_w` groups of grp_w adjacent blocks, the groups grp_stride blocks apart (one group
    // of CB adjacent blocks unless a fold epilogue needs the aliases of a bin in one workgroup: plan_fold_groups)
    uint32_t grp_w, grp_stride, grp_shift;  // grp_w = 1 << grp_shift
    uint32_t fact[20];
    uint32_t magic[20];     // ceil(2^32 / local ido of pass p): exact quotients for x < 2^16
    uint32_t tw_off[20];    // start of pass p in the per-pass twiddle table (fft_pass_twiddle_*)
};

// Lanes (columns / blocks) per workgroup.  Occupancy decides: a tile of <= 4096 elements means
// <= 512 threads and 32 KiB of LDS, i.e. three workgroups per CU whose load / pass / store phases
// overlap; so take the widest of {32,16} lanes that stays within 4096 elements and still yields
// kWantGroups workgroups, and fall back to 8 lanes (64-byte runs) up to the 8192-element cap.
constexpr uint32_t pick_lanes(uint32_t len, uint64_t lanes_total) {
    for (uint32_t c = 32; c >= 16; c /= 2)
        if ((uint64_t)len * c <= 4096 && (lanes_total + c - 1) / c >= kWantGroups) return c;
    return (uint64_t)len * 8 <= kTileElems ? 8u : 0u;
}
constexpr uint32_t ilog2(uint32_t v) {
    uint32_t s = 0;
    while ((1u << s) < v) ++s;
    return s;
}
constexpr uint32_t magic_of(uint32_t d) { return d <= 1 ? 0u : (uint32_t)((0x100000000ull + d - 1) / d); }

// pocketfft's cfftp::factorize (pocketfft.hh:1476-1497), usable in constant expressions (= kernels::fft_plan_factors)
constexpr int plan_factors_ce(uint64_t n, uint32_t* fact) {
    int nf = 0;
    uint64_t len = n;
    if (len == 0) return -1;
    if (len == 1) return 0;
    while ((len & 7) == 0) { fact[nf++] = 8; len >>= 3; }
    while ((len & 3) == 0) { fact[nf++] = 4; len >>= 2; }
    if ((len & 1) == 0) {
        len >>= 1;
        fact[nf++] = 2;
        const uint32_t t = fact[0];
        fact[0] = fact[nf - 1];
        fact[nf - 1] = t;
    }
    for (uint64_t d = 3; d * d <= len; d += 2)
        while (len % d == 0) {
            if (nf >= 60) return -1;
            fact[nf++] = (uint32_t)d;
            len /= d;
        }
    if (len > 1) {
        if (nf >= 60 || len > 0xffffffffull) return -1;
        fact[nf++] = (uint32_t)len;
    }
    return nf;
}

// The plan as a constant expression: the generic launch path calls it at run time (with the A/B lane overrides), the
// per-plan specialisations evaluate it at compile time (static_plan below) and the kernels then see every radix,
// stride, shift and magic number as a literal.
constexpr bool build_tiled_plan(uint64_t n, uint64_t transforms, uint32_t force_ca, uint32_t force_cb, TiledPlan& p, bool,
                                bool force_two, uint32_t force_g = 0);
// split_small: a transform that WOULD fit one tile (n <= 8192) still takes the two-kernel form when there are too few
// transforms to fill the chip with one workgroup each -- 8 x 8000 points (the reference's multi-fm.yml) is 8 workgroups of
// one 8000-point transform each, 18.8 us; as column + block workgroups the same passes (same order, same bits) take two
// short launches.  Not for the fold epilogue (its lane groups are planned for the one-kernel form of such lengths).
constexpr bool build_tiled_plan(uint64_t n, uint64_t transforms, uint32_t force_ca, uint32_t force_cb, TiledPlan& p,
                                bool split_small = false) {
    if (split_small && n <= kTileElems && n >= 4096 && transforms != 0 && transforms <= 32) {
        TiledPlan two{};
        if (build_tiled_plan(n, transforms, force_ca, force_cb, two, false, true)) {
            p = two;
            return true;
        }
    }
    return build_tiled_plan(n, transforms, force_ca, force_cb, p, false, false);
}
constexpr bool build_tiled_plan(uint64_t n, uint64_t transforms, uint32_t force_ca, uint32_t force_cb, TiledPlan& p, bool,
                                bool force_two, uint32_t force_g) {
    if (n < 2 || n > (1ull << 26)) return false;
    uint32_t fact[64] = {};
    const int nf = plan_factors_ce(n, fact);
    if (nf <= 0 || nf > 20) return false;
    for (int i = 0; i < nf; ++i)
        if (fact[i] > kMaxGenericRadix) return false;
    bool any_generic = false;
    for (int i = 0; i < nf; ++i) any_generic = any_generic || is_generic_radix(fact[i]);
    p = TiledPlan{};
    p.n = (uint32_t)n;
    p.nf = (uint32_t)nf;
    for (int i = 0; i < nf; ++i) p.fact[i] = fact[i];
    if (transforms == 0) transforms = 1;
    if (n <= kTileElems && !force_two) {  // one kernel, whole transforms per workgroup
        p.g = 0;
        p.R1 = 1;
        p.S = (uint32_t)n;
        uint32_t cb = 1;  // transforms per workgroup: keep kWantGroups workgroups when possible
        while (cb < 32 && (uint64_t)p.S * cb * 2 <= kTileElems && transforms / (cb * 2) >= kWantGroups)
            cb *= 2;
        p.CB = cb;
    } else {
        // two kernels: the split that balances R1 and S, both within a tile of >= 8 lanes
        uint32_t best = 0;
        double best_score = 1e300;
        uint64_t r1 = 1;
        for (int g = 1; g < nf; ++g) {
            r1 *= fact[g - 1];
            const uint64_t s = n / r1;
            if (force_g) {  // the caller names the split (and the lanes): only the tile bound is checked here
                if ((uint32_t)g != force_g || r1 * (force_ca ? force_ca : 8) > kTileElems || s * (force_cb ? force_cb : 8) > kTileElems)
                    continue;
                best = (uint32_t)g;
                break;
            }
            if (r1 * 8 > kTileElems || s * 8 > kTileElems) continue;
            const double score = r1 > s ? (double)r1 / (double)s : (double)s / (double)r1;
            if (score < best_score) {
                best_score = score;
                best = (uint32_t)g;
            }
        }
        if (!best) return false;
        p.g = best;
        p.R1 = 1;
        for (uint32_t i = 0; i < best; ++i) p.R1 *= fact[i];
        p.S = p.n / p.R1;
        p.CA = pick_lanes(p.R1, transforms * p.S);
        p.CB = pick_lanes(p.S, transforms * p.R1);
        if (!p.CA || !p.CB) return false;
        if (force_cb) p.CB = force_cb;
        if (force_ca) p.CA = force_ca;
    }
    if (any_generic)  // every generic pass fits the slots of a workgroup of at most kMaxThreads threads
        for (uint32_t q = 0; q < p.nf; ++q)
            if (is_generic_radix(p.fact[q]) &&
                generic_pass_threads(p.fact[q], q < p.g ? (uint64_t)p.R1 * p.CA : (uint64_t)p.S * p.CB) > (uint64_t)kMaxThreads)
                return false;
    p.ca_shift = ilog2(p.CA ? p.CA : 1);
    p.cb_shift = ilog2(p.CB);
    p.grp_w = p.CB;
    p.grp_stride = p.R1;
    p.grp_shift = p.cb_shift;
    // local ido of every pass: passes < g run on R1-point columns, the rest on S-point blocks
    uint32_t m = p.R1;
    for (uint32_t q = 0; q < p.g; ++q) {
        m /= p.fact[q];
        p.magic[q] = magic_of(m);
    }
    m = p.S;
    for (uint32_t q = p.g; q < p.nf; ++q) {
        m /= p.fact[q];
        p.magic[q] = magic_of(m);
    }
    uint64_t off = 0, l1 = 1;
    for (uint32_t q = 0; q < p.nf; ++q) {
        const uint64_t ido = n / (l1 * p.fact[q]);
        p.tw_off[q] = (uint32_t)off;
        off += pass_table_entries(p.fact[q], ido);
        l1 *= p.fact[q];
    }
    return true;
}

// (closed experiments: plans with a generic radix run on the tiles -- multi-fm.yml's Filter FFTs 66.0 -> 36.4 us --, and
// short launches of lengths that fit a tile split into columns + blocks -- 131 -> 125 us per cycle; profiles/EXPERIMENTS.md)
inline bool generic_radix_tiles_enabled() { return true; }
inline bool small_split_enabled() { return true; }

// (the table of constant plans: see "per-plan specialisation" below)
struct StaticPlanKey { uint64_t n, transforms, fold; uint32_t force_ca, force_cb; };  // fold != 0: with the fold epilogue's lane groups
// 1-3: the plans as build_tiled_plan picks them; 4-9: the same transforms with other lane counts (A/B through
// JST_TILED_CA / JST_TILED_CB, which make the run-time plan match one of them)
constexpr StaticPlanKey kStaticPlans[] = {{0, 0, 0, 0, 0},
                                          {65536, 16, 0, 0, 0}, {160000, 100, 16000, 0, 0}, {16000, 100, 0, 0, 0},
                                          {65536, 16, 0, 16, 0}, {65536, 16, 0, 32, 0}, {65536, 16, 0, 0, 16},
                                          {160000, 100, 16000, 0, 4}, {160000, 100, 16000, 32, 0}, {160000, 100, 16000, 8, 0},
                                          // 10-12 (round 6): the power-of-two lengths between the register kernels' 16384 and 2^18 that split into
                                          // two tiles, both lane counts named (the plan is then the same for every transform count; make_tiled_plan
                                          // takes them from a few dozen column tiles on): spectrum chains of these lengths get the constant-plan
                                          // kernels -- LDS twiddles, the persistent forms -- instead of the run-time-plan ones: 256 x 32768 points
                                          // 81.5 -> 66-70 us, 64 x 131072 90.6 -> 71.5-75.8, 32 x 262144 98.1 -> 80.5 us per cycle, provider fast
                                          // (profiles/r06_experiments/c_tiled_persistent.log section 13)
                                          {32768, 1, 0, 32, 8}, {131072, 1, 0, 8, 16}, {262144, 1, 0, 8, 8}};
[[maybe_unused]] constexpr int kStaticPlanCount = 13;

bool make_tiled_plan(uint64_t n, uint64_t transforms, TiledPlan& p, bool allow_split = true) {
    constexpr uint32_t force_ca = 0, force_cb = 0;  // lanes per workgroup of the two kernels: picked below
    // Lane counts measured per plan with the specialised kernels (rocprofv3, profiles/r03_experiments/k_static_plan_lanes.log):
    // config 5's columns kernel 9.4 -> 7.8 us with 16 columns per workgroup instead of pick_lanes' 8 (128-byte instead of
    // 64-byte runs; the 52-register kernels keep enough workgroups in flight), config 3's 75.3 -> 70.7 us with 32.
    uint32_t ca = force_ca, cb = force_cb;
    if (!ca && !cb) {
        // (a cycle-batched span of config 5 is 16 * k transforms: it keeps the 16-transform plan -- in the two-kernel
        // form only the lane counts depend on the transform count -- and with it the kernels compiled for that plan)
        if (n == 65536 && transforms % 16 == 0) {
            ca = 16;
            transforms = 16;
        } else if (n == 160000 && transforms == 100) {
            ca = 32;
        } else {
            // the power-of-two plans 10-12: from 256 column tiles per launch on (fewer: the lane counts that follow the
            // transform count give more workgroups)
            for (int sp = 10; sp <= 12; ++sp)
                if (n == kStaticPlans[sp].n) {
                    TiledPlan q{};
                    if (build_tiled_plan(n, 1, kStaticPlans[sp].force_ca, kStaticPlans[sp].force_cb, q) &&
                        transforms * ((q.S + q.CA - 1) / q.CA) >= 256) {
                        ca = kStaticPlans[sp].force_ca;
                        cb = kStaticPlans[sp].force_cb;
                    }
                }
        }
    }
    if (!build_tiled_plan(n, transforms, ca, cb, p, allow_split && small_split_enabled())) return false;
    if (!generic_radix_tiles_enabled())
        for (uint32_t q = 0; q < p.nf; ++q)
            if (is_generic_radix(p.fact[q])) return false;
    return true;
}

// The fold epilogue's lane grouping (see FoldProductEpi below): part of the plan.
constexpr bool plan_fold_groups(TiledPlan& p, uint64_t fold) {
    if (fold == 0 || p.n % fold != 0 || p.n / fold > 64) return false;
    if (p.R1 == 1) return true;  // whole transforms per lane: every alias is in the lane's own column
    const uint32_t d = (uint32_t)(fold % p.R1);
    uint32_t a = p.R1, b = d;
    while (b) { const uint32_t t = a % b; a = b; b = t; }  // a = gcd(R1, d), gcd(R1, 0) = R1
    const uint32_t o = p.R1 / a;
    if (o > p.CB || p.CB % o != 0) return false;
    const uint32_t w = p.CB / o, stride = p.R1 / o;
    if (stride % w != 0) return false;  // every tile full, no group runs into the next
    p.grp_w = w;
    p.grp_stride = stride;
    p.grp_shift = ilog2(w);
    return true;
}

// ---- per-plan specialisation (round 3) ---------------------------------------------------------------------------
// The plans BASELINE.json's configurations run -- config 5: 16 x 65536 points; config 3: 100 x 160000 forward with the
// fold epilogue and 100 x 16000 inverse -- are also compiled with the plan as a CONSTANT: SP > 0 selects static_plan(SP)
// inside the kernels instead of the plan argument, every `rest / ido`, `x * pitch`, `1 << lane_shift`, radix switch and
// pass loop folds, and what is left of a pass is its butterflies, twiddles and LDS traffic (the generic index
// arithmetic was about a third of a pass's VALU work: quarter-rate v_mul_lo / v_mul_hi per butterfly).  The launcher
// takes a specialisation only when the run-time plan equals the constant one field for field.
constexpr TiledPlan static_plan(int sp) {
    TiledPlan p{};
    (void)build_tiled_plan(kStaticPlans[sp].n, kStaticPlans[sp].transforms, kStaticPlans[sp].force_ca,
                           kStaticPlans[sp].force_cb, p);
    if (kStaticPlans[sp].fold) (void)plan_fold_groups(p, kStaticPlans[sp].fold);
    return p;
}
constexpr bool same_plan(const TiledPlan& a, const TiledPlan& b) {
    if (a.n != b.n || a.nf != b.nf || a.g != b.g || a.R1 != b.R1 || a.S != b.S || a.CA != b.CA || a.CB != b.CB ||
        a.ca_shift != b.ca_shift || a.cb_shift != b.cb_shift || a.grp_w != b.grp_w || a.grp_stride != b.grp_stride ||
        a.grp_shift != b.grp_shift)
        return false;
    for (int i = 0; i < 20; ++i)
        if (a.fact[i] != b.fact[i] || a.magic[i] != b.magic[i] || a.tw_off[i] != b.tw_off[i]) return false;
    return true;
}

// Twiddles of the BLOCK passes (g..nf-1) are the same for every block of every transform -- PT[(c-1)*ido + i], a few hundred
// entries -- so a static-plan kernel keeps them in LDS behind its tile (round 6): a pass then has no global load between its
// two barriers (the L2 round trip of seven twiddles per butterfly was half of a radix-8 pass under load: tools/ubench/
// tiled_timeline_c5.hip, profiles/r06_experiments/c_tiled_persistent.log).  0: the plan keeps its twiddles in global memory (generic radix,
// whole transforms per lane, tables beyond 8 KiB).
#ifndef JST_TILED_TW_LDS
#define JST_TILED_TW_LDS 1
#endif
#ifndef JST_TILED_PERSIST  // A/B switch: 0 = one workgroup per tile everywhere (no persistent columns / blocks kernels)
#define JST_TILED_PERSIST 1
#endif
constexpr uint32_t block_twiddle_entries(const TiledPlan& p) {
    if (!JST_TILED_TW_LDS || p.g == 0) return 0;
    uint64_t total = 0, ido = p.S;
    for (uint32_t q = p.g; q < p.nf; ++q) {
        if (is_generic_radix(p.fact[q])) return 0;
        ido /= p.fact[q];
        total += pass_table_entries(p.fact[q], ido);
    }
    return total <= 1024 ? (uint32_t)total : 0u;
}

// ---- Multiply -> Fold behind the last pass (filter/block_impl.cc:444-497: fftSignal -> multiply -> fold) -----------
// fold (dsp/fold/module_impl_native_cpu.cc:103-172) sums the `decim` aliases idx = (m - off + g * fold) mod n of output
// bin m, in F64, g ascending, and divides by decim; the addends are products spectrum[idx] * h[idx] formed with the
// Multiply module's arithmetic.  Result q of block k of the second kernel sits at idx = k + R1 * q, so the aliases of a
// bin live in the blocks k + g * (fold mod R1) (mod R1): an orbit of o = R1 / gcd(R1, fold mod R1) blocks, R1 / o apart.
// A workgroup that holds whole orbits (CB / o adjacent blocks from each of the o groups) has every addend of its bins
// in LDS: the spectrum (8 B per sample written and read again) and the product are never materialised.
struct FoldProductEpi {
    static constexpr bool kTile = true;
    float2* out;             // dense [transforms, fold]
    const float2* h;         // the other Multiply operand along the transform axis (broadcast over transforms)
    int64_t h_stride;
    uint32_t fold, decim;
    uint32_t off;            // scalar offset (mod n) when chan_offsets == nullptr
    const uint64_t* chan_offsets;
    uint32_t chan_count, chan_div;  // channel of transform t = (t / chan_div) % chan_count
    bool spectrum_first;     // operand order of the product
    // idx += fold in tile coordinates (filled by the launcher from the plan): row += dq, block += dk (carry into the
    // row at R1), lane group += grp_step (mod the number of groups); idx -= n is row -= nq
    uint32_t dq, dk, nq, grp_step;
    // heads > 1 (the Filter block's multi-head form, filter/block_impl.cc:350-582): ONE spectrum per transform meets
    // `heads` operand rows h + hd * h_head_stride, each with its own fold offset chan_offsets[hd]; out is dense
    // [transforms, heads, fold].  The spectrum tile has to survive all heads but the last: those form their products
    // inside the alias walk (operand loads from L2 in the loop), the last head multiplies in place like heads == 1.
    uint32_t heads = 1;
    int64_t h_head_stride = 0;
};
template <class E>
constexpr bool is_tile_epilogue = requires { E::kTile; };

// ---- multiply_constant -> unpad behind the last pass (filter/block_impl.cc:499-560: ifft -> normalize -> unpad) -----
// The launcher's layout makes `base` the transform index (unit stride over the flattened outer axes): positions below
// body_len go to row `base` of the dense body tensor, the rest to row `base` of the dense tail tensor, each scaled
// like core/multiply_constant (complex x real: two products).
struct StoreScaledUnpad {
    float2* body;
    float2* tail;
    float c;
    uint32_t body_len, tail_len;
    template <bool CONTIG>
    __device__ __forceinline__ void store(int64_t base, int64_t, int pos, float2 v) const {
        const float2 r = mk(v.x * c, v.y * c);
        // plain stores: these land in 64-byte runs that neighbouring workgroups complete to full lines in L2; written
        // through at agent scope (like the dense outputs of the spectrum kernels) config 3 lost 2-3 %
        if ((uint32_t)pos < body_len) body[base * (int64_t)body_len + pos] = r;
        else tail[base * (int64_t)tail_len + ((uint32_t)pos - body_len)] = r;
    }
};

// ... with phase_correction between the scale and the unpad (filter/block_impl.cc:499-560 with a frequency-shifted head:
// ifft -> normalize -> phase_correction -> unpad): every sample of transform `base` times the correction of its
// (channel, batch) cell, dsp/phase_correction/module_impl_native_cpu.cc:95-115 -- the table the module keeps, read here
// (written for THIS cycle by the previous cycle's overlap kernel, see launch_overlap_heads_phase).
struct StoreScaledPhaseUnpad {
    float2* body;
    float2* tail;
    float c;
    uint32_t body_len, tail_len;
    const float2* corr;  // [channels, batches]
    uint32_t batches, batch_div, channels, chan_div;  // cell of transform t: batch (t / batch_div) % batches, channel (t / chan_div) % channels
    template <bool CONTIG>
    __device__ __forceinline__ void store(int64_t base, int64_t, int pos, float2 v) const {
        const uint32_t t = (uint32_t)base;
        const uint32_t b = batches == 1 ? 0u : (t / batch_div) % batches;
        const uint32_t ch = channels == 1 ? 0u : (t / chan_div) % channels;
        const float2 r = cmul_full(mk(v.x * c, v.y * c), corr[ch * batches + b]);
        if ((uint32_t)pos < body_len) body[base * (int64_t)body_len + pos] = r;
        else tail[base * (int64_t)tail_len + ((uint32_t)pos - body_len)] = r;
    }
};

// Workgroup b runs on XCD b % 8 (observed placement, used for speed only) and every XCD has an L2 of its own.
// Neighbouring tiles share cache lines (a column tile's 128-byte runs start 8 bytes off the line grid whenever the
// row length is odd in elements; a block tile writes 64-byte halves of 128-byte lines), so XCD k takes a CONTIGUOUS
// run of tiles: the second touch of a line is an L2 hit / an L2 merge instead of a second HBM transaction.
__device__ __forceinline__ uint32_t xcd_contiguous_tile(uint32_t b, uint32_t grid) {
    const uint32_t q = grid >> 3, r = grid & 7u, k = b & 7u;
    return k * q + (k < r ? k : r) + (b >> 3);
}

__device__ __forceinline__ void outer_bases(const FftLayout& L, uint64_t t, int64_t& in_base,
                                            int64_t& out_base) {
    in_base = (int64_t)L.in_offset;
    out_base = (int64_t)L.out_offset;
    if (L.outer_rank == 1) {  // one batch axis (every BASELINE config): no 64-bit division per workgroup / per tile
        in_base += (int64_t)t * L.in_outer_stride[0];
        out_base += (int64_t)t * L.out_outer_stride[0];
        return;
    }
    for (int a = L.outer_rank - 1; a >= 0; --a) {
        const uint64_t c = t % L.outer_shape[a];
        t /= L.outer_shape[a];
        in_base += (int64_t)c * L.in_outer_stride[a];
        out_base += (int64_t)c * L.out_outer_stride[a];
    }
}

// One radix-IP pass over an LDS tile, IN PLACE: every thread first reads the inputs of all of its
// butterflies into registers, the workgroup synchronises, then the outputs go back to the same
// buffer (Stockham positions) -- one buffer instead of a ping-pong pair, i.e. half the LDS and
// twice the resident workgroups per CU.  The workgroup has at least tile/8 threads, so a thread
// owns at most ceil(8/IP) butterflies: NB below is a compile-time function of the radix.
// Element (x, lane) lives at x * pitch + lane; `lanes` (1 << lane_shift) independent
// sub-transforms of `len` points run side by side.
//   butterfly (i, k):  reads x = i + ido*(j + IP*k), writes x = i + ido*(k + l1loc*c)
//   twiddle:           PT[(c-1)*ido_glob + ig], ig = tw_i0 + lane*tw_lane + tw_is*i the GLOBAL i of
//                      the butterfly; PT is this pass's slice of the per-pass table (pocketfft's
//                      own layout, comp_twiddle :1513-1535: tw[(j-1)*(ido-1)+i-1] = W[j*l1*i]), so
//                      adjacent lanes read adjacent entries instead of gathering W at stride c*l1.
template <int IP>
constexpr int butterflies_per_thread() { return IP <= 3 ? 4 : (IP <= 7 ? 2 : 1); }

// LB: the barrier of a kernel that keeps global loads in flight ACROSS its passes (the persistent blocks kernel's prefetch of
// the next tile): release / acquire on LDS only -- `__syncthreads()` waits for vmcnt(0), i.e. for the prefetch.
template <bool LB>
__device__ __forceinline__ void tile_barrier() {
    if constexpr (LB) lds_barrier();
    else __syncthreads();
}

// REG: PT is the thread's own table, PT[r * (IP - 1) + c - 1] for its butterfly r -- registers of a kernel whose threads meet the
// same (lane, i) in every tile (the persistent columns kernel) -- instead of the pass's table in memory.
template <int IP, bool FWD, bool LB = false, bool REG = false>
__device__ __forceinline__ void tile_pass(float2* __restrict__ buf, const float2* __restrict__ PT,
                                          uint32_t len, uint32_t lane_shift, uint32_t live_lanes,
                                          uint32_t pitch, uint32_t ido, uint32_t ido_magic,
                                          uint32_t l1loc, uint32_t ido_glob, uint32_t tw_is,
                                          uint32_t tw_i0, uint32_t tw_lane) {
    constexpr int NB = butterflies_per_thread<IP>();
    const uint32_t nb = (len / IP) << lane_shift;
    float2 x[NB][IP];
    uint32_t wr_base[NB], step[NB];  // step = global i of the butterfly; 0xffffffff marks an idle slot
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        const uint32_t b = threadIdx.x + (uint32_t)r * blockDim.x;
        const uint32_t lane = b & ((1u << lane_shift) - 1u), rest = b >> lane_shift;
        step[r] = 0xffffffffu;
        wr_base[r] = 0;
        if (b < nb && lane < live_lanes) {
            const uint32_t k = ido > 1 ? __umulhi(rest, ido_magic) : rest;
            const uint32_t i = rest - k * ido;
            const float2* rd = buf + (i + ido * IP * k) * pitch + lane;
#pragma unroll
            for (int j = 0; j < IP; ++j) x[r][j] = rd[(uint32_t)j * ido * pitch];
            wr_base[r] = (i + ido * k) * pitch + lane;
            step[r] = tw_i0 + lane * tw_lane + tw_is * i;
        }
    }
    tile_barrier<LB>();  // every input of this pass is in registers
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        if (step[r] == 0xffffffffu) continue;
        butterfly_any<IP, FWD>(x[r]);
        if (step[r] != 0) {
#pragma unroll
            for (int c = 1; c < IP; ++c)
                x[r][c] = special_mul<FWD>(x[r][c], REG ? PT[r * (IP - 1) + (c - 1)] : PT[(uint32_t)(c - 1) * ido_glob + step[r]]);
        }
        float2* wr = buf + wr_base[r];
#pragma unroll
        for (int c = 0; c < IP; ++c) wr[(uint32_t)c * l1loc * ido * pitch] = x[r][c];
    }
    tile_barrier<LB>();  // outputs visible before the next pass (or the store) reads them
}

// One GENERIC odd-radix pass (any prime 13 <= ip <= kMaxGenericRadix) over an LDS tile, in place: pocketfft's passg
// (pocketfft.hh:1314-1421) with its CH / CX planes kept in registers.  passg computes, per butterfly,
//   H[0] = CC[0];  H[j], H[ip-j] = CC[j] +/- CC[ip-j]  (j = 1..ipph-1);   X[0] = H[0] + H[1] + ... + H[ipph-1]
//   X[l], X[ip-l] from the wal[] sums (terms 1,2 first, then two at a time, then a single one: the association
//   decides the rounding);   out[l], out[ip-l] = (X[l] +/- X[ip-l]) * twiddle.
// Work item = (butterfly, PAIR of l): it reads the butterfly's ip inputs once, forms every H on the way (two adds
// per input, cheaper than a pass over the tile and a barrier), and feeds both l of the pair -- the pass is bound by
// LDS bandwidth (one item per (butterfly, l) with wal[] in LDS: 35 reads per item, 7.2 of the 8050-point kernel's
// 24 us; profiles/r04_experiments/h_tiled_timeline_8050.log).  A WAVE owns 64 adjacent butterflies of ONE pair, so
// l, the wal[] walk and the loop counts are wave-uniform: the walk runs on the scalar unit and the wal[] reads are
// broadcasts.  Pair 0 also forms X[0].
// The results (four, five for pair 0) wait in registers for the barrier, then go to their Stockham positions.
// A wave owns at most kGenericSlots (pair, 64-butterfly chunk) tasks: build_tiled_plan checks that.

template <bool FWD>
__device__ __forceinline__ void tile_pass_generic(uint32_t ip, float2* __restrict__ buf, const float2* __restrict__ PT,
                                                  uint32_t len, uint32_t lane_shift, uint32_t live_lanes,
                                                  uint32_t pitch, uint32_t ido, uint32_t ido_magic,
                                                  uint32_t l1loc, uint32_t ido_glob, uint32_t tw_is,
                                                  uint32_t tw_i0, uint32_t tw_lane) {
    const uint32_t ipph = (ip + 1u) / 2u, but = len / ip, lane_mask = (1u << lane_shift) - 1u;
    const uint32_t rs = ido * pitch;  // distance of two inputs j, j+1 of a butterfly
    const uint32_t per_pair = but << lane_shift, chunks = (per_pair + 63u) >> 6;
    const uint32_t ntask = ((ipph - 1u + 1u) / 2u) * chunks;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwaves = blockDim.x >> 6;
    // wal[m] = W[m * n/ip] (conjugated for the forward transform), staged in LDS: read at wave-uniform addresses in the
    // same batch as the inputs -- as scalar loads from the table every wal[] pair was a second and third round trip
    // per step of the walk (s_load, s_waitcnt lgkmcnt(0)) behind the LDS one
    __shared__ float2 wal_s[kMaxGenericRadix + 1];
    if (threadIdx.x < ip) {
        float2 w = PT[(ip - 1u) * ido_glob + threadIdx.x];
        if (FWD) w.y = -w.y;
        wal_s[threadIdx.x] = w;
    }
    __syncthreads();
    float2 o[kGenericSlots][5];  // out[l1], out[ip-l1], out[l2], out[ip-l2], out[0]
    uint32_t wr_base[kGenericSlots];  // 0xffffffff marks an idle slot
#pragma unroll
    for (int r = 0; r < kGenericSlots; ++r) {
        const uint32_t task = wave + (uint32_t)r * nwaves;  // wave-uniform
        wr_base[r] = 0xffffffffu;
#pragma unroll
        for (int q = 0; q < 5; ++q) o[r][q] = mk(0.0f, 0.0f);
        if (task >= ntask) continue;
        const uint32_t pair = task / chunks, chunk = task - pair * chunks;
        const uint32_t e = (chunk << 6) + (threadIdx.x & 63u);
        const uint32_t lane = e & lane_mask, u = e >> lane_shift;
        if (e >= per_pair || lane >= live_lanes) continue;
        const uint32_t la = 1u + 2u * pair, lb = la + 1u;  // the pair's l (lb may be ipph: then only la is real)
        const bool two = lb < ipph;
        const uint32_t k = ido > 1 ? __umulhi(u, ido_magic) : u, i = u - k * ido;
        const float2* h = buf + (i + ido * ip * k) * pitch + lane;
        wr_base[r] = (i + ido * k) * pitch + lane;
        auto wal = [&](uint32_t m) { return wal_s[m]; };
        // terms 1 and 2
        const float2 c0 = h[0];
        float2 t1 = h[rs], t2 = h[2u * rs], u1 = h[(ip - 1u) * rs], u2 = h[(ip - 2u) * rs];
        float2 h1 = cadd(t1, u1), h2 = cadd(t2, u2), g1 = csub(t1, u1), g2 = csub(t2, u2);  // H[1], H[2], H[ip-1], H[ip-2]
        float2 x0 = c0;
        x0.x += h1.x; x0.y += h1.y;
        x0.x += h2.x; x0.y += h2.y;
        float2 xa, xac, xb, xbc;
        uint32_t ia = 2u * la, ib = 2u * lb;
        {
            const float2 w1 = wal(la), w2 = wal(ia);
            xa.x = c0.x + w1.x * h1.x + w2.x * h2.x;
            xa.y = c0.y + w1.x * h1.y + w2.x * h2.y;
            xac.x = -w1.y * g1.y - w2.y * g2.y;
            xac.y = w1.y * g1.x + w2.y * g2.x;
        }
        if (two) {
            const float2 w1 = wal(lb), w2 = wal(ib);
            xb.x = c0.x + w1.x * h1.x + w2.x * h2.x;
            xb.y = c0.y + w1.x * h1.y + w2.x * h2.y;
            xbc.x = -w1.y * g1.y - w2.y * g2.y;
            xbc.y = w1.y * g1.x + w2.y * g2.x;
        } else {
            xb = xbc = mk(0.0f, 0.0f);
            ib = 0;
        }
        auto step = [&](uint32_t& iw, uint32_t l) {  // iwal += l; if (iwal > ip) iwal -= ip;  (never == ip: ip is prime)
            iw += l;
            if (iw > ip) iw -= ip;
            return wal(iw);
        };
        uint32_t j = 3u;
        const float2* hj = h + 3u * rs;          // CC[j]
        const float2* hjc = h + (ip - 3u) * rs;  // CC[ip - j]
        for (; j < ipph - 1u; j += 2u, hj += 2u * rs, hjc -= 2u * rs) {
            t1 = hj[0]; t2 = hj[rs]; u1 = hjc[0]; u2 = *(hjc - rs);
            h1 = cadd(t1, u1); g1 = csub(t1, u1);  // H[j], H[ip-j]
            h2 = cadd(t2, u2); g2 = csub(t2, u2);  // H[j+1], H[ip-j-1]
            x0.x += h1.x; x0.y += h1.y;
            x0.x += h2.x; x0.y += h2.y;
            {
                const float2 xw = step(ia, la);
                const float2 xw2 = step(ia, la);
                xa.x += h1.x * xw.x + h2.x * xw2.x;
                xa.y += h1.y * xw.x + h2.y * xw2.x;
                xac.x -= g1.y * xw.y + g2.y * xw2.y;
                xac.y += g1.x * xw.y + g2.x * xw2.y;
            }
            if (two) {
                const float2 xw = step(ib, lb);
                const float2 xw2 = step(ib, lb);
                xb.x += h1.x * xw.x + h2.x * xw2.x;
                xb.y += h1.y * xw.x + h2.y * xw2.x;
                xbc.x -= g1.y * xw.y + g2.y * xw2.y;
                xbc.y += g1.x * xw.y + g2.x * xw2.y;
            }
        }
        for (; j < ipph; ++j, hj += rs, hjc -= rs) {
            t1 = hj[0]; u1 = hjc[0];
            h1 = cadd(t1, u1); g1 = csub(t1, u1);
            x0.x += h1.x; x0.y += h1.y;
            {
                const float2 xw = step(ia, la);
                xa.x += h1.x * xw.x;
                xa.y += h1.y * xw.x;
                xac.x -= g1.y * xw.y;
                xac.y += g1.x * xw.y;
            }
            if (two) {
                const float2 xw = step(ib, lb);
                xb.x += h1.x * xw.x;
                xb.y += h1.y * xw.x;
                xbc.x -= g1.y * xw.y;
                xbc.y += g1.x * xw.y;
            }
        }
        float2 a1 = cadd(xa, xac), a2 = csub(xa, xac), b1 = cadd(xb, xbc), b2 = csub(xb, xbc);
        const uint32_t st = tw_i0 + lane * tw_lane + tw_is * i;  // the GLOBAL i of the butterfly
        if (st != 0) {
            a1 = special_mul<FWD>(a1, PT[(la - 1u) * ido_glob + st]);
            a2 = special_mul<FWD>(a2, PT[(ip - la - 1u) * ido_glob + st]);
            if (two) {
                b1 = special_mul<FWD>(b1, PT[(lb - 1u) * ido_glob + st]);
                b2 = special_mul<FWD>(b2, PT[(ip - lb - 1u) * ido_glob + st]);
            }
        }
        o[r][0] = a1; o[r][1] = a2; o[r][2] = b1; o[r][3] = b2; o[r][4] = x0;
    }
    __syncthreads();  // every input of this pass has been read
    JST_TSTAMP(13);
    const uint32_t ws = l1loc * ido * pitch;
#pragma unroll
    for (int r = 0; r < kGenericSlots; ++r) {
        if (wr_base[r] == 0xffffffffu) continue;
        const uint32_t task = wave + (uint32_t)r * nwaves;
        const uint32_t pair = task / chunks, la = 1u + 2u * pair, lb = la + 1u;
        float2* wr = buf + wr_base[r];
        wr[la * ws] = o[r][0];
        wr[(ip - la) * ws] = o[r][1];
        if (lb < ipph) {
            wr[lb * ws] = o[r][2];
            wr[(ip - lb) * ws] = o[r][3];
        }
        if (pair == 0) wr[0] = o[r][4];
    }
    __syncthreads();  // outputs visible before the next pass (or the store) reads them
}

// GEN: the kernel is compiled for plans that hold a generic radix (a second set of kernels: with the generic pass
// inlined into the one set, every kernel of this file paid registers / scratch for a pass most plans never run)
template <bool FWD, bool GEN, bool LB = false>
__device__ __forceinline__ void tile_pass_any(uint32_t ip, float2* buf, const float2* PT, uint32_t len,
                                              uint32_t lane_shift, uint32_t live_lanes, uint32_t pitch,
                                              uint32_t ido, uint32_t ido_magic, uint32_t l1loc,
                                              uint32_t ido_glob, uint32_t tw_is, uint32_t tw_i0,
                                              uint32_t tw_lane) {
#define JST_TP(IP)                                                                                \
    tile_pass<IP, FWD, LB>(buf, PT, len, lane_shift, live_lanes, pitch, ido, ido_magic, l1loc,    \
                       ido_glob, tw_is, tw_i0, tw_lane)
    switch (ip) {
        case 2: JST_TP(2); break;
        case 3: JST_TP(3); break;
        case 4: JST_TP(4); break;
        case 5: JST_TP(5); break;
        case 7: JST_TP(7); break;
        case 8: JST_TP(8); break;
        case 11: JST_TP(11); break;
        default:
            if constexpr (GEN)
                tile_pass_generic<FWD>(ip, buf, PT, len, lane_shift, live_lanes, pitch, ido, ido_magic, l1loc,
                                       ido_glob, tw_is, tw_i0, tw_lane);
            else
                __builtin_unreachable();  // build_tiled_plan admits 2,3,4,5,7,8,11 and generic primes only
            break;
    }
#undef JST_TP
}

// at least tile/8 threads (the in-place passes hold <= 8 points per thread); a multiple of 256 so
// that every SIMD of the CU gets the same number of wavefronts and a second / third workgroup fits
// the fewest threads the passes [first, last) of a plan need on a tile of `tile_elems` elements: a thread owns at most
// butterflies_per_thread(radix) butterflies of a pass (tile_pass)
constexpr uint64_t min_threads_for_passes(const TiledPlan& P, uint32_t first, uint32_t last, uint64_t tile_elems) {
    uint64_t need = 64;
    for (uint32_t q = first; q < last; ++q) {
        const uint32_t ip = P.fact[q];
        if (is_generic_radix(ip)) {
            if (generic_pass_threads(ip, tile_elems) > need) need = generic_pass_threads(ip, tile_elems);
            continue;
        }
        const uint64_t per = ip <= 3 ? 4 : (ip <= 7 ? 2 : 1);
        const uint64_t nb = tile_elems / ip;
        if ((nb + per - 1) / per > need) need = (nb + per - 1) / per;
    }
    return need;
}
constexpr unsigned threads_for(uint64_t tile_elems, uint64_t min_threads = 0) {
    uint64_t t = ((tile_elems / 8 + 255) / 256) * 256;
    if (t < 256) t = 256;
    if (t < min_threads) t = ((min_threads + 63) / 64) * 64;  // a generic-radix pass may need more (wave-granular tasks)
    if (t > (uint64_t)kMaxThreads) t = kMaxThreads;
    return (unsigned)t;
}

// ---- kernel A: passes 0..g-1 on CA adjacent columns ---------------------------------------------
template <bool FWD, class Pro, int SP = 0, bool GEN = false>
__global__ __launch_bounds__(kMaxThreads, GEN ? 4 : JST_TILED_MIN_WAVES) void fft_tile_columns_kernel(const FftLayout L,
                                                                    const TiledPlan Prt,
                                                                    const float2* __restrict__ W,
                                                                    const Pro pro,
                                                                    float2* __restrict__ scratch) {
    constexpr TiledPlan PS = static_plan(SP);  // SP > 0: the plan is a constant and the argument is ignored
    const TiledPlan& P = SP > 0 ? PS : Prt;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* buf0 = reinterpret_cast<float2*>(smem_raw);
    JST_TSTAMP(0);
    const uint32_t tiles_per_t = (P.S + P.CA - 1) >> P.ca_shift;
    const uint32_t bid = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    const uint64_t t = bid / tiles_per_t;
    const uint32_t c0 = (bid % tiles_per_t) << P.ca_shift;
    const uint32_t live = (P.S - c0 < P.CA) ? (P.S - c0) : P.CA;
    const uint32_t tile = P.R1 << P.ca_shift;
    int64_t in_base, out_base;
    outer_bases(L, t, in_base, out_base);
    // Eight loads in flight per thread (a tile is at most 8 elements per thread): written as a plain loop, hipcc
    // waits for every load before the LDS write that consumes it -- eight serial HBM round trips per workgroup
    // (round 1: ~10 of the ~20 us a column workgroup lived).
    for (uint32_t i0 = threadIdx.x; i0 < tile; i0 += 8 * blockDim.x) {
        float2 v[8];
        bool ok[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {  // unconditional loads from clamped positions: no branch between them
            const uint32_t idx = i0 + (uint32_t)k * blockDim.x;
            const uint32_t r = idx >> P.ca_shift, col = idx & (P.CA - 1u);
            ok[k] = idx < tile && col < live;
            const uint32_t rc = r < P.R1 ? r : P.R1 - 1u, cc = col < live ? col : live - 1u;
            v[k] = pro.template load<false>(in_base, L.in_axis_stride, (int)(c0 + cc + P.S * rc));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (ok[k]) buf0[i0 + (uint32_t)k * blockDim.x] = v[k];
    }
    __syncthreads();
    JST_TSTAMP(1);  // tile loaded
    uint32_t l1 = 1, m = P.R1;
#pragma unroll
    for (uint32_t p = 0; p < P.g; ++p) {
        const uint32_t ip = P.fact[p];
        m /= ip;  // local ido'
        tile_pass_any<FWD, GEN>(ip, buf0, W + P.tw_off[p], P.R1, P.ca_shift, live, P.CA, m, P.magic[p], l1,
                           m * P.S, P.S, c0, 1u);
        l1 *= ip;
        JST_TSTAMP(2 + p);  // pass p done
    }
    float2* o = scratch + t * P.n;
    for (uint32_t idx = threadIdx.x; idx < tile; idx += blockDim.x) {
        const uint32_t r = idx >> P.ca_shift, col = idx & (P.CA - 1u);
        // plain store: the scratch image is re-read at once by the second kernel and L2 / MALL absorb most of it; written
        // through at agent scope (see JST_STORE_AUX in fft_lds.hh) config 3 lost 3 % (128 MB per cycle)
#ifdef JST_TILED_SC1  // A/B switch
        if (col < live) store_agent(o + (c0 + col + P.S * r), buf0[idx]);
#else
        if (col < live) o[c0 + col + P.S * r] = buf0[idx];
#endif
    }
    JST_TSTAMP_FLUSH();  // stores issued
}

// ---- kernel A, persistent (round 6) --------------------------------------------------------------------------------
// Static plans, dense rows, launches of several transforms per workgroup.  A workgroup owns ONE column tile c0 and walks the
// transforms lane, lane + lanes, ...: what a thread needs besides the data is then the same in every tile -- the twiddles
// W[c * l1 * (c0 + column + S i')] of its butterflies (18-20 per thread: a column tile's table is as large as the tile and was
// re-read from L2 for every tile) and its eight window taps -- and stays in registers; the next transform's elements are
// requested into registers before the passes of this one (buffer loads: one descriptor per row, one byte offset per thread,
// wave-uniform offsets per element); all barriers are LDS-only; the passes have no memory instruction, so the requests stay in
// flight until the next commit.  One workgroup per tile, the tile's load -> passes -> store chain ran once per workgroup
// and a CU's slots were two-thirds full (tools/ubench/tiled_timeline_c5.hip: 6.5 of 8 workgroups alive, load 5.6 of 11.4 us).
template <int SP>
constexpr uint32_t columns_twiddle_count() {
    constexpr TiledPlan P = static_plan(SP);
    uint32_t n = 0;
    for (uint32_t p = 0; p < P.g; ++p) {
        const uint32_t ip = P.fact[p];
        n += (ip <= 3 ? 4u : (ip <= 7 ? 2u : 1u)) * (ip - 1u);
    }
    return n;
}
template <int SP>
constexpr uint32_t columns_twiddle_offset(uint32_t pass) {
    constexpr TiledPlan P = static_plan(SP);
    uint32_t n = 0;
    for (uint32_t p = 0; p < pass; ++p) {
        const uint32_t ip = P.fact[p];
        n += (ip <= 3 ? 4u : (ip <= 7 ? 2u : 1u)) * (ip - 1u);
    }
    return n;
}
constexpr bool columns_pipe_eligible(const TiledPlan& p) {
    if (!JST_TILED_PERSIST || p.g == 0) return false;
    for (uint32_t q = 0; q < p.g; ++q)
        if (is_generic_radix(p.fact[q])) return false;
    const uint64_t tile = (uint64_t)p.R1 * p.CA;
    const uint64_t t = threads_for(tile, min_threads_for_passes(p, 0, p.g, tile));
    return tile == 8ull * t && t % p.CA == 0;  // eight elements per thread, a thread's elements in ONE column
}
template <int SP>
constexpr unsigned columns_pipe_threads() {
    constexpr TiledPlan P = static_plan(SP);
    return threads_for((uint64_t)P.R1 * P.CA, min_threads_for_passes(P, 0, P.g, (uint64_t)P.R1 * P.CA));
}

template <bool FWD, int SP, uint32_t PASS>
__device__ __forceinline__ void columns_pipe_passes(float2* buf0, const float2* tw, uint32_t live, uint32_t c0) {
    constexpr TiledPlan P = static_plan(SP);
    if constexpr (PASS < P.g) {
        constexpr uint32_t ip = P.fact[PASS];
        uint32_t l1 = 1, m = P.R1;
        for (uint32_t q = 0; q <= PASS; ++q) {
            m /= P.fact[q];
            if (q < PASS) l1 *= P.fact[q];
        }
        tile_pass<(int)ip, FWD, true, true>(buf0, tw + columns_twiddle_offset<SP>(PASS), P.R1, P.ca_shift, live, P.CA, m, P.magic[PASS], l1,
                                            m * P.S, P.S, c0, 1u);
        columns_pipe_passes<FWD, SP, PASS + 1>(buf0, tw, live, c0);
    }
}

template <bool FWD, class Pro, int SP>
__global__ __launch_bounds__(columns_pipe_threads<SP>(), 4) void fft_tile_columns_pipe_kernel(const FftLayout L, const float2* __restrict__ W,
                                                                                             const Pro pro, float2* __restrict__ scratch) {
    constexpr TiledPlan P = static_plan(SP);
    constexpr uint32_t T = columns_pipe_threads<SP>();
    constexpr uint32_t kTw = columns_twiddle_count<SP>();
    constexpr uint32_t dr = T >> P.ca_shift;  // rows between two of a thread's elements
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* buf0 = reinterpret_cast<float2*>(smem_raw);
    const uint32_t tiles_per_t = (P.S + P.CA - 1) >> P.ca_shift;
    const uint32_t j = xcd_contiguous_tile(blockIdx.x, gridDim.x);  // neighbours in j: the same XCD (they share cache lines)
    const uint32_t ct = j % tiles_per_t, lanes = gridDim.x / tiles_per_t;
    uint32_t t = j / tiles_per_t;
    if (t >= lanes || t >= (uint32_t)L.transforms) return;  // (a grid that is not a multiple of the tile count)
    const uint32_t c0 = ct << P.ca_shift;
    const uint32_t live = (P.S - c0 < P.CA) ? (P.S - c0) : P.CA;
    const uint32_t tid = threadIdx.x, col = tid & (P.CA - 1u), r0 = tid >> P.ca_shift;
    // element k of this thread: row r0 + k dr of column c0 + col, i.e. position c0 + col + S (r0 + k dr) of the transform
    const uint32_t voff = (c0 + col + P.S * r0) * (uint32_t)sizeof(float2);
    // a column past the (ragged last) tile: out of every descriptor's bounds -- loads return zeros, stores are dropped.  Only the
    // VGPR offset is bounds-checked (the scalar offset is added behind the check), so the padded prologue, whose rows end
    // at `valid`, carries the whole position in the VGPR offset.
    const uint32_t voff_lane = col < live ? voff : 0x80000000u;
    constexpr bool kBoundInVgpr = requires { pro.valid; };
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    uint32_t row_bytes = P.n * (uint32_t)sizeof(float2);
    if constexpr (requires { pro.valid; }) row_bytes = pro.valid * (uint32_t)sizeof(float2);  // the pad reads as zeros: the descriptor's bound
    float2 pv[8];
    auto prefetch = [&](uint32_t tt) {
        int64_t ib, ob;
        outer_bases(L, tt, ib, ob);
        const rsrc_t r_in = make_rsrc(pro.row(ib), row_bytes);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t step = (uint32_t)k * dr * P.S * (uint32_t)sizeof(float2);
            const v2f v = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r_in, kBoundInVgpr ? voff_lane + step : voff_lane,
                                                                                          kBoundInVgpr ? 0u : step, JST_TILED_STREAM_LOADS ? 2 : 0));
            pv[k] = mk(v.x, v.y);
        }
    };
    prefetch(t);
    // this thread's twiddles: butterfly r of pass p is b = tid + r T -> (lane, i); step = c0 + lane + S i (tile_pass)
    float2 tw[kTw];
    {
        uint32_t at = 0, m = P.R1;
#pragma unroll
        for (uint32_t p = 0; p < P.g; ++p) {
            const uint32_t ip = P.fact[p];
            m /= ip;
            const uint32_t nbt = ip <= 3 ? 4u : (ip <= 7 ? 2u : 1u);
#pragma unroll
            for (uint32_t r = 0; r < nbt; ++r) {
                const uint32_t b = tid + r * T, lane = b & (P.CA - 1u), rest = b >> P.ca_shift;
                const uint32_t i = rest % m;
                uint32_t step = c0 + lane + P.S * i;
                if (lane >= live || b >= ((P.R1 / ip) << P.ca_shift)) step = 0;  // an idle slot: any entry
#pragma unroll
                for (uint32_t c = 1; c < ip; ++c) tw[at++] = W[P.tw_off[p] + (c - 1u) * (m * P.S) + step];
            }
        }
    }
    // the window taps of this thread's eight positions: a second tile in LDS behind the first (in registers beside the twiddles
    // the kernel spilled 16 dwords, and a scratch reload waits vmcnt(0)); each thread reads back what it wrote itself
    float2* winl = buf0 + (P.R1 << P.ca_shift);
    if constexpr (Pro::kHasOperand) {
        const rsrc_t r_w = make_rsrc(pro.operand_row(), P.n * (uint32_t)sizeof(float2));
#pragma unroll
        for (int k = 0; k < 8; ++k) winl[tid + (uint32_t)k * T] = buf_load_f2(r_w, voff_lane, (uint32_t)k * dr * P.S * (uint32_t)sizeof(float2));
    }
    // Everything requested so far has landed before the loop is entered: inside it hipcc then waits for the prefetched registers
    // only (vmcnt(15..8): the eight requests, the eight stores behind them) -- with the twiddles' loads still pending at the
    // loop's head it put `s_waitcnt vmcnt(10)` / `vmcnt(7)` in front of their first use in EVERY iteration, i.e. waited for
    // part of the prefetch in the middle of the passes.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    while (true) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if constexpr (Pro::kHasOperand) buf0[tid + (uint32_t)k * T] = pro.apply(pv[k], winl[tid + (uint32_t)k * T]);
            else buf0[tid + (uint32_t)k * T] = pv[k];
        }
        lds_barrier();
        const uint32_t tn = t + lanes;
        const bool more = tn < (uint32_t)L.transforms;
        if (more) prefetch(tn);
        columns_pipe_passes<FWD, SP, 0>(buf0, tw, live, c0);
        const rsrc_t r_out = make_rsrc(scratch + (size_t)t * P.n, P.n * (uint32_t)sizeof(float2));
#pragma unroll
        for (int k = 0; k < 8; ++k) {  // plain stores: the second kernel re-reads the image at once (see the one-tile form)
            const float2 v = buf0[tid + (uint32_t)k * T];
            __builtin_amdgcn_raw_buffer_store_b64(v2u{f2u(v.x), f2u(v.y)}, r_out, voff_lane, (uint32_t)k * dr * P.S * (uint32_t)sizeof(float2), 0);
        }
        if (!more) break;
        lds_barrier();  // every thread has read its results before the next tile is committed
        t = tn;
    }
}

// ---- kernel B: passes g..nf-1 on CB adjacent blocks (or whole transforms when g == 0) -----------
// PERSIST (round 6; static plans with R1 > 1 whose block twiddles live in LDS and whose tile is at most eight elements per
// thread: persist_eligible): the workgroup takes the tiles blockIdx.x, + gridDim.x, ... of `ntiles` and requests the NEXT
// tile's elements into registers before it runs the passes of this one.  One workgroup per tile, what a tile's load -> passes
// -> store chain costs barely shows in the kernel: with the block twiddles in LDS a workgroup of config 5's 128-transform
// launch lives 8.45 instead of 12.6 us and the launch takes 36.9 instead of 38.7 us -- fewer workgroups are alive on average
// (1.8 per CU of 3-4 slots; synthetic workgroups of the same shape that only wait fill 88-95 % of their slots, refill gap
// 0.5-0.8 us: tools/ubench/launch_rate.hip, so it is not the dispatcher), every workgroup pays its start-up (arguments, table,
// the first tile's round trip with every other new workgroup's) and the two or three workgroups of a CU do not interleave
// their phases (tools/ubench/tiled_timeline_c5.hip, profiles/r06_experiments/c_tiled_persistent.log); a persistent loop
// WITHOUT the prefetch serialises load and passes in those few workgroups (round 5, u_...log).  With both, a tile costs
// max(load, passes + epilogue) and the start-up is paid once per workgroup.  All barriers are LDS-only (tile_barrier<true>): the
// ordinary workgroup barrier waits for vmcnt(0), i.e. for the prefetch.
template <bool FWD, class Pro, class Epi, int SP = 0, bool GEN = false, bool PERSIST = false>
__global__ __launch_bounds__(kMaxThreads, GEN ? 4 : JST_TILED_MIN_WAVES) void fft_tile_blocks_kernel(const FftLayout L,
                                                                   const TiledPlan Prt,
                                                                   const float2* __restrict__ W,
                                                                   const Pro pro, const Epi epi,
                                                                   const float2* __restrict__ scratch,
                                                                   const uint32_t ntiles) {
    static_assert(!PERSIST || (SP > 0 && !GEN), "persistent form: static plans only");
    constexpr TiledPlan PS = static_plan(SP);  // SP > 0: the plan is a constant and the argument is ignored
    const TiledPlan& P = SP > 0 ? PS : Prt;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ int64_t lane_in[32], lane_out[32];  // per lane of the tile: tensor row bases
    __shared__ uint32_t lane_off[32];              // fold epilogue: per lane, the fold offset of its transform (mod n)
    const uint32_t pitch = P.CB | 1u;  // odd pitch: the x-major global loops stay conflict-free
    float2* buf0 = reinterpret_cast<float2*>(smem_raw);
    // static plans: the block passes' twiddle tables behind the tile (block_twiddle_entries), requested with the tile's loads
    constexpr uint32_t kTwLds = SP > 0 && !GEN ? block_twiddle_entries(PS) : 0u;
    float2* twl = buf0 + (size_t)PS.S * (PS.CB | 1u);
    // one workgroup per tile: the table is REQUESTED here and written to LDS behind the requests for the tile (fill_lane_table
    // below) -- written at once, its L2 round trip stood in front of the tile's loads
    constexpr uint32_t kTwThreads = SP > 0 ? threads_for((uint64_t)PS.S * PS.CB, min_threads_for_passes(PS, PS.g, PS.nf, (uint64_t)PS.S * PS.CB)) : 1u;
    constexpr uint32_t kTwPer = (kTwLds > 0 && !PERSIST) ? (kTwLds + kTwThreads - 1) / kTwThreads : 0u;
    float2 twv[kTwPer > 0 ? kTwPer : 1];
    if constexpr (kTwPer > 0) {
#pragma unroll
        for (uint32_t j = 0; j < kTwPer; ++j) {
            const uint32_t e = threadIdx.x + j * kTwThreads;
            twv[j] = W[PS.tw_off[PS.g] + (e < kTwLds ? e : kTwLds - 1u)];
        }
    }
    JST_TSTAMP(0);
    const uint32_t grp_gap = P.grp_stride - P.grp_w;  // blocks skipped between two groups of lanes (see block_of below)
    const uint32_t tile_count = PERSIST ? ntiles : gridDim.x;
    // R1 > 1: a tile is CB adjacent blocks of ONE transform; R1 == 1: CB adjacent transforms
    auto decode = [&](uint32_t tile_index, uint64_t& t0, uint32_t& k0, uint32_t& live) {
        const uint32_t bid = xcd_contiguous_tile(tile_index, tile_count);
        if (P.R1 > 1) {
            const uint32_t tiles_per_t = (P.R1 + P.CB - 1) >> P.cb_shift;
            t0 = bid / tiles_per_t;
            k0 = (bid % tiles_per_t) * P.grp_w;  // grp_w == CB unless the lanes are split into alias groups
            live = P.grp_w != P.CB ? P.CB : ((P.R1 - k0 < P.CB) ? (P.R1 - k0) : P.CB);
        } else {
            t0 = (uint64_t)bid << P.cb_shift;
            k0 = 0;
            live = (uint32_t)((L.transforms - t0 < P.CB) ? (L.transforms - t0) : P.CB);
        }
    };
    // PERSIST: the elements idx = threadIdx.x + k * blockDim.x (k < 8) of a tile, requested one tile ahead
    float2 pv[8];
    auto prefetch = [&](uint32_t tile_index) {
        uint64_t t0;
        uint32_t k0, live;
        decode(tile_index, t0, k0, live);
        const uint32_t tile = P.S * live;
        // one descriptor per tile (wave-uniform: SGPRs) and a 32-bit byte offset per element -- with flat addresses hipcc keeps
        // eight 64-bit pointers alive through the loop (80 VGPRs and scratch); elements past the tile read as zeros (the
        // descriptor's bound), nobody commits them.  The thread index is made opaque per tile so that the offsets are
        // recomputed (two or three VALU instructions) instead of living in registers across the passes.
        const uint32_t span = P.grp_w == P.CB ? tile : (P.CB / P.grp_w - 1u) * P.grp_stride * P.S + P.grp_w * P.S;
#ifdef JST_TILED_DIAG_NOLOAD  // timing diagnostics only: every tile reads the first one (cache resident)
        t0 = 0; k0 = 0;
#endif
        const rsrc_t r_tile = make_rsrc(scratch + (t0 * P.R1 + k0) * P.S, span * (uint32_t)sizeof(float2));
        uint32_t tid = threadIdx.x;
        __asm__ volatile("" : "+v"(tid));
        typedef float v2f __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t idx = tid + (uint32_t)k * blockDim.x;
            uint32_t off = idx;
            if (P.grp_w != P.CB) {  // lanes in groups grp_stride blocks apart (fold epilogue)
                const uint32_t cidx = idx < tile ? idx : tile - 1u;
                off = cidx + ((cidx / P.S) >> P.grp_shift) * grp_gap * P.S;
            }
            const v2f v = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r_tile, off * (uint32_t)sizeof(float2), 0, JST_TILED_STREAM_LOADS ? 2 : 0));
            pv[k] = mk(v.x, v.y);
        }
    };
    uint32_t tile_index = blockIdx.x;
    if constexpr (PERSIST) {
        prefetch(tile_index);
        // the twiddle table behind the first tile's requests: both round trips overlap (2.3 us in front of the first tile
        // otherwise; the table's LDS writes wait for its own loads, which retire behind the tile's)
        if constexpr (kTwLds > 0)
            for (uint32_t e = threadIdx.x; e < kTwLds; e += blockDim.x) twl[e] = W[PS.tw_off[PS.g] + e];
        // hipcc's waits for the prefetched registers count the VMEM instructions issued BEHIND the requests -- at the loop's
        // head it takes the minimum over the ways in: none on the way in from here, the epilogue's stores on the back edge,
        // so every commit waited vmcnt(7..0): for all of the previous tile's stores to be acknowledged.  As many stores on
        // this way in: out of bounds of an empty descriptor, dropped by the memory unit, counted by the compiler.
        if constexpr (!is_tile_epilogue<Epi>) {
            constexpr uint32_t kT = threads_for((uint64_t)PS.S * PS.CB, min_threads_for_passes(PS, PS.g, PS.nf, (uint64_t)PS.S * PS.CB));
            constexpr uint32_t iters = ((PS.S << PS.cb_shift) + kT - 1) / kT;
            const rsrc_t r_none = make_rsrc(scratch, 0u);
#pragma unroll
            for (uint32_t it = 0; it < iters; ++it) __builtin_amdgcn_raw_buffer_store_b32(it, r_none, threadIdx.x * 4u, it * 4096u, 0);
        }
    }
    while (true) {
    uint64_t t0;
    uint32_t k0, live;
    if constexpr (PERSIST) JST_TSTAMP(11);  // top of a tile
#ifdef JST_TILED_TIMELINE
    if constexpr (PERSIST && !is_tile_epilogue<Epi>) { const uint32_t round_ = (tile_index - blockIdx.x) / gridDim.x; if (round_ < 3) JST_TSTAMP(6 + round_); }
#endif
    decode(tile_index, t0, k0, live);
    // PERSIST (R1 > 1): every lane of the tile belongs to transform t0 -- its bases and fold offset are wave-uniform values in
    // scalar registers, not a table in LDS (and one barrier less per tile)
    int64_t tile_in = 0, tile_out = 0;
    uint32_t tile_off = 0;
    if constexpr (PERSIST) {
        outer_bases(L, t0, tile_in, tile_out);
        if constexpr (is_tile_epilogue<Epi>)
            tile_off = epi.chan_offsets ? (uint32_t)(epi.chan_offsets[(t0 / epi.chan_div) % epi.chan_count] % P.n) : epi.off;
    }
    // one workgroup per tile: the per-lane table of row bases (and fold offsets).  The scratch image is addressed without it, so
    // with g > 0 it is filled BETWEEN the requests for the tile and their commit (its divisions run while the loads are in
    // flight, and the barrier behind the tile publishes it too); whole transforms (g == 0) need it for the loads themselves.
    auto fill_lane_table = [&]() {
    if constexpr (kTwPer > 0) {
#pragma unroll
        for (uint32_t j = 0; j < kTwPer; ++j) {
            const uint32_t e = threadIdx.x + j * kTwThreads;
            if (e < kTwLds) twl[e] = twv[j];
        }
    }
    if (threadIdx.x < live) {
        int64_t ib, ob;
        outer_bases(L, P.R1 > 1 ? t0 : t0 + threadIdx.x, ib, ob);
        lane_in[threadIdx.x] = ib;
        lane_out[threadIdx.x] = ob;
        if constexpr (is_tile_epilogue<Epi>) {  // 64-bit divisions: once per lane, not once per output bin
            const uint64_t t = P.R1 > 1 ? t0 : t0 + threadIdx.x;
            lane_off[threadIdx.x] = epi.chan_offsets
                ? (uint32_t)(epi.chan_offsets[(t / epi.chan_div) % epi.chan_count] % P.n) : epi.off;
        }
    }
    };
    if constexpr (!PERSIST)
        if (P.g == 0) {
            fill_lane_table();
            __syncthreads();
        }
    const uint32_t tile = P.S * live;
    // block of lane kb: CB adjacent blocks, or CB / grp_w groups of grp_w adjacent blocks grp_stride apart
    auto block_of = [&](uint32_t kb) { return k0 + kb + (kb >> P.grp_shift) * grp_gap; };
    // load: x fastest (contiguous in memory for both the dense scratch and a dense input row)
    const float2* blk = scratch + (t0 * P.R1 + k0) * P.S;  // only dereferenced when g > 0
    // eight loads in flight per thread: unconditional loads from clamped indices (a conditional load is a branch and
    // hipcc drains vmcnt at each of them), the g == 0 / g > 0 choice hoisted out of the unrolled body
    auto load_tile = [&](auto from_scratch) {
        bool first = true;
        for (uint32_t i0 = threadIdx.x; i0 < tile; i0 += 8 * blockDim.x) {
            float2 v[8];
            uint32_t slot[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t idx = i0 + (uint32_t)k * blockDim.x;
                const uint32_t cidx = idx < tile ? idx : tile - 1u;
                const uint32_t kb = cidx / P.S, x = cidx - kb * P.S;
                if constexpr (decltype(from_scratch)::value) v[k] = stream_load(blk + (cidx + (kb >> P.grp_shift) * grp_gap * P.S));
                else v[k] = pro.template load<false>(lane_in[kb], L.in_axis_stride, (int)x);
                slot[k] = idx < tile ? x * pitch + kb : 0xffffffffu;
            }
            if constexpr (decltype(from_scratch)::value)
                if (first) fill_lane_table();
            first = false;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (slot[k] != 0xffffffffu) buf0[slot[k]] = v[k];
        }
        if constexpr (decltype(from_scratch)::value)
            if (first) fill_lane_table();  // (a thread without an element of the tile)
    };
    if constexpr (PERSIST) {  // the tile arrived in registers (requested while the previous tile was in the passes)
        JST_TSTAMP(12);  // bases known
        uint32_t tid = threadIdx.x;
        __asm__ volatile("" : "+v"(tid));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t idx = tid + (uint32_t)k * blockDim.x;
            const uint32_t kb = idx / P.S, x = idx - kb * P.S;
            if (idx < tile) buf0[x * pitch + kb] = pv[k];
        }
        JST_TSTAMP(14);  // tile committed (thread 0's share)
    } else {
        if (P.g == 0) load_tile(std::false_type{});
        else load_tile(std::true_type{});
    }
    tile_barrier<PERSIST>();
    JST_TSTAMP(1);  // tile loaded
    const uint32_t next_tile = tile_index + gridDim.x;
    const bool more = PERSIST && next_tile < tile_count;
    if constexpr (PERSIST)
        if (more) prefetch(next_tile);  // in flight through the passes and the epilogue
    const float2* src = buf0;
    uint32_t l1 = P.R1, ido = P.S;
#pragma unroll
    for (uint32_t p = P.g; p < P.nf; ++p) {
        const uint32_t ip = P.fact[p];
        ido /= ip;
        if constexpr (kTwLds > 0)
            tile_pass_any<FWD, GEN, PERSIST>(ip, buf0, twl + (P.tw_off[p] - P.tw_off[P.g]), P.S, P.cb_shift, live, pitch, ido, P.magic[p],
                                        l1 / P.R1, ido, 1u, 0u, 0u);
        else
            tile_pass_any<FWD, GEN, PERSIST>(ip, buf0, W + P.tw_off[p], P.S, P.cb_shift, live, pitch, ido, P.magic[p],
                                        l1 / P.R1, ido, 1u, 0u, 0u);
        l1 *= ip;
        JST_TSTAMP(2 + (p - P.g));  // pass done
    }
    if constexpr (is_tile_epilogue<Epi>) {
        // Multiply -> Fold.  Phase 1: every element of the tile becomes its product with h, in place (all of a
        // thread's h loads in flight before the first multiply).  Phase 2: one thread per output bin walks the bin's
        // aliases through LDS -- idx += fold is (block group + grp_step, q + dq [+ carry]) with no division.
        const uint32_t F = epi.fold, N = P.n;
        const uint32_t elems = P.S << P.cb_shift;
        const double divisor = (double)epi.decim;
        const uint32_t per_lane = P.R1 > 1 ? (F + P.R1 - 1) / P.R1 : F;
        const uint32_t total = per_lane << P.cb_shift;
        const uint32_t groups = P.CB >> P.grp_shift;
        for (uint32_t hd = 0; hd < epi.heads; ++hd) {
        const float2* hh = epi.h + (int64_t)hd * epi.h_head_stride;
        // The last head may overwrite the spectrum tile with its products (every thread multiplies, eight operand loads
        // in flight, then the walk reads finished products).  With several heads that extra pass over the tile only
        // pays when the walk has few threads to hide its operand loads behind (multi-fm.yml: 805 walkers of 1024
        // threads, products on the fly 2.0 us against 4.4 us, profiles/r04_experiments/h_tiled_timeline_8050.log).
        const bool in_place = hd + 1u == epi.heads && (epi.heads == 1u || total * 4u < blockDim.x);
        if (in_place) {
        if (epi.heads > 1) tile_barrier<PERSIST>();  // every walk of the earlier heads has read the spectrum tile
        for (uint32_t e0 = threadIdx.x; e0 < elems; e0 += 8 * blockDim.x) {
            float2 hv[8];
            uint32_t slot[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t e = e0 + (uint32_t)k * blockDim.x;
                const uint32_t ec = e < elems ? e : elems - 1u;
                const uint32_t kb = ec & (P.CB - 1u), q = ec >> P.cb_shift;
                const uint32_t idx = P.R1 > 1 ? block_of(kb) + P.R1 * q : q;
                hv[k] = hh[(int64_t)idx * epi.h_stride];
                slot[k] = (e < elems && kb < live) ? q * pitch + kb : 0xffffffffu;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (slot[k] != 0xffffffffu)
                    buf0[slot[k]] = epi.spectrum_first ? cmul_full(buf0[slot[k]], hv[k]) : cmul_full(hv[k], buf0[slot[k]]);
        }
        tile_barrier<PERSIST>();
        JST_TSTAMP(8);  // products of the last head in place
        }
        for (uint32_t e = threadIdx.x; e < total; e += blockDim.x) {
            const uint32_t kb = e & (P.CB - 1u), q = e >> P.cb_shift;
            if (kb >= live) continue;
            const uint32_t r = P.R1 > 1 ? block_of(kb) + P.R1 * q : q;
            if (r >= F) continue;
            const uint64_t t = P.R1 > 1 ? t0 : t0 + kb;
            const uint32_t off = epi.heads > 1 ? (epi.chan_offsets ? (uint32_t)(epi.chan_offsets[hd] % P.n) : epi.off)
                                               : (PERSIST ? tile_off : lane_off[kb]);
            const uint32_t off_q = off / F, off_r = off - off_q * F;
            uint32_t m = r + off_r;  // output bin whose addends are the alias class of r
            uint32_t steps = epi.decim - off_q;  // r is alias number `decim - steps` of bin m ...
            if (m >= F) { m -= F; steps -= 1u; }
            if (steps >= epi.decim) steps -= epi.decim;  // ... so the first addend (m - off) mod n is `steps` aliases on
            uint32_t idx = m + N - off;
            if (idx >= N) idx -= N;
            // position of idx in the tile: row qq, block kk_low, lane group grp (lane `within` of the group never changes)
            uint32_t qq = idx, grp = 0, kk_low = 0;
            const uint32_t within = kb & (P.grp_w - 1u);
            if (P.R1 > 1) {
                qq = idx / P.R1;
                kk_low = idx - qq * P.R1;
                grp = ((kb >> P.grp_shift) + steps * epi.grp_step) & (groups - 1u);  // groups is a power of two
            }
            double sr = 0.0, si = 0.0;
            uint32_t cur = 0;  // idx of the alias `advance` just returned (for the operand of a product formed on the fly)
            auto advance = [&]() {  // (idx, qq, grp) of the next alias; returns the LDS slot of the current one
                const uint32_t at = qq * pitch + (grp << P.grp_shift) + within;
                cur = idx;
                idx += F;
                qq += epi.dq;
                if (P.R1 > 1) {
                    kk_low += epi.dk;
                    if (kk_low >= P.R1) { kk_low -= P.R1; qq += 1u; }
                    grp += epi.grp_step;
                    if (grp >= groups) grp -= groups;
                }
                if (idx >= N) { idx -= N; qq -= epi.nq; }
                return at;
            };
            uint32_t g = 0;
            if (in_place) {
            for (; g + 5u <= epi.decim; g += 5u) {  // five LDS reads in flight, added in order
                float2 pr[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) pr[u] = src[advance()];
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    sr += (double)pr[u].x;
                    si += (double)pr[u].y;
                }
            }
            for (; g < epi.decim; ++g) {
                const float2 pr = src[advance()];
                sr += (double)pr.x;
                si += (double)pr.y;
            }
            } else {
            for (; g + 5u <= epi.decim; g += 5u) {  // five spectrum reads (LDS) and five operand loads (L2) in flight
                float2 sp[5], hv[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    sp[u] = src[advance()];
                    hv[u] = hh[(int64_t)cur * epi.h_stride];
                }
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const float2 pr = epi.spectrum_first ? cmul_full(sp[u], hv[u]) : cmul_full(hv[u], sp[u]);
                    sr += (double)pr.x;
                    si += (double)pr.y;
                }
            }
            for (; g < epi.decim; ++g) {
                const float2 sp = src[advance()];
                const float2 hv = hh[(int64_t)cur * epi.h_stride];
                const float2 pr = epi.spectrum_first ? cmul_full(sp, hv) : cmul_full(hv, sp);
                sr += (double)pr.x;
                si += (double)pr.y;
            }
            }
            epi.out[(t * epi.heads + hd) * F + m] = mk((float)(sr / divisor), (float)(si / divisor));  // 32-byte runs: plain store, merged in L2
        }
        JST_TSTAMP(9 + (hd & 1u));  // walk of head hd done (threads of wave 0)
        }
    } else
    // store result q of block (t, k) at k + R1*q: block index fastest when R1 > 1 (adjacent k are
    // adjacent in memory), q fastest for whole transforms
    if constexpr (PERSIST) {
        // The same stores with a compile-time trip count and no loop: hipcc counts the VMEM instructions issued behind the
        // prefetch only through straight-line code -- behind a loop of stores it waits for the prefetched registers with
        // vmcnt(7..0), i.e. for every store of this epilogue to be acknowledged (profiles/r06_experiments/c_tiled_persistent.log).
        constexpr uint32_t kT = threads_for((uint64_t)PS.S * PS.CB, min_threads_for_passes(PS, PS.g, PS.nf, (uint64_t)PS.S * PS.CB));
        constexpr uint32_t total = PS.S << PS.cb_shift, iters = (total + kT - 1) / kT;
        if constexpr (requires(rsrc_t rr) { epi.store_buf(rr, 0u, 0u, float2{}); } && PS.grp_w == PS.CB && kT % PS.CB == 0) {
            // dense rows (the launcher checks out_axis_stride == 1): one descriptor on the transform's output row, ONE byte
            // offset per thread -- (kb + R1 q) elements -- and a wave-uniform offset per store; with a 64-bit address per
            // store the eight unrolled epilogues ran the kernel out of its 80 VGPRs, and a scratch reload waits vmcnt(0)
            using E = std::remove_cv_t<std::remove_reference_t<decltype(epi)>>;
            const rsrc_t r_out = make_rsrc(epi.row(tile_out), P.n * E::kElemBytes);
            const uint32_t kb = threadIdx.x & (P.CB - 1u), q0 = threadIdx.x >> P.cb_shift;
            const uint32_t voff = (kb + P.R1 * q0) * E::kElemBytes;
            const float2* sp = src + (q0 * pitch + kb);
#pragma unroll
            for (uint32_t it = 0; it < iters; ++it) {
                constexpr uint32_t dq = kT >> PS.cb_shift;  // rows between two of a thread's outputs
                if (total % kT == 0 || threadIdx.x + it * kT < total) {
                    // F32 epilogues: PLAIN stores.  A workgroup's outputs are CB x 4 = 32-byte runs R1 x 4 bytes apart; four
                    // neighbouring tiles complete a 128-byte line.  Written through at agent scope (the register kernels'
                    // policy for their full-line stores) every run is a partial write to HBM and the whole kernel ran at
                    // the memory system's partial-write rate (2.8 TB/s, the prefetch arriving 8 us late); left dirty in
                    // L2 the neighbours' runs merge (JST_TILED_EPI_SC1=1: A/B)
#if !defined(JST_TILED_EPI_SC1)
                    if constexpr (requires { epi.value(float2{}); } && E::kElemBytes == 4)
#ifdef JST_TILED_DIAG_NOSTORE  // timing diagnostics only: the store goes out of the descriptor's bounds (dropped)
                        __builtin_amdgcn_raw_buffer_store_b32(f2u(epi.value(sp[it * dq * pitch])), r_out, voff + 0x40000000u,
                                                              (k0 + P.R1 * (it * dq)) * E::kElemBytes, 0);
#else
                        __builtin_amdgcn_raw_buffer_store_b32(f2u(epi.value(sp[it * dq * pitch])), r_out, voff,
                                                              (k0 + P.R1 * (it * dq)) * E::kElemBytes, 0);
#endif
                    else
#endif
                        epi.store_buf(r_out, voff, (k0 + P.R1 * (it * dq)) * E::kElemBytes, sp[it * dq * pitch]);
                }
            }
        } else {
#pragma unroll
            for (uint32_t it = 0; it < iters; ++it) {
                const uint32_t idx = threadIdx.x + it * kT;
                const uint32_t kb = idx & (P.CB - 1u), q = idx >> P.cb_shift;
                if (total % kT == 0 || idx < total)  // every lane is live (persist_eligible)
                    epi.template store<false>(tile_out, L.out_axis_stride, (int)(block_of(kb) + P.R1 * q), src[q * pitch + kb]);
            }
        }
    } else if (P.R1 > 1) {
        const uint32_t total = P.S << P.cb_shift;
        for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
            const uint32_t kb = idx & (P.CB - 1u), q = idx >> P.cb_shift;
            // (short launches keep the epilogues' agent-scope stores: with plain ones the launch ends on the write-back of
            // its dirty lines -- 16 x 65536 per cycle 23.6 against 22.2 us, run_r06g.sh; the persistent form above, which only
            // long launches take, stores plainly)
            if (kb < live)
                epi.template store<false>(lane_out[kb], L.out_axis_stride,
                                          (int)(block_of(kb) + P.R1 * q), src[q * pitch + kb]);
        }
    } else {
        for (uint32_t idx = threadIdx.x; idx < tile; idx += blockDim.x) {
            const uint32_t kb = idx / P.S, q = idx - kb * P.S;
            epi.template store<false>(lane_out[kb], L.out_axis_stride, (int)q, src[q * pitch + kb]);
        }
    }
    JST_TSTAMP(5);  // epilogue done
    if (!more) break;
    tile_barrier<PERSIST>();  // every thread is done with the tile (and the lane tables) before the next one is written
    tile_index = next_tile;
    }
    JST_TSTAMP_FLUSH();  // stores issued
}

// The blocks kernel's persistent form (fft_tile_blocks_kernel<..., PERSIST>): see there.
constexpr bool persist_eligible(const TiledPlan& p) {
    if (!JST_TILED_PERSIST || p.g == 0 || p.R1 <= 1 || block_twiddle_entries(p) == 0) return false;
    if (p.grp_w == p.CB && p.R1 % p.CB != 0) return false;  // no ragged last tile: every lane of every tile is live
    const uint64_t tile = (uint64_t)p.S * p.CB;
    return tile <= 8ull * threads_for(tile, min_threads_for_passes(p, p.g, p.nf, tile));
}
int tiled_compute_units() {
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    return cus;
}
// grid of a persistent launch: as many workgroups as the chip holds at once (occupancy x CUs), lowered to the count that
// gives every workgroup the same number of rounds (2048 tiles on 768 slots: 3 rounds -> 683 workgroups -> 688, a multiple of
// eight for the XCD-contiguous tile order)
inline uint64_t persistent_slots(const void* kernel, unsigned threads, size_t lds) {  // workgroups the chip holds at once
    struct Seen { const void* kernel; unsigned threads; size_t lds; int per_cu; };
    thread_local Seen seen[8] = {};  // the occupancy query costs microseconds of host time: once per (kernel, shape) and thread
    thread_local int used = 0;
    int per_cu = 0;
    for (int i = 0; i < used; ++i)
        if (seen[i].kernel == kernel && seen[i].threads == threads && seen[i].lds == lds) per_cu = seen[i].per_cu;
    if (per_cu == 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)threads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        seen[used < 8 ? used++ : 7] = Seen{kernel, threads, lds, per_cu};
    }
    return (uint64_t)per_cu * (uint64_t)tiled_compute_units();
}
inline unsigned persistent_grid(const void* kernel, unsigned threads, size_t lds, uint64_t tiles) {
    const uint64_t slots = persistent_slots(kernel, threads, lds);
    if (tiles <= slots) return (unsigned)tiles;
    const uint64_t rounds = (tiles + slots - 1) / slots;
    uint64_t grid = (tiles + rounds - 1) / rounds;
    grid = (grid + 7) & ~7ull;
    return (unsigned)(grid < slots ? grid : slots);
}

inline bool plan_has_generic_radix(const TiledPlan& P) {
    for (uint32_t q = 0; q < P.nf; ++q)
        if (is_generic_radix(P.fact[q])) return true;
    return false;
}

template <bool FWD, class Pro, class Epi, int SP, bool GEN, bool kPersist>
hipError_t launch_tiled_blocks(const TiledPlan& P, const FftLayout& L, const float2* W, const Pro& pro,
                               const Epi& epi, float2* scratch, hipStream_t s);

template <bool FWD, class Pro, class Epi, int SP = 0, bool GEN = false>
hipError_t launch_tiled_sp(const TiledPlan& P, const FftLayout& L, const float2* W, const Pro& pro,
                           const Epi& epi, float2* scratch, hipStream_t s) {
    if (L.transforms == 0) return hipSuccess;
    if constexpr (SP == 0 && !GEN)
        if (plan_has_generic_radix(P)) return launch_tiled_sp<FWD, Pro, Epi, 0, true>(P, L, W, pro, epi, scratch, s);
    (void)hipGetLastError();
    bool columns_done = false;
    if constexpr (SP > 0 && !GEN && columns_pipe_eligible(static_plan(SP)) && requires { pro.row(0); pro.operand_row(); }) {
        // the persistent columns kernel: dense rows and at least three transforms per workgroup (fewer: the start-up -- twiddles,
        // window taps, the first tile -- is not paid back; config 5's single stream per cycle keeps one workgroup per tile)
        if (!scratch) return hipErrorInvalidValue;
        bool dense = L.in_axis_stride == 1;
        if constexpr (requires { pro.wstride; }) dense = dense && pro.wstride == 1;
        auto kp = fft_tile_columns_pipe_kernel<FWD, Pro, SP>;
        constexpr unsigned threads_p = columns_pipe_threads<SP>();
        constexpr size_t lds_p = (size_t)static_plan(SP).R1 * static_plan(SP).CA * sizeof(float2) * (Pro::kHasOperand ? 2 : 1);
        const uint64_t tiles_per_t = (P.S + P.CA - 1) / P.CA;
        if (dense && L.transforms < 0x7fffffffull) {
            const hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(kp), (int)(kTileElems * sizeof(float2)));
            if (e != hipSuccess) return e;
            const uint64_t slots = persistent_slots(reinterpret_cast<const void*>(kp), threads_p, lds_p);
            uint64_t lanes = slots / tiles_per_t;
            if (lanes >= 1 && L.transforms >= 3 * lanes) {
                const uint64_t rounds = (L.transforms + lanes - 1) / lanes;
                lanes = (L.transforms + rounds - 1) / rounds;  // the fewest lanes that keep the number of rounds
                hipLaunchKernelGGL(kp, dim3((unsigned)(lanes * tiles_per_t)), dim3(threads_p), lds_p, s, L, W, pro, scratch);
                columns_done = true;
            }
        }
    }
    if (P.g > 0 && !columns_done) {
        if (!scratch) return hipErrorInvalidValue;
        const size_t lds_a = (size_t)P.R1 * P.CA * sizeof(float2);
        auto ka = fft_tile_columns_kernel<FWD, Pro, SP, GEN>;
        {  // tiles never exceed kTileElems
            const hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(ka), (int)(kTileElems * sizeof(float2)));
            if (e != hipSuccess) return e;
        }
        const uint64_t blocks = L.transforms * ((P.S + P.CA - 1) / P.CA);
        if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
        hipLaunchKernelGGL(ka, dim3((unsigned)blocks), dim3(threads_for((uint64_t)P.R1 * P.CA, min_threads_for_passes(P, 0, P.g, (uint64_t)P.R1 * P.CA))), lds_a, s, L, P, W,
                           pro, scratch);
    }
    // The persistent form for LONG launches of the plain epilogues only.  Measured, same box, alternating (tools/ubench/
    // run_r06f.sh, profiles/r06_experiments/c_tiled_persistent.log): config 5 cycle-batched (4096 / 8192 tiles on 768 slots) 10.1-10.4 against
    // 10.9 us and 77-78 against 80-81 us per cycle; one launch per cycle (256 / 2048 tiles) 23.5 against 22.3 and 79 against
    // 77.5 us -- below four rounds the start-up (every workgroup's first tile arrives together) costs more than the
    // prefetch saves --; the fold epilogue's own loads and store loop put vmcnt(0) in front of every commit (config 3:
    // 0.225 against 0.205 ms), and rewritten without loops or branches around its VMEM instructions, operand in two halves,
    // thread index opaque (80 VGPRs, no scratch) it ran 89.8 us against 86.3 for one workgroup per tile: two 768-thread
    // workgroups per CU either way, and the kernel is within 2x of its VALU floor (33 M wavefront instructions).  It
    // addresses dense output rows.
    constexpr bool kPersist = SP > 0 && !GEN && !is_tile_epilogue<Epi> && persist_eligible(static_plan(SP));
    if constexpr (kPersist) {
        const uint64_t tiles = L.transforms * ((P.R1 + P.CB - 1) / P.CB);
        if (L.out_axis_stride == 1 && tiles >= 4ull * 3ull * (uint64_t)tiled_compute_units())
            return launch_tiled_blocks<FWD, Pro, Epi, SP, GEN, true>(P, L, W, pro, epi, scratch, s);
    }
    return launch_tiled_blocks<FWD, Pro, Epi, SP, GEN, false>(P, L, W, pro, epi, scratch, s);
}

template <bool FWD, class Pro, class Epi, int SP, bool GEN, bool kPersist>
hipError_t launch_tiled_blocks(const TiledPlan& P, const FftLayout& L, const float2* W, const Pro& pro,
                               const Epi& epi, float2* scratch, hipStream_t s) {
    size_t lds_b = (size_t)P.S * (P.CB | 1u) * sizeof(float2);
    if constexpr (SP > 0 && !GEN) lds_b += (size_t)block_twiddle_entries(static_plan(SP)) * sizeof(float2);
    auto kb = fft_tile_blocks_kernel<FWD, Pro, Epi, SP, GEN, kPersist>;
    {  // pitch CB|1 adds at most one lane of padding per row
        const hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(kb), (int)(2 * kTileElems * sizeof(float2)));
        if (e != hipSuccess) return e;
    }
    if (lds_b > 2 * kTileElems * sizeof(float2)) return hipErrorInvalidValue;
    const uint64_t blocks = P.R1 > 1 ? L.transforms * ((P.R1 + P.CB - 1) / P.CB)
                                     : (L.transforms + P.CB - 1) / P.CB;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    const unsigned threads_b = threads_for((uint64_t)P.S * P.CB, min_threads_for_passes(P, P.g, P.nf, (uint64_t)P.S * P.CB));
    const unsigned grid_b = kPersist ? persistent_grid(reinterpret_cast<const void*>(kb), threads_b, lds_b, blocks) : (unsigned)blocks;
    hipLaunchKernelGGL(kb, dim3(grid_b), dim3(threads_b), lds_b, s, L, P, W, pro, epi, (const float2*)scratch, (uint32_t)blocks);
    return hipGetLastError();
}

// JST_TILED_STATIC=0: A/B switch, always the generic kernels
inline bool static_plans_enabled() {
    static const bool on = [] { const char* e = getenv("JST_TILED_STATIC"); return !(e && e[0] == '0'); }();
    return on;
}

// WANT...: the specialisations this call site is compiled for.  The first whose constant plan equals the run-time
// plan is taken; none: the generic kernels.
template <bool FWD, class Pro, class Epi, int... WANT>
hipError_t launch_tiled(const TiledPlan& P, const FftLayout& L, const float2* W, const Pro& pro,
                        const Epi& epi, float2* scratch, hipStream_t s) {
    if constexpr (sizeof...(WANT) > 0) {
        if (static_plans_enabled()) {
            hipError_t result = hipSuccess;
            bool taken = false;
            auto attempt = [&](auto id) {
                constexpr int SPI = decltype(id)::value;
                constexpr TiledPlan SPl = static_plan(SPI);
                // the constant plan is the run-time plan field for field; the transform count itself only sizes the grid
                // of the two-kernel form (R1 > 1), so a whole multiple of the plan's count runs the same kernels
                const bool count_ok = L.transforms == kStaticPlans[SPI].transforms ||
                                      (P.R1 > 1 && kStaticPlans[SPI].fold == 0 && L.transforms % kStaticPlans[SPI].transforms == 0);
                if (!taken && count_ok && same_plan(P, SPl)) {
                    taken = true;
                    result = launch_tiled_sp<FWD, Pro, Epi, SPI>(P, L, W, pro, epi, scratch, s);
                }
            };
            (attempt(std::integral_constant<int, WANT>{}), ...);
            if (taken) return result;
        }
    }
    return launch_tiled_sp<FWD, Pro, Epi, 0>(P, L, W, pro, epi, scratch, s);
}

template <class Pro, class Epi>
hipError_t dispatch_dir(bool forward, const TiledPlan& P, const FftLayout& L, const float2* W,
                        const Pro& pro, const Epi& epi, float2* scratch, hipStream_t s) {
    return forward ? launch_tiled<true>(P, L, W, pro, epi, scratch, s)
                   : launch_tiled<false>(P, L, W, pro, epi, scratch, s);
}

}  // namespace

// Per-pass twiddle table for the tiled kernels: pass p occupies (ip-1)*ido entries laid out
// [c-1][i] with value W[c * l1 * i] (i = 0 is present but never read); a generic-radix pass is followed by its
// wal[0..ip-1] = W[m * n/ip] (pass_table_entries).  Entry count / filler.
uint64_t fft_pass_twiddle_count(uint64_t n) {
    uint32_t fact[64];
    const int nf = fft_plan_factors(n, fact);
    uint64_t total = 0, l1 = 1;
    for (int q = 0; q < nf; ++q) {
        total += pass_table_entries(fact[q], n / (l1 * fact[q]));
        l1 *= fact[q];
    }
    return total;
}
void fft_pass_twiddle_fill(uint64_t n, const float* w_interleaved, float* out_interleaved) {
    uint32_t fact[64];
    const int nf = fft_plan_factors(n, fact);
    uint64_t off = 0, l1 = 1;
    for (int q = 0; q < nf; ++q) {
        const uint64_t ip = fact[q], ido = n / (l1 * ip);
        for (uint64_t c = 1; c < ip; ++c)
            for (uint64_t i = 0; i < ido; ++i) {
                const uint64_t src = c * l1 * i, dst = off + (c - 1) * ido + i;
                out_interleaved[2 * dst] = w_interleaved[2 * src];
                out_interleaved[2 * dst + 1] = w_interleaved[2 * src + 1];
            }
        off += (ip - 1) * ido;
        if (is_generic_radix((uint32_t)ip))
            for (uint64_t m = 0; m < ip; ++m, ++off) {
                out_interleaved[2 * off] = w_interleaved[2 * (m * (n / ip))];
                out_interleaved[2 * off + 1] = w_interleaved[2 * (m * (n / ip)) + 1];
            }
        l1 *= ip;
    }
}

bool fft_tiled_supported(uint64_t n) {
    TiledPlan p;
    return make_tiled_plan(n, 1, p);
}
bool fft_tiled_needs_scratch(uint64_t n) {  // two kernels whatever the transform count
    TiledPlan p;
    return make_tiled_plan(n, 1, p, false) && p.g > 0;
}
bool fft_tiled_may_use_scratch(uint64_t n) {  // ... or for a handful of transforms only (build_tiled_plan: split_small)
    TiledPlan p;
    return make_tiled_plan(n, 1, p, true) && p.g > 0;
}

hipError_t launch_fft_c2c_tiled(uint64_t n, bool forward, const FftLayout& L, const float2* W,
                                const float2* in, float2* out, float2* scratch, hipStream_t s) {
    TiledPlan p;
    if (!make_tiled_plan(n, L.transforms, p)) return hipErrorInvalidValue;
    return dispatch_dir(forward, p, L, W, LoadCF32{in}, StoreCF32{out}, scratch, s);
}

// Multiply (broadcast window) -> forward FFT with the product formed in the transform's first load and the spectrum
// stored as it is: the spectrum_engine block's chain when something other than Amplitude reads the spectrum (its AGC).
hipError_t launch_fft_c2c_tiled_windowed(uint64_t n, const FftLayout& L, const float2* W, const float2* in,
                                         const float2* window, int64_t window_stride, float2* out, float2* scratch,
                                         hipStream_t s) {
    TiledPlan p;
    if (!make_tiled_plan(n, L.transforms, p)) return hipErrorInvalidValue;
    return launch_tiled<true>(p, L, W, LoadCF32TimesWindow{in, window, window_stride}, StoreCF32{out}, scratch, s);
}

hipError_t launch_fft_c2c_tiled_padded(uint64_t n, uint64_t valid, bool forward, const FftLayout& L,
                                       const float2* W, const float2* in, float2* out,
                                       float2* scratch, hipStream_t s) {
    TiledPlan p;
    if (valid > n || !make_tiled_plan(n, L.transforms, p)) return hipErrorInvalidValue;
    return dispatch_dir(forward, p, L, W, LoadCF32Padded{in, (uint32_t)valid}, StoreCF32{out}, scratch, s);
}

// The plan of a transform with the fold epilogue.  Lengths beyond a tile: the balanced split, as everywhere.  A length that
// fits one tile with only a handful of transforms (the reference's multi-fm.yml: 8 x 8050 points folded to 805 -- eight
// workgroups, 29.7 us): the split and the block lanes are SEARCHED for a pair the fold's alias orbits allow
// (plan_fold_groups: 8050 = 10 x 805 with two blocks per workgroup, 40 block workgroups), the one with the most block
// workgroups -- up to 128, then the largest tile -- wins; none: one kernel.
bool make_tiled_fold_plan(uint64_t n, uint64_t transforms, uint64_t fold, TiledPlan& p) {
    if (small_split_enabled() && generic_radix_tiles_enabled() && n <= kTileElems && n >= 4096 && transforms != 0 && transforms <= 32) {
        uint32_t fact[64] = {};
        const int nf = plan_factors_ce(n, fact);
        uint64_t best_groups = 0;
        TiledPlan best{};
        for (int g = 1; g < nf; ++g)
            for (uint32_t cb = 2; cb <= 32; cb *= 2) {
                TiledPlan q{};
                if (!build_tiled_plan(n, transforms, 0, cb, q, false, true, (uint32_t)g) || !plan_fold_groups(q, fold)) continue;
                uint64_t groups = transforms * ((q.R1 + q.CB - 1) / q.CB);
                if (groups <= transforms) continue;  // no more workgroups than the one-kernel form has
                if (groups > 128) groups = 128;      // enough to fill the chip: beyond that, the larger tile (later, larger cb)
                if (groups >= best_groups) {
                    best_groups = groups;
                    best = q;
                }
            }
        if (best_groups) {
            p = best;
            return true;
        }
    }
    return make_tiled_plan(n, transforms, p, false) && plan_fold_groups(p, fold);
}

bool fft_tiled_fold_supported(uint64_t n, uint64_t transforms, uint64_t fold) {
    TiledPlan p;
    return make_tiled_fold_plan(n, transforms, fold, p);
}

hipError_t launch_fft_c2c_tiled_padded_fold(uint64_t n, uint64_t valid, bool forward, const FftLayout& L,
                                            const float2* W, const float2* in, float2* scratch,
                                            const FoldProductArgs& f, hipStream_t s) {
    TiledPlan p;
    if (valid > n || !make_tiled_fold_plan(n, L.transforms, f.fold, p)) return hipErrorInvalidValue;
    FoldProductEpi epi{f.out, f.h, f.h_stride, (uint32_t)f.fold, (uint32_t)(n / f.fold), (uint32_t)(f.offset % n),
                       f.chan_offsets, (uint32_t)f.chan_count, (uint32_t)f.chan_div, f.spectrum_first, 0, 0, 0, 0,
                       (uint32_t)(f.heads ? f.heads : 1), f.h_head_stride};
    if (p.R1 > 1) {
        epi.dq = (uint32_t)(f.fold / p.R1);
        epi.dk = (uint32_t)(f.fold % p.R1);
        epi.nq = p.n / p.R1;
        epi.grp_step = epi.dk / p.grp_stride;  // dk is a multiple of grp_stride = gcd(R1, dk) (0 when dk == 0)
    } else {
        epi.dq = (uint32_t)f.fold;
        epi.nq = p.n;
    }
    if (forward)
        return launch_tiled<true, LoadCF32Padded, FoldProductEpi, 2, 7, 8, 9>(p, L, W, LoadCF32Padded{in, (uint32_t)valid}, epi, scratch, s);
    return dispatch_dir(forward, p, L, W, LoadCF32Padded{in, (uint32_t)valid}, epi, scratch, s);
}

hipError_t launch_fft_c2c_tiled_scaled_unpad(uint64_t n, bool forward, const FftLayout& L, const float2* W,
                                             const float2* in, float2* scratch, float2* body, float2* tail,
                                             float constant, uint64_t body_len, hipStream_t s) {
    TiledPlan p;
    if (body_len > n || !make_tiled_plan(n, L.transforms, p)) return hipErrorInvalidValue;
    FftLayout T = L;  // output side: `base` = transform index
    int64_t stride = 1;
    for (int a = T.outer_rank - 1; a >= 0; --a) {
        T.out_outer_stride[a] = stride;
        stride *= (int64_t)T.outer_shape[a];
    }
    T.out_offset = 0;
    T.out_axis_stride = 0;
    const StoreScaledUnpad unpad{body, tail, constant, (uint32_t)body_len, (uint32_t)(n - body_len)};
    if (!forward) return launch_tiled<false, LoadCF32, StoreScaledUnpad, 3>(p, T, W, LoadCF32{in}, unpad, scratch, s);
    return dispatch_dir(forward, p, T, W, LoadCF32{in}, unpad, scratch, s);
}

hipError_t launch_fft_c2c_tiled_scaled_phase_unpad(uint64_t n, const FftLayout& L, const float2* W, const float2* in,
                                                   float2* scratch, float2* body, float2* tail, float constant,
                                                   uint64_t body_len, const float2* corr, uint64_t batches,
                                                   uint64_t batch_div, uint64_t channels, uint64_t chan_div, hipStream_t s) {
    TiledPlan p;
    if (body_len > n || !make_tiled_plan(n, L.transforms, p) || L.transforms >> 32 || batch_div == 0 || chan_div == 0)
        return hipErrorInvalidValue;
    FftLayout T = L;  // output side: `base` = transform index (as launch_fft_c2c_tiled_scaled_unpad)
    int64_t stride = 1;
    for (int a = T.outer_rank - 1; a >= 0; --a) {
        T.out_outer_stride[a] = stride;
        stride *= (int64_t)T.outer_shape[a];
    }
    T.out_offset = 0;
    T.out_axis_stride = 0;
    const StoreScaledPhaseUnpad epi{body, tail, constant, (uint32_t)body_len, (uint32_t)(n - body_len), corr,
                                    (uint32_t)batches, (uint32_t)batch_div, (uint32_t)channels, (uint32_t)chan_div};
    return launch_tiled<false>(p, T, W, LoadCF32{in}, epi, scratch, s);  // the Filter's inverse transform only
}

hipError_t launch_spectrum_fused_tiled(uint64_t n, const FftLayout& L, const float2* W,
                                       const float2* in, const float2* window,
                                       int64_t window_stride, float* out, float amp_coeff,
                                       bool with_range, float range_scale, float range_offset,
                                       bool fast, float guard_h0, float guard_h1, float2* scratch,
                                       hipStream_t s) {
    TiledPlan p;
    if (!make_tiled_plan(n, L.transforms, p)) return hipErrorInvalidValue;
    const LoadCF32TimesWindow pro{in, window, window_stride};
    if (with_range) {
        if (fast)  // config 5's plan as a constant for provider fast too (round 6: the lean epilogue, 17 instead of ~105 VALU per output)
            return launch_tiled<true, LoadCF32TimesWindow, StoreAmplitudeRangeT<true>, 4, 10, 11, 12>(
                p, L, W, pro, StoreAmplitudeRangeT<true>{out, amp_coeff, range_scale, range_offset, dev::BinGuard{guard_h0, guard_h1}},
                scratch, s);
        return launch_tiled<true, LoadCF32TimesWindow, StoreAmplitudeRangeT<false>, 1, 4, 5, 6, 10, 11, 12>(
            p, L, W, pro, StoreAmplitudeRangeT<false>{out, amp_coeff, range_scale, range_offset, dev::BinGuard{}}, scratch, s);
    }
    if (fast) return launch_tiled<true>(p, L, W, pro, StoreAmplitudeT<true>{out, amp_coeff}, scratch, s);
    return launch_tiled<true>(p, L, W, pro, StoreAmplitudeT<false>{out, amp_coeff}, scratch, s);
}

}  // namespace jst::kernels
