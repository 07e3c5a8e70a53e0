// floor_bisect.hip -- where do the microseconds between the streaming floor and the fused kernel's skeleton go?
//
// Round-2 left this open: tools/ubench/stream_floor.hip `rows<<<512,512>>>` moves the headline launch's traffic
// (1024 x 4096 cf32 in, f32 out) in 7.2-7.5 us, the fused kernel's skeleton (load -> window -> last pass -> trivial
// store, fused_bench variant b8) needs 13.8 us.  This harness walks from one to the other ONE INGREDIENT AT A TIME:
// the same persistent row kernel with compile-time switches for
//   RING    the input comes from a ring of 16 slots (512 MiB > Infinity Cache) instead of the same 32 MiB every launch
//   LOADK   0 flat 8-byte loads | 1 buffer-descriptor 8-byte loads (zero-record descriptor past the end, like the
//           product kernel) | 2 buffer 16-byte loads, even/odd lane pairs on two 512-byte segments (the pattern a
//           dwordx4 prologue of the pipe kernel would have) | 3 buffer 16-byte loads, contiguous
//   STOREK  0 flat 4-byte plain | 1 flat 4-byte agent scope (sc1) | 2 buffer 4-byte sc1 (the product kernel's stores)
//           | 3 buffer 16-byte sc1, quad-transposed pattern (lane q of a quad writes 4 consecutive floats of output
//           row segment q) | 4 buffer 16-byte sc1 contiguous | 5 = 3 with plain policy | 6 = 2 with plain policy
//   LDSB    dynamic LDS bytes per workgroup (73 KiB = the product kernel's two exchange buffers + twiddle table)
//   WINDOW  the Multiply operand: 8 L2 loads per thread and transform, re-requested element by element behind the stores
//   EXCH    0..3 LDS exchanges (ds_write_b64 x8 -> LDS-only barrier -> ds_read_b64 x8) with the product kernel's layouts
//   VG      a register footprint of at least VG VGPRs (the product kernel allocates 104)
// Every variant reports back-to-back launch time, the event-pair median, the spread of workgroup starts / ends
// (s_memrealtime, one extra launch) and a permutation-invariant checksum of the output bits (all variants of one
// WINDOW class must agree).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -I cyberether_amd/csrc/kernels -I cyberether_amd/csrc \
//        tools/ubench/floor_bisect.hip -o tools/ubench/bin/floor_bisect
#include "fft_lds_r03_variants.hh"  // the round-3 header with its A/B switches (the product header dropped them)

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
using namespace jst::dev;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int AUX>
__device__ __forceinline__ void buf_store_f4x(rsrc_t r, uint32_t voff, uint32_t soff, v4f v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), r, voff, soff, AUX);
}
template <int AUX>
__device__ __forceinline__ void buf_store_f1x(rsrc_t r, uint32_t voff, uint32_t soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(f2u(v), r, voff, soff, AUX);
}

template <int LOADK>
__device__ __forceinline__ void load_row(float2 (&v)[8], const float2* row, rsrc_t r, int tid) {
    if constexpr (LOADK == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = row[tid + 512 * j];
    } else if constexpr (LOADK == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = buf_load_f2(r, (uint32_t)tid * 8u, (uint32_t)(4096 * j));
    } else if constexpr (LOADK == 2) {
        const uint32_t voff = (uint32_t)(((tid >> 1) * 2 + 2048 * (tid & 1)) * 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const v4f q = buf_load_f4(r, voff, (uint32_t)(4096 * k));
            v[2 * k] = mk(q.x, q.y);
            v[2 * k + 1] = mk(q.z, q.w);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const v4f q = buf_load_f4(r, (uint32_t)tid * 16u, (uint32_t)(8192 * k));
            v[2 * k] = mk(q.x, q.y);
            v[2 * k + 1] = mk(q.z, q.w);
        }
    }
}

template <int LOADK, int STOREK, int WINDOW, int EXCH, int VG, bool TL>
__global__ __launch_bounds__(512, 2) void skel(const float2* __restrict__ in, float* __restrict__ out,
                                               const float2* __restrict__ win, int nrows,
                                               unsigned long long* __restrict__ tl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* bufA = reinterpret_cast<float2*>(smem_raw);
    float2* bufB = bufA + lds_elems(4096);
    const int tid = threadIdx.x;
    if constexpr (TL) {
        if (tid == 0) tl[2 * blockIdx.x] = wall_clock64();
    }
    if constexpr (VG > 0) {
        if constexpr (VG > 96) asm volatile("" ::: "v103");
        else if constexpr (VG > 64) asm volatile("" ::: "v79");
    }
    int r = blockIdx.x;
    if (r >= nrows) return;
    float2 cur[8], opnd[8];
    const rsrc_t r_w = make_rsrc(win, 4096u * 8u);
    {
        const rsrc_t r_in = make_rsrc(in + (size_t)r * 4096, 4096u * 8u);
        if constexpr (WINDOW) load_row<((LOADK == 0 || LOADK == 4) ? 1 : LOADK)>(opnd, win, r_w, tid);
        load_row<(LOADK == 4 ? 1 : LOADK)>(cur, in + (size_t)r * 4096, r_in, tid);
    }
    bool flip = false;
    while (true) {
        float2 x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = WINDOW ? cmul_full(cur[j], opnd[j]) : cur[j];
        const int rn = r + gridDim.x;
        const bool more = rn < nrows;
        if constexpr (LOADK == 0) {
            if (more) load_row<0>(cur, in + (size_t)rn * 4096, r_w, tid);
        } else if constexpr (LOADK == 4) {  // buffer loads under a condition (the form the product kernel avoids)
            if (more) load_row<1>(cur, nullptr, make_rsrc(in + (size_t)rn * 4096, 4096u * 8u), tid);
        } else {
            const rsrc_t r_in = make_rsrc(in + (size_t)(more ? rn : r) * 4096, more ? 4096u * 8u : 0u);
            load_row<LOADK>(cur, nullptr, r_in, tid);
        }
        // ---- LDS exchanges with the product kernel's layouts -----------------------------------
        float2* b0 = flip ? bufB : bufA;
        float2* b1 = flip ? bufA : bufB;
#pragma unroll
        for (int e = 0; e < EXCH; ++e) {
            float2* wr = b0 + pphys(tid);
#pragma unroll
            for (int c = 0; c < 8; ++c) wr[pcphys(c * 512)] = x[c];
            lds_barrier();
            const int ido = e == 0 ? 64 : (e == 1 ? 8 : 1);
            const int i = tid & (ido - 1), k = tid / ido;
            const float2* rd = b0 + pphys(i + ido * 8 * k);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                typedef const volatile __attribute__((address_space(3))) unsigned long long* lds_u64_ptr;
                const unsigned long long bits = *(lds_u64_ptr)(rd + (e == 0 ? pcphys(64 * b) : (e == 1 ? pcphys(8 * b) : pcphys(b))));
                x[b] = __builtin_bit_cast(float2, bits);
            }
            float2* t = b0; b0 = b1; b1 = t;
        }
        if constexpr (EXCH & 1) flip = !flip;
        // ---- stores --------------------------------------------------------------------------------
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = x[j].x + x[j].y;
        float* orow = out + (size_t)r * 4096;
        const rsrc_t r_out = make_rsrc(orow, 4096u * 4u);
        const rsrc_t r_wn = make_rsrc(win, more ? 4096u * 8u : 0u);
        constexpr bool REQ = WINDOW == 1;  // WINDOW == 2: the operand stays in registers, nothing is re-requested
        if constexpr (STOREK == 0 || STOREK == 1 || STOREK == 2 || STOREK == 6) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (STOREK == 0) orow[tid + 512 * j] = f[j];
                else if constexpr (STOREK == 1) __hip_atomic_store(orow + tid + 512 * j, f[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if constexpr (STOREK == 2) buf_store_f1x<16>(r_out, (uint32_t)tid * 4u, (uint32_t)(2048 * j), f[j]);
                else buf_store_f1x<0>(r_out, (uint32_t)tid * 4u, (uint32_t)(2048 * j), f[j]);
                if constexpr (REQ) {
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (LOADK == 2 || LOADK == 3) {
                        if (j & 1) {  // a 16-byte operand request behind every second output
                            const int k = j >> 1;
                            const uint32_t voff = LOADK == 2 ? (uint32_t)(((tid >> 1) * 2 + 2048 * (tid & 1)) * 8) : (uint32_t)tid * 16u;
                            const v4f q = buf_load_f4(r_wn, voff, (uint32_t)((LOADK == 2 ? 4096 : 8192) * k));
                            opnd[2 * k] = mk(q.x, q.y);
                            opnd[2 * k + 1] = mk(q.z, q.w);
                        }
                    } else {
                        opnd[j] = buf_load_f2(r_wn, (uint32_t)tid * 8u, (uint32_t)(4096 * j));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const v4f q = {f[4 * h], f[4 * h + 1], f[4 * h + 2], f[4 * h + 3]};
                const uint32_t voff = (STOREK == 4) ? (uint32_t)tid * 16u : (uint32_t)(((tid & ~3) + 512 * (tid & 3)) * 4);
                if constexpr (STOREK == 5) buf_store_f4x<0>(r_out, voff, (uint32_t)(8192 * h), q);
                else buf_store_f4x<16>(r_out, voff, (uint32_t)(8192 * h), q);
                if constexpr (REQ) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) {
                        const int k = 2 * h + k2;
                        if constexpr (LOADK == 2 || LOADK == 3) {
                            const uint32_t vo = LOADK == 2 ? (uint32_t)(((tid >> 1) * 2 + 2048 * (tid & 1)) * 8) : (uint32_t)tid * 16u;
                            const v4f w4 = buf_load_f4(r_wn, vo, (uint32_t)((LOADK == 2 ? 4096 : 8192) * k));
                            opnd[2 * k] = mk(w4.x, w4.y);
                            opnd[2 * k + 1] = mk(w4.z, w4.w);
                        } else {
                            opnd[2 * k] = buf_load_f2(r_wn, (uint32_t)tid * 8u, (uint32_t)(4096 * (2 * k)));
                            opnd[2 * k + 1] = buf_load_f2(r_wn, (uint32_t)tid * 8u, (uint32_t)(4096 * (2 * k + 1)));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (!more) break;
        r = rn;
    }
    if constexpr (TL) {
        if (tid == 0) tl[2 * blockIdx.x + 1] = wall_clock64();
    }
}

// stream_floor.hip's kernel, verbatim (the 7.2-7.5 us reference point)
__global__ __launch_bounds__(512, 2) void rows(const float2* __restrict__ in, float* __restrict__ out, int nrows) {
    int r = blockIdx.x;
    float2 cur[8], nxt[8];
    if (r < nrows)
        for (int j = 0; j < 8; ++j) cur[j] = in[(size_t)r * 4096 + threadIdx.x + 512 * j];
    for (; r < nrows; r += gridDim.x) {
        const int rn = r + gridDim.x;
        if (rn < nrows)
            for (int j = 0; j < 8; ++j) nxt[j] = in[(size_t)rn * 4096 + threadIdx.x + 512 * j];
        for (int j = 0; j < 8; ++j) out[(size_t)r * 4096 + threadIdx.x + 512 * j] = cur[j].x + cur[j].y;
        for (int j = 0; j < 8; ++j) cur[j] = nxt[j];
    }
}

__global__ void checksum_kernel(const uint32_t* v, uint64_t n, unsigned long long* acc) {
    unsigned long long s = 0, q = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long b = v[i];
        s += b;
        q += b * b + (b >> 7);
    }
    atomicAdd(&acc[0], s);
    atomicAdd(&acc[1], q);
}

struct Ctx {
    float2 *in, *win;
    float* out;
    unsigned long long *acc, *tl;
    hipStream_t st;
    int reps;
    const char* filter;
};

template <class L>
static void timeit(Ctx& c, const char* name, int grid, L launch, L launch_tl) {
    if (c.filter && !strstr(name, c.filter)) return;
    constexpr int SLOTS = 16;
    for (int i = 0; i < 20; ++i) launch(i % SLOTS);
    CK(hipStreamSynchronize(c.st));
    std::vector<hipEvent_t> ev(2 * c.reps);
    for (auto& e : ev) CK(hipEventCreate(&e));
    for (int i = 0; i < c.reps; ++i) {
        CK(hipEventRecord(ev[2 * i], c.st));
        launch(i % SLOTS);
        CK(hipEventRecord(ev[2 * i + 1], c.st));
    }
    CK(hipStreamSynchronize(c.st));
    std::vector<float> us(c.reps);
    for (int i = 0; i < c.reps; ++i) { float ms; CK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); us[i] = ms * 1e3f; }
    std::sort(us.begin(), us.end());
    for (auto& e : ev) CK(hipEventDestroy(e));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f, sum = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a, c.st));
        for (int i = 0; i < c.reps; ++i) launch(i % SLOTS);
        CK(hipEventRecord(b, c.st)); CK(hipStreamSynchronize(c.st));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms); sum += ms;
    }
    // start / end spread of the workgroups of one launch (wall clock, 100 MHz)
    CK(hipMemsetAsync(c.tl, 0, 2 * 1024 * sizeof(unsigned long long), c.st));
    launch_tl(5);
    CK(hipStreamSynchronize(c.st));
    std::vector<unsigned long long> tl(2 * 1024);
    CK(hipMemcpy(tl.data(), c.tl, tl.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0;
    for (int g = 0; g < grid; ++g) {
        if (!tl[2 * g]) continue;
        s0 = std::min(s0, tl[2 * g]); s1 = std::max(s1, tl[2 * g]);
        e0 = std::min(e0, tl[2 * g + 1]); e1 = std::max(e1, tl[2 * g + 1]);
    }
    // checksum of slot 3's output
    CK(hipMemsetAsync(c.out, 0xff, 1024ull * 4096 * 4, c.st));
    CK(hipMemsetAsync(c.acc, 0, 16, c.st));
    launch(3);
    checksum_kernel<<<1024, 256, 0, c.st>>>((const uint32_t*)c.out, 1024ull * 4096, c.acc);
    unsigned long long h[2];
    CK(hipMemcpy(h, c.acc, 16, hipMemcpyDeviceToHost));
    printf("%-44s grid %4d | b2b %6.2f us (mean %6.2f) | events median %6.2f min %6.2f | starts %5.2f us ends %5.2f us span %6.2f us | chk %016llx %016llx\n",
           name, grid, best * 1e3 / c.reps, sum * 1e3 / (3 * c.reps), us[c.reps / 2], us[0],
           s1 >= s0 ? (s1 - s0) * 0.01 : -1.0, e1 >= e0 ? (e1 - e0) * 0.01 : -1.0, e1 >= s0 ? (e1 - s0) * 0.01 : -1.0, h[0], h[1]);
    fflush(stdout);
}

template <int LOADK, int STOREK, int WINDOW, int EXCH, int VG>
static void variant(Ctx& c, const char* name, int grid, size_t lds, bool ring) {
    auto k = skel<LOADK, STOREK, WINDOW, EXCH, VG, false>;
    auto kt = skel<LOADK, STOREK, WINDOW, EXCH, VG, true>;
    size_t need = EXCH > 0 ? std::max(lds, (size_t)2 * lds_elems(4096) * sizeof(float2)) : lds;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    CK(hipFuncSetAttribute((const void*)kt, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    const size_t slot_elems = 1024ull * 4096;
    auto l = [&, k](int slot) { k<<<grid, 512, need, c.st>>>(c.in + (ring ? slot * slot_elems : 0), c.out, c.win, 1024, c.tl); };
    auto lt = [&, kt](int slot) { kt<<<grid, 512, need, c.st>>>(c.in + (ring ? slot * slot_elems : 0), c.out, c.win, 1024, c.tl); };
    std::function<void(int)> f1 = l, f2 = lt;
    timeit(c, name, grid, f1, f2);
}



int main(int argc, char** argv) {
    Ctx c{};
    c.reps = argc > 1 ? atoi(argv[1]) : 300;
    c.filter = argc > 2 ? argv[2] : nullptr;
    const size_t slot_elems = 1024ull * 4096;
    CK(hipMalloc(&c.in, 16 * slot_elems * 8));
    CK(hipMalloc(&c.win, 4096 * 8));
    CK(hipMalloc(&c.out, slot_elems * 4));
    CK(hipMalloc(&c.acc, 16));
    CK(hipMalloc(&c.tl, 2 * 1024 * 8));
    CK(hipStreamCreate(&c.st));
    {
        std::vector<float2> h(slot_elems);
        uint32_t s = 12345u;
        for (int sl = 0; sl < 16; ++sl) {
            for (auto& v : h) {
                s = s * 1664525u + 1013904223u; v.x = ((s >> 8) * (1.0f / 16777216.0f)) - 0.5f;
                s = s * 1664525u + 1013904223u; v.y = ((s >> 8) * (1.0f / 16777216.0f)) - 0.5f;
            }
            CK(hipMemcpy(c.in + sl * slot_elems, h.data(), slot_elems * 8, hipMemcpyHostToDevice));
        }
        std::vector<float2> w(4096);
        for (int i = 0; i < 4096; ++i) {
            const double t = 0.42 - 0.5 * cos(6.283185307179586 * i / 4095) + 0.08 * cos(2 * 6.283185307179586 * i / 4095);
            w[i] = make_float2((float)((i & 1) ? -t : t), 0.0f);
        }
        CK(hipMemcpy(c.win, w.data(), 4096 * 8, hipMemcpyHostToDevice));
    }
    const size_t L73 = fft_pipe_lds_bytes(4096);
    // ---- the reference point and the ring -------------------------------------------------------
    {
        std::function<void(int)> f = [&](int) { rows<<<512, 512, 0, c.st>>>(c.in, c.out, 1024); };
        std::function<void(int)> ft = f;
        timeit(c, "A0 rows (stream_floor.hip), same 32 MiB", 512, f, ft);
        std::function<void(int)> g = [&](int slot) { rows<<<512, 512, 0, c.st>>>(c.in + slot * slot_elems, c.out, 1024); };
        timeit(c, "A1 rows, ring of 16 slots", 512, g, g);
    }
    variant<0, 0, 0, 0, 0>(c, "A2 skel flat8/plain4, same 32 MiB", 512, 0, false);
    variant<0, 0, 0, 0, 0>(c, "A3 skel flat8/plain4, ring", 512, 0, true);
    // ---- stores ----------------------------------------------------------------------------------
    variant<0, 1, 0, 0, 0>(c, "B1 flat8 / flat4 sc1", 512, 0, true);
    variant<1, 6, 0, 0, 0>(c, "B2 buf8 / buf4 plain", 512, 0, true);
    variant<1, 2, 0, 0, 0>(c, "B3 buf8 / buf4 sc1 (product addressing)", 512, 0, true);
    variant<1, 3, 0, 0, 0>(c, "B4 buf8 / buf16 sc1 quad pattern", 512, 0, true);
    variant<1, 4, 0, 0, 0>(c, "B5 buf8 / buf16 sc1 contiguous", 512, 0, true);
    variant<1, 5, 0, 0, 0>(c, "B6 buf8 / buf16 plain quad pattern", 512, 0, true);
    // ---- loads -----------------------------------------------------------------------------------
    variant<2, 2, 0, 0, 0>(c, "C1 buf16 pairs / buf4 sc1", 512, 0, true);
    variant<3, 2, 0, 0, 0>(c, "C2 buf16 contiguous / buf4 sc1", 512, 0, true);
    variant<2, 3, 0, 0, 0>(c, "C3 buf16 pairs / buf16 sc1 quad", 512, 0, true);
    variant<3, 4, 0, 0, 0>(c, "C4 buf16 contiguous / buf16 sc1 contiguous", 512, 0, true);
    // ---- occupancy ingredients -------------------------------------------------------------------
    variant<1, 2, 0, 0, 0>(c, "D1 B3 + 73 KiB LDS", 512, L73, true);
    variant<1, 2, 0, 0, 104>(c, "D2 B3 + 73 KiB LDS + 104 VGPR", 512, L73, true);
    variant<1, 2, 0, 0, 104>(c, "D3 D2 grid 1024 (one row each)", 1024, L73, true);
    variant<1, 2, 0, 0, 104>(c, "D4 D2 grid 256 (four rows each)", 256, L73, true);
    variant<2, 3, 0, 0, 104>(c, "D5 C3 + 73 KiB LDS + 104 VGPR", 512, L73, true);
    // ---- window operand --------------------------------------------------------------------------
    variant<1, 2, 1, 0, 104>(c, "E1 D2 + window (8-byte L2 re-requests)", 512, L73, true);
    variant<2, 3, 1, 0, 104>(c, "E2 D5 + window (16-byte L2 re-requests)", 512, L73, true);
    variant<1, 2, 2, 0, 104>(c, "E3 D2 + window RESIDENT in VGPRs", 512, L73, true);
    variant<2, 3, 2, 0, 104>(c, "E4 D5 + window RESIDENT in VGPRs", 512, L73, true);
    variant<2, 2, 2, 0, 104>(c, "E5 buf16 pairs / buf4 sc1 + window RESIDENT", 512, L73, true);
    // ---- flat versus buffer addressing (A3 9.0 us vs B2 10.3 us) --------------------------------------
    variant<0, 6, 0, 0, 0>(c, "G1 flat8 (conditional) / buf4 plain", 512, 0, true);
    variant<1, 0, 0, 0, 0>(c, "G2 buf8 (unconditional) / flat4 plain", 512, 0, true);
    variant<4, 6, 0, 0, 0>(c, "G3 buf8 (conditional) / buf4 plain", 512, 0, true);
    variant<4, 0, 0, 0, 0>(c, "G4 buf8 (conditional) / flat4 plain", 512, 0, true);
    // ---- LDS exchanges ---------------------------------------------------------------------------
    variant<1, 2, 1, 1, 104>(c, "F1 E1 + 1 exchange", 512, L73, true);
    variant<1, 2, 1, 2, 104>(c, "F2 E1 + 2 exchanges", 512, L73, true);
    variant<1, 2, 1, 3, 104>(c, "F3 E1 + 3 exchanges", 512, L73, true);
    variant<2, 3, 1, 3, 104>(c, "F4 E2 + 3 exchanges", 512, L73, true);
    return 0;
}
