// fft_quad.hh -- the fused 4096-point spectrum kernel with ONE wavefront per SIMD and transform (round 5).
//
// Same arithmetic as fft_pipe_kernel (fft_lds.hh): pocketfft's plan 8 x 8 x 8 x 8 (pocketfft.hh:1476-1497), its radix-8
// butterfly and twiddle rule (:1141-1225), operation for operation -- every output bit is the pipelined kernel's
// (tools/ubench/quad_bench.hip compares all values and index bytes of a 16384-transform launch).  What changes is who
// computes what, where the exchanges live, and how the input arrives:
//
//   * a transform is owned by 256 threads = FOUR wavefronts, 16 points (two radix-8 butterflies per pass) per thread.
//     A workgroup's wavefronts go to the four SIMDs of a CU one each, so no two wavefronts that meet at a barrier share a
//     VALU arbiter.  The 512-thread kernel put wavefronts w and w + 4 of a workgroup on one SIMD: through the barrier-free
//     stretch (last pass, epilogue, next transform's first pass: 3 of 5.3 us) the arbiter's oldest-first rule lets the
//     older one run ahead, and barrier 0 then waits ~1.0 us for the younger (profiles/r04_experiments/q_...log: 26 % of a
//     transform in barrier waits).  Here the four wavefronts of a SIMD belong to four different workgroups in four
//     different phases, and the barriers behind the passes cost 0.5 % of a transform each
//     (profiles/r05_experiments/c_quad_timeline.log);
//   * FOUR workgroups per CU (the same 16 wavefronts, <= 128 VGPRs) need the exchange in ONE buffer of 36 KiB: the passes
//     run IN PLACE.  A pass's butterfly reads eight slots and writes its eight results back into the same slots, so a
//     slot is only ever touched by one thread per pass: one barrier per exchange (behind the writes) and none between a
//     pass's reads and its writes.  In place the data ends up digit-reversed; nothing is moved for that -- every pass just
//     addresses the slots where its operands are.  With n = (n3 n2 n1 n0) in base 8, input element n (pocketfft's CC of
//     pass 0) sits at slot  phys(n) = n0 + 8 n1 + 72 n2 + 570 n3,  and
//         pass 0 (ido 512) butterfly i                -> slots  phys(i) + 570 b              b = 0..7
//         pass 1 (ido 64)  butterfly (i, k)           -> slots  i + 72 b + 570 k
//         pass 2 (ido 8)   butterfly (i, k_lo, k_hi)  -> slots  i + 8 b + 72 k_hi + 570 k_lo
//         pass 3 (ido 1)   butterfly k = (k2 k1 k0)   -> slots  b + 8 k2 + 72 k1 + 570 k0    (to the epilogue)
//     (tools/quad_index_model.py replays the chain against numpy.fft).  Strides 8 and 72 = 8 mod 32 keep passes 0-2
//     conflict-free under gfx950's rules (ds_read_b64: two groups of 32 lanes on 64 banks, slot mod 32 distinct;
//     ds_write_b64: four groups of 16 lanes on 32 banks, slot mod 16 distinct); 570 (EVEN, see below) costs the pass-3
//     reads a two-way conflict: +32 LDS cycles per wavefront and transform on an LDS that is ~35 % busy.  All offsets
//     inside a butterfly are compile-time immediates;
//   * the next transform's rows arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`), straight into the slots pass 0 reads
//     them from -- no staging registers.  A register prefetch of 16 rows per thread (32 VGPRs) on top of the resident
//     pass-0 twiddles (28) and window taps (16) does not fit the 128 VGPRs four workgroups per CU allow (first version of
//     this file: 18 VGPRs of scratch).  LDS-DMA writes base + 16 x lane, 1 KiB per wavefront instruction: all slot strides
//     are even, so a lane's two elements are a 16-byte aligned pair of neighbours n, n + 1; the pad slots between runs take
//     zeros (their lanes' offsets point past the descriptor's records).  Pass 0 is then an in-place pass like the others,
//     the Multiply with the window applied on the way in;
//   * the 36 pieces of transform n + 1 may be issued once every wavefront has read its pass-3 operands of transform n: all
//     16 per thread are read up front, then one barrier the four wavefronts reach in step.  The pieces are spread over the
//     first outputs of the epilogue (issued back to back they held the wavefront for 18 % of a transform:
//     c_quad_timeline.log) and have the rest of it to land; behind the epilogue a wavefront waits for its own pieces
//     (`s_waitcnt vmcnt(n)`: n = the stores issued behind the last piece; LDS-DMA and stores retire in order) and the
//     barrier at the top of the next transform publishes everyone's.  Five barriers per transform;
//   * transforms are handed out DYNAMICALLY.  With the static round robin (transform t to workgroup t mod grid) the 1024
//     workgroups of a 16384-transform launch start over 12 us (the dispatcher), run at visibly different speeds and end
//     between 132 and 171 us: 15 % of the grid x time rectangle is workgroups that have not started or have already
//     finished (profiles/r05_experiments/g_workgroup_lifetimes.log).  A workgroup takes transform `blockIdx` first and
//     every further one from a device counter (one scalar atomic per transform, issued by wavefront 0 in front of the
//     wait for its LDS-DMA pieces and handed to the other wavefronts through an LDS word behind a barrier that is there
//     anyway); the last workgroup out re-arms the counters for the next launch;
//   * everything else is the pipelined kernel's: persistent workgroups, resident window taps and pass-0 twiddles, small
//     twiddle tables in LDS, LDS-only barriers, wave priorities, the epilogue functors and their cache policies.
//
// Dense CF32 rows only, one batch axis, N = 4096 only: the launcher (fft_side.hip) falls back to fft_pipe_kernel elsewhere.
#pragma once

#include "fft_lds.hh"

namespace jst::dev {

constexpr int kQuadN = 4096;
constexpr int kQuadT = 256;
constexpr int kQS2 = 72, kQS3 = 570;                      // slot strides of digits n2, n3 (n0: 1, n1: 8)
constexpr int kQuadPieces = 36;                           // 36 x 128 slots = 4608 >= 7 * 570 + 7 * 72 + 64 = 4558
constexpr int kQuadDataElems = kQuadPieces * 128;
constexpr int kQuadTw1 = 0, kQuadTw2 = 64 * 7, kQuadTwEntries = 64 * 7 + 8 * 7;  // pass 1: W[c 8 i], pass 2: W[c 64 i]
constexpr size_t fft_quad_lds_bytes() { return (size_t)(kQuadDataElems + kQuadTwEntries) * sizeof(float2) + 16; }  // + the hand-over word
static_assert(4 * ((fft_quad_lds_bytes() + 1279) / 1280) * 1280 <= 160 * 1024, "four workgroups per CU");

#ifndef JST_QUAD_PRIO_PASS
#define JST_QUAD_PRIO_PASS 3
#endif
#ifndef JST_QUAD_PRIO_EPI
#define JST_QUAD_PRIO_EPI 0
#endif
#ifndef JST_QUAD_WAVES  // wavefronts per SIMD the register allocation is bounded for: 4 = four workgroups per CU
#define JST_QUAD_WAVES 4
#endif
#ifndef JST_QUAD_PIN  // the fast epilogue's polynomial constants pinned in VGPRs (fft_lds.hh: pin_constants); 0: left to the compiler
#define JST_QUAD_PIN 1
#endif
#ifndef JST_QUAD_STATIC_NUM  // share of a launch's rounds handed out statically (see fft_quad_body)
#define JST_QUAD_STATIC_NUM 3
#define JST_QUAD_STATIC_DEN 4
#endif
// A workgroup whose claim counter has run out tries the partner counters (see fft_quad_body).  Measured, same box, three
// alternating runs (tools/ubench/run_r05p.sh, profiles/r05_experiments/p_claim_stealing.log): 162.5-165.4 us against
// 163.6-165.2 without -- the eight shares end 6 us apart either way, and what remains of the tail is the last transform of
// every workgroup.  Off.
#ifndef JST_QUAD_STEAL
#define JST_QUAD_STEAL 0
#endif
#ifndef JST_QUAD_SPREAD  // outputs between two LDS-DMA pieces in the epilogue (0: all nine back to back in front of it)
#define JST_QUAD_SPREAD 1
#endif
#ifndef JST_QUAD_DMA_AUX  // cache policy of the pieces: the input stream is read once (2 = nt), see fft_side.hip
#define JST_QUAD_DMA_AUX JST_LOAD_AUX
#endif
// s_waitcnt vmcnt(n) alone (gfx9 encoding: vmcnt[3:0] in bits 3:0, vmcnt[5:4] in bits 15:14; expcnt 7, lgkmcnt 15 = no wait)
#define JST_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt((((n) & 15) | (((n) >> 4) << 14)) | 0x0F70)

// Phase timeline (tools/ubench/quad_bench.hip -DJST_QUAD_TIMELINE only): wavefront 0 of every workgroup sums the clock64 ticks it
// spends in each phase over all its transforms and writes the sums when the workgroup ends.
#ifdef JST_QUAD_TIMELINE
__device__ unsigned long long* jst_qtl_base;
#define JST_QSTAMP(k)                                            \
    do {                                                         \
        if (wv == 0) {                                           \
            const unsigned long long now_ = clock64();           \
            qacc[k] += now_ - qlast;                             \
            qlast = now_;                                        \
        }                                                        \
    } while (0)
#else
#define JST_QSTAMP(k) do {} while (0)
#endif

// one batch axis (what the launcher admits): row t of dense [B, N] tensors, or of a ring
__device__ __forceinline__ void quad_bases(const FftLayout& L, uint64_t t, int64_t& in_base, int64_t& out_base) {
    t = fft_ring_row(L, t);
    in_base = (int64_t)L.in_offset + (int64_t)t * L.in_outer_stride[0];
    out_base = (int64_t)L.out_offset + (int64_t)t * L.out_outer_stride[0];
}

__device__ __forceinline__ float2 quad_lds_read(const float2* p) {
    // one ds_read_b64 per element, never paired into ds_read2_b64 (half the LDS rate): see pipe_passes
    typedef const volatile __attribute__((address_space(3))) unsigned long long* lds_u64_ptr;
    const unsigned long long bits = *(lds_u64_ptr)(p);
    return __builtin_bit_cast(float2, bits);
}

// twiddles of one radix-8 butterfly, in place, lanes with i == 0 untouched (pocketfft.hh:1141-1225)
template <bool FWD>
__device__ __forceinline__ void quad_twiddle(unsigned i, float2 (&y)[8], const float2 (&w)[7]) {
    twiddle_inplace3<FWD>(i, y[1], y[2], y[3], w[0], w[1], w[2]);
    twiddle_inplace4<FWD>(i, y[4], y[5], y[6], y[7], w[3], w[4], w[5], w[6]);
}

// One piece = `buffer_load_dwordx4 voff, rsrc, 0 offen lds`: 16 bytes per lane from rsrc.base + voff to LDS address M0 + 16 x
// lane.  Inline asm, so that the waits are OURS: through `__builtin_amdgcn_raw_ptr_buffer_load_lds` hipcc tracks the pieces
// in its own vmcnt bookkeeping and, merging the loop's paths, put `s_waitcnt vmcnt(2)` in front of the top barrier -- every
// transform then waited for all but two of the previous epilogue's 32 stores.  M0 is compiler-reserved: saved and
// restored inside the statement; `s_nop 0`: an SALU write of M0 needs one wait state in front of the LDS-DMA that reads it;
// `s_nop 4` covers a descriptor SGPR a VALU instruction (v_readfirstlane) may just have written.
typedef uint32_t quad_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void quad_dma_piece(quad_v4u rsrc, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    if constexpr (JST_QUAD_DMA_AUX == 2)
        __asm__ volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen nt lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rsrc) : "memory");
    else
        __asm__ volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rsrc) : "memory");
}
__device__ __forceinline__ quad_v4u quad_row_rsrc(const void* row, uint32_t bytes) {
    const uint64_t a = (uint64_t)(uintptr_t)row;
    return quad_v4u{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}

// sched: kQuadSchedWords zeroed U32 words of device memory owned by the caller (eight claim counters and the count of finished
// workgroups, kQuadSchedStride words = 256 bytes apart), or null for the static round robin.
constexpr uint32_t kQuadSchedStride = 64, kQuadSchedWords = 9 * kQuadSchedStride;
// The next unclaimed transform, asked for by ONE wavefront of the workgroup: a SCALAR atomic (one request per wavefront, the
// answer in an SGPR, counted by lgkmcnt -- not queued behind the epilogue's stores in vmcnt order) issued in front of the
// wait for the wavefront's own LDS-DMA pieces, so that the two waits overlap; request and wait are ONE statement (between
// two statements hipcc may spill the destination SGPR -- this kernel spills ~40 -- before the answer has arrived).  Through
// `__hip_atomic_fetch_add` the answer came back behind `s_waitcnt vmcnt(0)`, i.e. behind every store of the previous
// epilogue: 173 -> 226 us per launch (profiles/r05_experiments/h_dynamic_first.log).
template <int VMCNT>
__device__ __forceinline__ uint32_t quad_claim_and_wait(uint32_t* counter) {
    uint32_t v = 1u;
    __asm__ volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt vmcnt(%2)\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(counter), "n"(VMCNT) : "memory");
    return v;
}

template <bool FWD, class Pro, class Epi>
__device__ __forceinline__ void fft_quad_body(const FftLayout& L, const float2* __restrict__ W, const Pro& pro,
                                              const Epi& epi_arg, const uint32_t bid, const uint32_t grid, uint32_t* sched) {
    constexpr int N = kQuadN, T = kQuadT;
    static_assert(Pro::kRawBytes == 8, "CF32 rows");
    constexpr int STORES = epi_has_side<Epi>() ? 2 : 1;  // VMEM instructions per output
    Epi epi = epi_arg;
#if JST_QUAD_PIN
    if constexpr (requires { epi.pin_constants_lean(); }) epi.pin_constants_lean();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    typedef __attribute__((address_space(3))) unsigned char* lds_bytes_t;
    float2* twl = reinterpret_cast<float2*>(smem_raw);
    float2* data = twl + kQuadTwEntries;  // 4032 bytes in: 16-byte aligned
    volatile uint32_t* handover = reinterpret_cast<volatile uint32_t*>(data + kQuadDataElems);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- per-workgroup constants -------------------------------------------------------------------------------------
    float2 tw0[2][7];  // pass 0: W[c * i], i = tid + 256 j
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 1; c < 8; ++c) tw0[j][c - 1] = W[(unsigned)c * (unsigned)(tid + T * j)];
    float2 twv[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int g = tid + q * T;
        unsigned widx = 0;
        if (g < kQuadTw2) widx = (unsigned)(g % 7 + 1) * 8u * (unsigned)(g / 7);
        else if (g < kQuadTwEntries) widx = (unsigned)((g - kQuadTw2) % 7 + 1) * 64u * (unsigned)((g - kQuadTw2) / 7);
        twv[q] = W[widx];
    }
    uint32_t t = bid;
    const uint32_t total = (uint32_t)L.transforms;  // < 2^31 (the launcher checks)
    if (t >= total) return;
    // the transform after this one: static round robin, or the next one nobody has taken (wavefront 0 asks, see below)
    // The first `srounds` rounds are the static round robin, the rest is claimed one transform at a time -- from EIGHT counters
    // (workgroup b uses counter b mod 8, which hands out the transforms = b mod 8 of the claimed region; the dispatcher
    // deals workgroups to the eight XCDs in that order, so a counter's clients are one XCD's).  Atomics on one address are
    // served ~25 ns apart for the whole device: with one counter and 1024 workgroups claiming every ~10 us the requests
    // queued (all sixteen rounds claimed: 225 us per launch instead of 172; a 1024-transform launch 45 us instead of 18 --
    // h_dynamic_first.log, i_hybrid_one_counter.log); what the claims are for is the END of a launch, where the
    // workgroups' speeds have drifted apart.
    const uint32_t srounds = sched ? (total / grid) * JST_QUAD_STATIC_NUM / JST_QUAD_STATIC_DEN : 0xffffffffu;
    uint32_t rnd = 0;  // round of the transform in hand
    uint32_t fetched = t + grid;
    int64_t in_base, out_base;
    quad_bases(L, t, in_base, out_base);
    float2 opnd[16];  // element e = 8 j + b of this thread: position (tid + 256 j) + 512 b
    if constexpr (Pro::kHasOperand) {
        const rsrc_t r_opnd = make_rsrc(pro.operand_row(), (uint32_t)N * 8u);
#pragma unroll
        for (int e = 0; e < 16; ++e) opnd[e] = buf_load_f2(r_opnd, (uint32_t)tid * 8u, (uint32_t)(T * (e >> 3) + 512 * (e & 7)) * 8u);
    }
    // this lane's source byte offset in each of the wavefront's nine pieces (piece q = wv + 4 m fills slots [128 q, 128 q + 128))
    uint32_t src[9];
#pragma unroll
    for (int m = 0; m < 9; ++m) {
        const uint32_t e = 128u * (uint32_t)(wv + 4 * m) + 2u * (uint32_t)lane;
        const uint32_t n3 = e / (uint32_t)kQS3, r = e - n3 * (uint32_t)kQS3;
        const uint32_t n2 = r / (uint32_t)kQS2, rr = r - n2 * (uint32_t)kQS2;
        src[m] = (rr < 64u && n3 < 8u) ? (rr + 64u * n2 + 512u * n3) * 8u : 0x80000000u;  // past the records: zeros
    }
    const uint32_t lds_data = (uint32_t)(uintptr_t)((lds_bytes_t)smem_raw) + (uint32_t)kQuadTwEntries * 8u + 1024u * (uint32_t)wv;
    {
        const quad_v4u r_in = quad_row_rsrc(pro.row(in_base), (uint32_t)N * 8u);
#pragma unroll
        for (int m = 0; m < 9; ++m) quad_dma_piece(r_in, src[m], lds_data + 4096u * (uint32_t)m);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (tid + q * T < kQuadTwEntries) twl[tid + q * T] = twv[q];

    // The passes' slot and table addresses are formed where they are used, from the lane index and wave-uniform terms (an
    // empty asm makes the wavefront index opaque per transform): hoisted out of the loop they are six VGPRs that live
    // through the whole transform, and at 128 VGPRs two of the piece offsets then went to scratch -- with `s_waitcnt
    // vmcnt(0)` in front of their reloads, in the middle of the epilogue's stores.
    const bool young = bid >= (grid >> 1);
    // the first transform's pieces: everything issued so far (window taps, twiddles, pieces, the counter's answer) has landed
    uint32_t home = bid & 7u;  // the claim counter this workgroup asks (its own share first)
    if (wv == 0 && 1u >= srounds) fetched = srounds * grid + 8u * quad_claim_and_wait<0>(sched + kQuadSchedStride * home) + home;
    else JST_WAIT_VMCNT(0);
    if (tid == 0) *handover = fetched;
#ifdef JST_QUAD_TIMELINE
    unsigned long long qacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, qlast = clock64(), qiters = 0;
    const unsigned long long qwall0 = wall_clock64();
#endif

    while (true) {
        int wq = wv;
        __asm__ volatile("" : "+s"(wq));
        float2* p0 = data + lane + kQS2 * wq;                                       // pass 0: + 288 j + 570 b
        float2* p1 = data + lane + kQS3 * wq;                                       // pass 1: + 2280 j + 72 b
        float2* p2 = data + (lane & 7) + kQS2 * (lane >> 3) + kQS3 * wq;            // pass 2: + 2280 j + 8 b
        const float2* p3 = data + kQS3 * (lane & 7) + kQS2 * (lane >> 3) + 8 * wq;  // pass 3: + 32 j + b
        const float2* t1 = twl + kQuadTw1 + lane * 7;
        const float2* t2 = twl + kQuadTw2 + (lane & 7) * 7;
        // ---- this wavefront's pieces have landed (waited for behind the epilogue); the barrier publishes everyone's ---------
        __builtin_amdgcn_s_setprio(JST_QUAD_PRIO_PASS);
        lds_barrier();
        JST_QSTAMP(0);  // top barrier
        // which transform comes next (wavefront 0 left it in the hand-over word in front of the barrier), and -- wavefront 0, a
        // whole transform ahead -- which one after that
        const uint32_t tn = (uint32_t)__builtin_amdgcn_readfirstlane((int)*handover);
        const bool more = tn < total;
        int64_t nin = 0, nout = 0;
        quad_bases(L, more ? tn : t, nin, nout);
        fetched = tn + grid;
        // ---- pass 0: in place, Multiply applied on the way in ---------------------------------------------------------------
        // (all passes: BOTH butterflies' operands are requested before the first is computed -- their slots are disjoint, and a
        // pass is then one LDS round trip deep instead of two)
        {
            float2 y[2][8];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int b = 0; b < 8; ++b) y[j][b] = quad_lds_read(p0 + 4 * kQS2 * j + kQS3 * b);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float2* s = p0 + 4 * kQS2 * j;
#pragma unroll
                for (int b = 0; b < 8; ++b) y[j][b] = pro.apply(y[j][b], opnd[8 * j + b]);
                butterfly<8, FWD>(y[j]);
                quad_twiddle<FWD>((unsigned)(tid + T * j), y[j], tw0[j]);
#pragma unroll
                for (int c = 0; c < 8; ++c) s[kQS3 * c] = y[j][c];
            }
        }
        JST_QSTAMP(1);  // pass 0
        lds_barrier();
        JST_QSTAMP(2);  // barrier 0
        // ---- pass 1 ---------------------------------------------------------------------------------------------------------
        {
            float2 y[2][8], w[7];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int b = 0; b < 8; ++b) y[j][b] = quad_lds_read(p1 + 4 * kQS3 * j + kQS2 * b);
#pragma unroll
            for (int c = 0; c < 7; ++c) w[c] = t1[c];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float2* s = p1 + 4 * kQS3 * j;
                butterfly<8, FWD>(y[j]);
                quad_twiddle<FWD>((unsigned)lane, y[j], w);
#pragma unroll
                for (int c = 0; c < 8; ++c) s[kQS2 * c] = y[j][c];
            }
        }
        JST_QSTAMP(3);  // pass 1
        lds_barrier();
        JST_QSTAMP(4);  // barrier 1
        // ---- pass 2 ---------------------------------------------------------------------------------------------------------
        {
            float2 y[2][8], w[7];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int b = 0; b < 8; ++b) y[j][b] = quad_lds_read(p2 + 4 * kQS3 * j + 8 * b);
#pragma unroll
            for (int c = 0; c < 7; ++c) w[c] = t2[c];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float2* s = p2 + 4 * kQS3 * j;
                butterfly<8, FWD>(y[j]);
                quad_twiddle<FWD>((unsigned)(lane & 7), y[j], w);
#pragma unroll
                for (int c = 0; c < 8; ++c) s[8 * c] = y[j][c];
            }
        }
        JST_QSTAMP(5);  // pass 2
        lds_barrier();
        JST_QSTAMP(6);  // barrier 2
        // ---- pass 3: all sixteen operands, then the slots are free for the next transform's rows ----------------------------
        float2 z[2][8];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int b = 0; b < 8; ++b) z[j][b] = quad_lds_read(p3 + 32 * j + b);
        lds_barrier();
        JST_QSTAMP(7);  // pass-3 reads + the barrier behind them
        const quad_v4u r_next = quad_row_rsrc(pro.row(nin), (uint32_t)N * 8u);
        if constexpr (JST_QUAD_SPREAD == 0) {
            if (more) {
#pragma unroll
                for (int m = 0; m < 9; ++m) quad_dma_piece(r_next, src[m], lds_data + 4096u * (uint32_t)m);
            }
        }
        JST_QSTAMP(8);  // pieces issued (JST_QUAD_SPREAD == 0)
        if (young) __builtin_amdgcn_s_setprio(JST_QUAD_PRIO_EPI + 1);
        else __builtin_amdgcn_s_setprio(JST_QUAD_PRIO_EPI);
        const rsrc_t r_out = make_rsrc(epi.row(out_base), (uint32_t)N * Epi::kElemBytes);
        // index bytes: column group (k >> 7) + 4 c of output k + 512 c, k = tid + 256 j: the wavefront's group of j = 0 in the
        // descriptor base, 2 j + 4 c groups of side_pitch x 128 bytes in the scalar offset
        rsrc_t r_side = r_out;
        if constexpr (epi_has_side<Epi>()) r_side = epi.side_rsrc_group(fft_ring_row(L, (uint64_t)t), (uint32_t)N, (uint32_t)(wv >> 1));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            butterfly<8, FWD>(z[j]);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint32_t voff = (uint32_t)tid * Epi::kElemBytes;
                const uint32_t soff = (uint32_t)(T * j + 512 * c) * Epi::kElemBytes;
                if constexpr (epi_has_side<Epi>()) {
                    float v;
                    uint32_t index;
                    epi.side_compute(z[j][c], v, index);
                    buf_store_f1(r_out, voff, soff, v);
                    buf_store_u8(r_side, (uint32_t)tid & 127u, (uint32_t)(T * j + 512 * c) * epi.side_pitch, index);
                } else {
                    epi.store_buf(r_out, voff, soff, z[j][c]);
                }
                __builtin_amdgcn_sched_barrier(0);  // at most two epilogues in flight (VGPRs)
                if constexpr (JST_QUAD_SPREAD > 0) {
                    // piece m behind output JST_QUAD_SPREAD * m.  Not past the last transform: a piece still in flight when the
                    // workgroup ends would land in LDS that already belongs to another workgroup.
                    const int o = 8 * j + c;  // constant after unrolling
                    if (more && o % JST_QUAD_SPREAD == 0 && o / JST_QUAD_SPREAD < 9) {
                        quad_dma_piece(r_next, src[o / JST_QUAD_SPREAD], lds_data + 4096u * (uint32_t)(o / JST_QUAD_SPREAD));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        JST_QSTAMP(9);  // pass 3 + epilogue
#ifdef JST_QUAD_TIMELINE
        ++qiters;
#endif
        if (!more) break;
        // the next transform's pieces: behind the last one in vmcnt order are exactly the stores of the outputs that followed
        // it (tools/check_quad_isa.py counts them in the ISA)
        constexpr int kAfter = JST_QUAD_SPREAD == 0 ? 16 : 16 - (8 * JST_QUAD_SPREAD + 1);
        static_assert(kAfter >= 0, "nine pieces inside sixteen outputs");
        // ... and, wavefront 0, the transform after the next (the other wavefronts read the word behind the next barrier, and read
        // it last behind the top barrier of THIS transform, four barriers ago)
        ++rnd;  // the round of transform tn; `fetched` (tn + grid so far) is for round rnd + 1
        if (wv == 0 && rnd + 1u >= srounds) {
            fetched = srounds * grid + 8u * quad_claim_and_wait<kAfter * STORES>(sched + kQuadSchedStride * home) + home;
#if JST_QUAD_STEAL
            // this counter's share is handed out: look at the partner counters (b ^ 4, then ^ 2, then ^ 1: the eight shares
            // end pairwise, then in fours, then together) and stay with the first that still has work
            for (uint32_t hop = 4u; fetched >= total && hop != 0u; hop >>= 1) {
                const uint32_t other = home ^ hop;
                const uint32_t got = srounds * grid + 8u * quad_claim_and_wait<63>(sched + kQuadSchedStride * other) + other;
                if (got < total) {
                    fetched = got;
                    home = other;
                }
            }
#endif
        } else {
            JST_WAIT_VMCNT(kAfter * STORES);
        }
        JST_QSTAMP(10);  // own pieces landed
        if (tid == 0) *handover = fetched;
        t = tn;
        out_base = nout;
    }
    if (sched && tid == 0) {
        // last workgroup out re-arms the counters (every workgroup's requests have been answered before it counts itself out)
        if (__hip_atomic_fetch_add(sched + 8 * kQuadSchedStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == grid - 1u) {
            for (int x = 0; x <= 8; ++x) __hip_atomic_store(sched + x * kQuadSchedStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#ifdef JST_QUAD_TIMELINE
    if (tid == 0) {
        for (int k = 0; k < 12; ++k) jst_qtl_base[(size_t)bid * 16 + k] = qacc[k];
        jst_qtl_base[(size_t)bid * 16 + 12] = qiters;
        jst_qtl_base[(size_t)bid * 16 + 13] = qwall0;           // 100 MHz wall clock: first transform starts
        jst_qtl_base[(size_t)bid * 16 + 14] = wall_clock64();   // last transform done
    }
#endif
}

template <bool FWD, class Pro, class Epi>
__global__ __launch_bounds__(kQuadT, JST_QUAD_WAVES) void fft_quad_kernel(const FftLayout L, const float2* __restrict__ W, const Pro pro,
                                                                        const Epi epi, uint32_t* sched) {
    fft_quad_body<FWD, Pro, Epi>(L, W, pro, epi, blockIdx.x, gridDim.x, sched);
}

}  // namespace jst::dev
