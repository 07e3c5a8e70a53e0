// rfft.hip -- real-input transforms of the FFT module: pocketfft's rfftp (FFTPACK halfcomplex
// passes radf2/3/4/5 and radb2/3/4/5, pocketfft.hh:1574-2075) for the two calls the reference
// makes with F32 input, r2r_fftpack(real2hermitian = forward, forward) and r2c
// (src/domains/dsp/fft/module_impl_native_cpu.cc:142-167), plus the generic radix radfg / radbg
// (:1753-1893, 2076-2208) for prime factors above 5.
//
// Not a hot path (the spectrum chain casts F32 to CF32 first): one launch per pass over dense
// F32[transforms][n] ping-pong buffers, one thread per work item.  A pass with (l1, ido) has, per
// k in [0, l1), one "head" item (the i = 0 outputs, plus the i = ido-1 outputs when ido is even)
// and (ido-1)/2 "pair" items (i = 2, 4, ...).  Every expression below is the reference's, one
// rounding per operation (-ffp-contract=off), so the halfcomplex output is bit-identical.
#include "device_math.hh"
#include "kernels.hh"

namespace jst::kernels {

using namespace jst::dev;

namespace {

constexpr int kBlock = 256;

#define MULPM(a, b, c, d, e, f) { a = (c) * (e) + (d) * (f); b = (c) * (f) - (d) * (e); }
#define WA(x, i) wa[(i) + (x) * (ido - 1)]

// ---- forward (real -> halfcomplex): CC(a,b,c) = cc[a + ido*(b + l1*c)], CH(a,b,c) = ch[a + ido*(b + IP*c)]
template <int IP>
__device__ __forceinline__ void radf_item(uint32_t ido, uint32_t l1, const float* __restrict__ cc,
                                          float* __restrict__ ch, const float* __restrict__ wa,
                                          uint32_t k, uint32_t item) {
#define CC(a, b, c) cc[(a) + ido * ((b) + l1 * (c))]
#define CH(a, b, c) ch[(a) + ido * ((b) + IP * (c))]
    const uint32_t i = 2 * item, ic = ido - i;
    if constexpr (IP == 2) {
        if (item == 0) {
            CH(0, 0, k) = CC(0, k, 0) + CC(0, k, 1);
            CH(ido - 1, 1, k) = CC(0, k, 0) - CC(0, k, 1);
            if ((ido & 1) == 0) {
                CH(0, 1, k) = -CC(ido - 1, k, 1);
                CH(ido - 1, 0, k) = CC(ido - 1, k, 0);
            }
            return;
        }
        float tr2, ti2;
        MULPM(tr2, ti2, WA(0, i - 2), WA(0, i - 1), CC(i - 1, k, 1), CC(i, k, 1))
        CH(i - 1, 0, k) = CC(i - 1, k, 0) + tr2;
        CH(ic - 1, 1, k) = CC(i - 1, k, 0) - tr2;
        CH(i, 0, k) = ti2 + CC(i, k, 0);
        CH(ic, 1, k) = ti2 - CC(i, k, 0);
    } else if constexpr (IP == 3) {
        constexpr float taur = -0.5f, taui = 0.8660254037844386467637231707529362f;
        if (item == 0) {
            const float cr2 = CC(0, k, 1) + CC(0, k, 2);
            CH(0, 0, k) = CC(0, k, 0) + cr2;
            CH(0, 2, k) = taui * (CC(0, k, 2) - CC(0, k, 1));
            CH(ido - 1, 1, k) = CC(0, k, 0) + taur * cr2;
            return;
        }
        float di2, di3, dr2, dr3;
        MULPM(dr2, di2, WA(0, i - 2), WA(0, i - 1), CC(i - 1, k, 1), CC(i, k, 1))
        MULPM(dr3, di3, WA(1, i - 2), WA(1, i - 1), CC(i - 1, k, 2), CC(i, k, 2))
        {
            const float t1 = dr2 + dr3, t2 = dr3 - dr2, t3 = di2 + di3, t4 = di2 - di3;
            dr2 = t1; di2 = t3; dr3 = t4; di3 = t2;
        }
        CH(i - 1, 0, k) = CC(i - 1, k, 0) + dr2;
        CH(i, 0, k) = CC(i, k, 0) + di2;
        const float tr2 = CC(i - 1, k, 0) + taur * dr2, ti2 = CC(i, k, 0) + taur * di2;
        const float tr3 = taui * dr3, ti3 = taui * di3;
        CH(i - 1, 2, k) = tr2 + tr3;
        CH(ic - 1, 1, k) = tr2 - tr3;
        CH(i, 2, k) = ti3 + ti2;
        CH(ic, 1, k) = ti3 - ti2;
    } else if constexpr (IP == 4) {
        constexpr float hsqt2 = 0.707106781186547524400844362104849f;
        if (item == 0) {
            const float tr1 = CC(0, k, 3) + CC(0, k, 1);
            CH(0, 2, k) = CC(0, k, 3) - CC(0, k, 1);
            const float tr2 = CC(0, k, 0) + CC(0, k, 2);
            CH(ido - 1, 1, k) = CC(0, k, 0) - CC(0, k, 2);
            CH(0, 0, k) = tr2 + tr1;
            CH(ido - 1, 3, k) = tr2 - tr1;
            if ((ido & 1) == 0) {
                const float ti1 = -hsqt2 * (CC(ido - 1, k, 1) + CC(ido - 1, k, 3));
                const float tq1 = hsqt2 * (CC(ido - 1, k, 1) - CC(ido - 1, k, 3));
                CH(ido - 1, 0, k) = CC(ido - 1, k, 0) + tq1;
                CH(ido - 1, 2, k) = CC(ido - 1, k, 0) - tq1;
                CH(0, 3, k) = ti1 + CC(ido - 1, k, 2);
                CH(0, 1, k) = ti1 - CC(ido - 1, k, 2);
            }
            return;
        }
        float ci2, ci3, ci4, cr2, cr3, cr4;
        MULPM(cr2, ci2, WA(0, i - 2), WA(0, i - 1), CC(i - 1, k, 1), CC(i, k, 1))
        MULPM(cr3, ci3, WA(1, i - 2), WA(1, i - 1), CC(i - 1, k, 2), CC(i, k, 2))
        MULPM(cr4, ci4, WA(2, i - 2), WA(2, i - 1), CC(i - 1, k, 3), CC(i, k, 3))
        const float tr1 = cr4 + cr2, tr4 = cr4 - cr2, ti1 = ci2 + ci4, ti4 = ci2 - ci4;
        const float tr2 = CC(i - 1, k, 0) + cr3, tr3 = CC(i - 1, k, 0) - cr3;
        const float ti2 = CC(i, k, 0) + ci3, ti3 = CC(i, k, 0) - ci3;
        CH(i - 1, 0, k) = tr2 + tr1;
        CH(ic - 1, 3, k) = tr2 - tr1;
        CH(i, 0, k) = ti1 + ti2;
        CH(ic, 3, k) = ti1 - ti2;
        CH(i - 1, 2, k) = tr3 + ti4;
        CH(ic - 1, 1, k) = tr3 - ti4;
        CH(i, 2, k) = tr4 + ti3;
        CH(ic, 1, k) = tr4 - ti3;
    } else {
        constexpr float tr11 = 0.3090169943749474241022934171828191f, ti11 = 0.9510565162951535721164393333793821f,
                        tr12 = -0.8090169943749474241022934171828191f, ti12 = 0.5877852522924731291687059546390728f;
        if (item == 0) {
            const float cr2 = CC(0, k, 4) + CC(0, k, 1), ci5 = CC(0, k, 4) - CC(0, k, 1);
            const float cr3 = CC(0, k, 3) + CC(0, k, 2), ci4 = CC(0, k, 3) - CC(0, k, 2);
            CH(0, 0, k) = CC(0, k, 0) + cr2 + cr3;
            CH(ido - 1, 1, k) = CC(0, k, 0) + tr11 * cr2 + tr12 * cr3;
            CH(0, 2, k) = ti11 * ci5 + ti12 * ci4;
            CH(ido - 1, 3, k) = CC(0, k, 0) + tr12 * cr2 + tr11 * cr3;
            CH(0, 4, k) = ti12 * ci5 - ti11 * ci4;
            return;
        }
        float di2, di3, di4, di5, dr2, dr3, dr4, dr5;
        MULPM(dr2, di2, WA(0, i - 2), WA(0, i - 1), CC(i - 1, k, 1), CC(i, k, 1))
        MULPM(dr3, di3, WA(1, i - 2), WA(1, i - 1), CC(i - 1, k, 2), CC(i, k, 2))
        MULPM(dr4, di4, WA(2, i - 2), WA(2, i - 1), CC(i - 1, k, 3), CC(i, k, 3))
        MULPM(dr5, di5, WA(3, i - 2), WA(3, i - 1), CC(i - 1, k, 4), CC(i, k, 4))
        {
            const float t1 = dr2 + dr5, t2 = dr5 - dr2, t3 = di2 + di5, t4 = di2 - di5;
            dr2 = t1; di2 = t3; dr5 = t4; di5 = t2;
        }
        {
            const float t1 = dr3 + dr4, t2 = dr4 - dr3, t3 = di3 + di4, t4 = di3 - di4;
            dr3 = t1; di3 = t3; dr4 = t4; di4 = t2;
        }
        CH(i - 1, 0, k) = CC(i - 1, k, 0) + dr2 + dr3;
        CH(i, 0, k) = CC(i, k, 0) + di2 + di3;
        const float tr2 = CC(i - 1, k, 0) + tr11 * dr2 + tr12 * dr3, ti2 = CC(i, k, 0) + tr11 * di2 + tr12 * di3;
        const float tr3 = CC(i - 1, k, 0) + tr12 * dr2 + tr11 * dr3, ti3 = CC(i, k, 0) + tr12 * di2 + tr11 * di3;
        const float tr5 = ti11 * dr5 + ti12 * dr4, ti5 = ti11 * di5 + ti12 * di4;
        const float tr4 = ti12 * dr5 - ti11 * dr4, ti4 = ti12 * di5 - ti11 * di4;
        CH(i - 1, 2, k) = tr2 + tr5;
        CH(ic - 1, 1, k) = tr2 - tr5;
        CH(i, 2, k) = ti5 + ti2;
        CH(ic, 1, k) = ti5 - ti2;
        CH(i - 1, 4, k) = tr3 + tr4;
        CH(ic - 1, 3, k) = tr3 - tr4;
        CH(i, 4, k) = ti4 + ti3;
        CH(ic, 3, k) = ti4 - ti3;
    }
#undef CC
#undef CH
}

// ---- backward (halfcomplex -> real): CC(a,b,c) = cc[a + ido*(b + IP*c)], CH(a,b,c) = ch[a + ido*(b + l1*c)]
template <int IP>
__device__ __forceinline__ void radb_item(uint32_t ido, uint32_t l1, const float* __restrict__ cc,
                                          float* __restrict__ ch, const float* __restrict__ wa,
                                          uint32_t k, uint32_t item) {
#define CC(a, b, c) cc[(a) + ido * ((b) + IP * (c))]
#define CH(a, b, c) ch[(a) + ido * ((b) + l1 * (c))]
    const uint32_t i = 2 * item, ic = ido - i;
    if constexpr (IP == 2) {
        if (item == 0) {
            CH(0, k, 0) = CC(0, 0, k) + CC(ido - 1, 1, k);
            CH(0, k, 1) = CC(0, 0, k) - CC(ido - 1, 1, k);
            if ((ido & 1) == 0) {
                CH(ido - 1, k, 0) = 2 * CC(ido - 1, 0, k);
                CH(ido - 1, k, 1) = -2 * CC(0, 1, k);
            }
            return;
        }
        CH(i - 1, k, 0) = CC(i - 1, 0, k) + CC(ic - 1, 1, k);
        const float tr2 = CC(i - 1, 0, k) - CC(ic - 1, 1, k);
        const float ti2 = CC(i, 0, k) + CC(ic, 1, k);
        CH(i, k, 0) = CC(i, 0, k) - CC(ic, 1, k);
        MULPM(CH(i, k, 1), CH(i - 1, k, 1), WA(0, i - 2), WA(0, i - 1), ti2, tr2)
    } else if constexpr (IP == 3) {
        constexpr float taur = -0.5f, taui = 0.8660254037844386467637231707529362f;
        if (item == 0) {
            const float tr2 = 2 * CC(ido - 1, 1, k);
            const float cr2 = CC(0, 0, k) + taur * tr2;
            CH(0, k, 0) = CC(0, 0, k) + tr2;
            const float ci3 = 2 * taui * CC(0, 2, k);
            CH(0, k, 2) = cr2 + ci3;
            CH(0, k, 1) = cr2 - ci3;
            return;
        }
        const float tr2 = CC(i - 1, 2, k) + CC(ic - 1, 1, k), ti2 = CC(i, 2, k) - CC(ic, 1, k);
        const float cr2 = CC(i - 1, 0, k) + taur * tr2, ci2 = CC(i, 0, k) + taur * ti2;
        CH(i - 1, k, 0) = CC(i - 1, 0, k) + tr2;
        CH(i, k, 0) = CC(i, 0, k) + ti2;
        const float cr3 = taui * (CC(i - 1, 2, k) - CC(ic - 1, 1, k)), ci3 = taui * (CC(i, 2, k) + CC(ic, 1, k));
        const float dr3 = cr2 + ci3, dr2 = cr2 - ci3, di2 = ci2 + cr3, di3 = ci2 - cr3;
        MULPM(CH(i, k, 1), CH(i - 1, k, 1), WA(0, i - 2), WA(0, i - 1), di2, dr2)
        MULPM(CH(i, k, 2), CH(i - 1, k, 2), WA(1, i - 2), WA(1, i - 1), di3, dr3)
    } else if constexpr (IP == 4) {
        constexpr float sqrt2 = 1.414213562373095048801688724209698f;
        if (item == 0) {
            const float tr2 = CC(0, 0, k) + CC(ido - 1, 3, k), tr1 = CC(0, 0, k) - CC(ido - 1, 3, k);
            const float tr3 = 2 * CC(ido - 1, 1, k), tr4 = 2 * CC(0, 2, k);
            CH(0, k, 0) = tr2 + tr3;
            CH(0, k, 2) = tr2 - tr3;
            CH(0, k, 3) = tr1 + tr4;
            CH(0, k, 1) = tr1 - tr4;
            if ((ido & 1) == 0) {
                const float ti1 = CC(0, 3, k) + CC(0, 1, k), ti2 = CC(0, 3, k) - CC(0, 1, k);
                const float ur2 = CC(ido - 1, 0, k) + CC(ido - 1, 2, k), ur1 = CC(ido - 1, 0, k) - CC(ido - 1, 2, k);
                CH(ido - 1, k, 0) = ur2 + ur2;
                CH(ido - 1, k, 1) = sqrt2 * (ur1 - ti1);
                CH(ido - 1, k, 2) = ti2 + ti2;
                CH(ido - 1, k, 3) = -sqrt2 * (ur1 + ti1);
            }
            return;
        }
        const float tr2 = CC(i - 1, 0, k) + CC(ic - 1, 3, k), tr1 = CC(i - 1, 0, k) - CC(ic - 1, 3, k);
        const float ti1 = CC(i, 0, k) + CC(ic, 3, k), ti2 = CC(i, 0, k) - CC(ic, 3, k);
        const float tr4 = CC(i, 2, k) + CC(ic, 1, k), ti3 = CC(i, 2, k) - CC(ic, 1, k);
        const float tr3 = CC(i - 1, 2, k) + CC(ic - 1, 1, k), ti4 = CC(i - 1, 2, k) - CC(ic - 1, 1, k);
        CH(i - 1, k, 0) = tr2 + tr3;
        const float cr3 = tr2 - tr3;
        CH(i, k, 0) = ti2 + ti3;
        const float ci3 = ti2 - ti3;
        const float cr4 = tr1 + tr4, cr2 = tr1 - tr4, ci2 = ti1 + ti4, ci4 = ti1 - ti4;
        MULPM(CH(i, k, 1), CH(i - 1, k, 1), WA(0, i - 2), WA(0, i - 1), ci2, cr2)
        MULPM(CH(i, k, 2), CH(i - 1, k, 2), WA(1, i - 2), WA(1, i - 1), ci3, cr3)
        MULPM(CH(i, k, 3), CH(i - 1, k, 3), WA(2, i - 2), WA(2, i - 1), ci4, cr4)
    } else {
        constexpr float tr11 = 0.3090169943749474241022934171828191f, ti11 = 0.9510565162951535721164393333793821f,
                        tr12 = -0.8090169943749474241022934171828191f, ti12 = 0.5877852522924731291687059546390728f;
        if (item == 0) {
            const float ti5 = CC(0, 2, k) + CC(0, 2, k), ti4 = CC(0, 4, k) + CC(0, 4, k);
            const float tr2 = CC(ido - 1, 1, k) + CC(ido - 1, 1, k), tr3 = CC(ido - 1, 3, k) + CC(ido - 1, 3, k);
            CH(0, k, 0) = CC(0, 0, k) + tr2 + tr3;
            const float cr2 = CC(0, 0, k) + tr11 * tr2 + tr12 * tr3, cr3 = CC(0, 0, k) + tr12 * tr2 + tr11 * tr3;
            float ci4, ci5;
            MULPM(ci5, ci4, ti5, ti4, ti11, ti12)
            CH(0, k, 4) = cr2 + ci5;
            CH(0, k, 1) = cr2 - ci5;
            CH(0, k, 3) = cr3 + ci4;
            CH(0, k, 2) = cr3 - ci4;
            return;
        }
        const float tr2 = CC(i - 1, 2, k) + CC(ic - 1, 1, k), tr5 = CC(i - 1, 2, k) - CC(ic - 1, 1, k);
        const float ti5 = CC(i, 2, k) + CC(ic, 1, k), ti2 = CC(i, 2, k) - CC(ic, 1, k);
        const float tr3 = CC(i - 1, 4, k) + CC(ic - 1, 3, k), tr4 = CC(i - 1, 4, k) - CC(ic - 1, 3, k);
        const float ti4 = CC(i, 4, k) + CC(ic, 3, k), ti3 = CC(i, 4, k) - CC(ic, 3, k);
        CH(i - 1, k, 0) = CC(i - 1, 0, k) + tr2 + tr3;
        CH(i, k, 0) = CC(i, 0, k) + ti2 + ti3;
        const float cr2 = CC(i - 1, 0, k) + tr11 * tr2 + tr12 * tr3, ci2 = CC(i, 0, k) + tr11 * ti2 + tr12 * ti3;
        const float cr3 = CC(i - 1, 0, k) + tr12 * tr2 + tr11 * tr3, ci3 = CC(i, 0, k) + tr12 * ti2 + tr11 * ti3;
        float ci4, ci5, cr5, cr4;
        MULPM(cr5, cr4, tr5, tr4, ti11, ti12)
        MULPM(ci5, ci4, ti5, ti4, ti11, ti12)
        const float dr4 = cr3 + ci4, dr3 = cr3 - ci4, di3 = ci3 + cr4, di4 = ci3 - cr4;
        const float dr5 = cr2 + ci5, dr2 = cr2 - ci5, di2 = ci2 + cr5, di5 = ci2 - cr5;
        MULPM(CH(i, k, 1), CH(i - 1, k, 1), WA(0, i - 2), WA(0, i - 1), di2, dr2)
        MULPM(CH(i, k, 2), CH(i - 1, k, 2), WA(1, i - 2), WA(1, i - 1), di3, dr3)
        MULPM(CH(i, k, 3), CH(i - 1, k, 3), WA(2, i - 2), WA(2, i - 1), di4, dr4)
        MULPM(CH(i, k, 4), CH(i - 1, k, 4), WA(3, i - 2), WA(3, i - 1), di5, dr5)
    }
#undef CC
#undef CH
}
#undef MULPM
#undef WA

template <int IP, bool R2HC>
__global__ __launch_bounds__(kBlock) void rfft_pass_kernel(const float* __restrict__ src,
                                                           float* __restrict__ dst,
                                                           const float* __restrict__ wa,
                                                           uint64_t transforms, uint32_t n,
                                                           uint32_t ido, uint32_t l1) {
    const uint32_t items = 1 + (ido - 1) / 2;
    const uint64_t per_t = (uint64_t)l1 * items, total = transforms * per_t;
    for (uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x; g < total;
         g += (uint64_t)gridDim.x * kBlock) {
        const uint64_t t = g / per_t;
        const uint32_t w = (uint32_t)(g % per_t), item = w % items, k = w / items;
        if constexpr (R2HC) radf_item<IP>(ido, l1, src + t * n, dst + t * n, wa, k, item);
        else radb_item<IP>(ido, l1, src + t * n, dst + t * n, wa, k, item);
    }
}

// ---- generic radix (radfg / radbg, pocketfft.hh:1753-1893, 2076-2208) ---------------------------
// The reference walks whole arrays in phases and uses BOTH buffers (cc is overwritten); each phase
// is independent per butterfly (j pair, k, i) or per column ik = i + ido*k, so every phase becomes
// one launch.  C1/CH(a,b,c) = x[a + ido*(b + l1*c)], CC(a,b,c) = x[a + ido*(b + ip*c)],
// C2/CH2(a,b) = x[a + idl1*b]; csarr = the ip-th roots of unity (comp_twiddle's tws).
struct GenDims {
    uint32_t n, ido, ip, l1, ipph, idl1, items;  // items = 1 + (ido-1)/2 per (j pair, k)
};
#define C1(x, a, b, c) x[(a) + d.ido * ((b) + d.l1 * (c))]
#define CCX(x, a, b, c) x[(a) + d.ido * ((b) + d.ip * (c))]
#define C2(x, a, b) x[(a) + d.idl1 * (b)]

// radfg phase 1+2 (in place on cc): twiddle the (j, jc) pairs, MPINPLACE on i = 0
__global__ __launch_bounds__(kBlock) void radfg_pre_kernel(float* __restrict__ ccbuf,
                                                           const float* __restrict__ wa, GenDims d,
                                                           uint64_t transforms) {
    const uint64_t per_t = (uint64_t)(d.ipph - 1) * d.l1 * d.items, total = transforms * per_t;
    for (uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x; g < total;
         g += (uint64_t)gridDim.x * kBlock) {
        float* cc = ccbuf + (g / per_t) * d.n;
        uint32_t w = (uint32_t)(g % per_t);
        const uint32_t item = w % d.items;
        w /= d.items;
        const uint32_t k = w % d.l1, j = 1 + w / d.l1, jc = d.ip - j;
        if (item == 0) {  // MPINPLACE(C1(0,k,jc), C1(0,k,j))
            const float t = C1(cc, 0, k, jc);
            C1(cc, 0, k, jc) = C1(cc, 0, k, jc) - C1(cc, 0, k, j);
            C1(cc, 0, k, j) = t + C1(cc, 0, k, j);
            continue;
        }
        const uint32_t i = 2 * item - 1;
        const uint32_t idij = (j - 1) * (d.ido - 1) + i - 1, idij2 = (jc - 1) * (d.ido - 1) + i - 1;
        const float t1 = C1(cc, i, k, j), t2 = C1(cc, i + 1, k, j), t3 = C1(cc, i, k, jc), t4 = C1(cc, i + 1, k, jc);
        const float x1 = wa[idij] * t1 + wa[idij + 1] * t2, x2 = wa[idij] * t2 - wa[idij + 1] * t1,
                    x3 = wa[idij2] * t3 + wa[idij2 + 1] * t4, x4 = wa[idij2] * t4 - wa[idij2 + 1] * t3;
        C1(cc, i, k, j) = x3 + x1;
        C1(cc, i + 1, k, jc) = x3 - x1;
        C1(cc, i + 1, k, j) = x2 + x4;
        C1(cc, i, k, jc) = x2 - x4;
    }
}
// The l loop shared by radfg (FORWARD: cc -> ch, wrap test iang >= ip) and radbg (ch -> cc, wrap test
// iang > ip), then the column sum into column 0 of `sum_dst` (radfg: ch, from cc; radbg: ch, in place).
template <bool FORWARD>
__global__ __launch_bounds__(kBlock) void radg_mix_kernel(const float* __restrict__ srcbuf,
                                                          float* __restrict__ dstbuf,
                                                          const float* __restrict__ csarr, GenDims d,
                                                          uint64_t transforms) {
    const uint64_t total = transforms * d.idl1;
    for (uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x; g < total;
         g += (uint64_t)gridDim.x * kBlock) {
        const uint64_t t = g / d.idl1;
        const uint32_t ik = (uint32_t)(g % d.idl1);
        const float* src = srcbuf + t * d.n;
        float* dst = dstbuf + t * d.n;
        for (uint32_t l = 1, lc = d.ip - 1; l < d.ipph; ++l, --lc) {
            float a = C2(src, ik, 0) + csarr[2 * l] * C2(src, ik, 1) + csarr[4 * l] * C2(src, ik, 2);
            float b = csarr[2 * l + 1] * C2(src, ik, d.ip - 1) + csarr[4 * l + 1] * C2(src, ik, d.ip - 2);
            uint32_t iang = 2 * l, j = 3, jc = d.ip - 3;
            auto step = [&]() {
                iang += l;
                if (FORWARD ? (iang >= d.ip) : (iang > d.ip)) iang -= d.ip;
            };
            for (; j + 3 < d.ipph; j += 4, jc -= 4) {
                step();
                const float ar1 = csarr[2 * iang], ai1 = csarr[2 * iang + 1];
                step();
                const float ar2 = csarr[2 * iang], ai2 = csarr[2 * iang + 1];
                step();
                const float ar3 = csarr[2 * iang], ai3 = csarr[2 * iang + 1];
                step();
                const float ar4 = csarr[2 * iang], ai4 = csarr[2 * iang + 1];
                a += ar1 * C2(src, ik, j) + ar2 * C2(src, ik, j + 1) + ar3 * C2(src, ik, j + 2) + ar4 * C2(src, ik, j + 3);
                b += ai1 * C2(src, ik, jc) + ai2 * C2(src, ik, jc - 1) + ai3 * C2(src, ik, jc - 2) + ai4 * C2(src, ik, jc - 3);
            }
            for (; j + 1 < d.ipph; j += 2, jc -= 2) {
                step();
                const float ar1 = csarr[2 * iang], ai1 = csarr[2 * iang + 1];
                step();
                const float ar2 = csarr[2 * iang], ai2 = csarr[2 * iang + 1];
                a += ar1 * C2(src, ik, j) + ar2 * C2(src, ik, j + 1);
                b += ai1 * C2(src, ik, jc) + ai2 * C2(src, ik, jc - 1);
            }
            for (; j < d.ipph; ++j, --jc) {
                step();
                const float ar = csarr[2 * iang], ai = csarr[2 * iang + 1];
                a += ar * C2(src, ik, j);
                b += ai * C2(src, ik, jc);
            }
            C2(dst, ik, l) = a;
            C2(dst, ik, lc) = b;
        }
        if (FORWARD) {  // CH2(ik,0) = C2(ik,0) + C2(ik,1) + ... (:1869-1872)
            float s0 = C2(src, ik, 0);
            for (uint32_t j = 1; j < d.ipph; ++j) s0 += C2(src, ik, j);
            C2(dst, ik, 0) = s0;
        }
    }
}
// radbg: CH2(ik,0) += CH2(ik,1) + ... in place (:2176-2178); separate launch because radg_mix
// reads column 0 of ch for every l
__global__ __launch_bounds__(kBlock) void radbg_sum_kernel(float* __restrict__ chbuf, GenDims d,
                                                           uint64_t transforms) {
    const uint64_t total = transforms * d.idl1;
    for (uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x; g < total;
         g += (uint64_t)gridDim.x * kBlock) {
        float* ch = chbuf + (g / d.idl1) * d.n;
        const uint32_t ik = (uint32_t)(g % d.idl1);
        float s0 = C2(ch, ik, 0);
        for (uint32_t j = 1; j < d.ipph; ++j) s0 += C2(ch, ik, j);
        C2(ch, ik, 0) = s0;
    }
}
// radfg phase 5: CC <- CH (halfcomplex packing, :1877-1893); also the j = 0 plane
__global__ __launch_bounds__(kBlock) void radfg_post_kernel(float* __restrict__ ccbuf,
                                                            const float* __restrict__ chbuf, GenDims d,
                                                            uint64_t transforms) {
    const uint64_t pairs = (uint64_t)(d.ipph - 1) * d.l1 * d.items, plane = d.idl1;
    const uint64_t per_t = pairs + plane, total = transforms * per_t;
    for (uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x; g < total;
         g += (uint64_t)gridDim.x * kBlock) {
        const uint64_t t = g / per_t;
        float* cc = ccbuf + t * d.n;
        const float* ch = chbuf + t * d.n;
        uint64_t w = g % per_t;
        if (w >= pairs) {  // CC(i,0,k) = CH(i,k,0)
            const uint32_t ik = (uint32_t)(w - pairs), i = ik % d.ido, k = ik / d.ido;
            CCX(cc, i, 0, k) = C1(ch, i, k, 0);
            continue;
        }
        const uint32_t item = (uint32_t)(w % d.items);
        w /= d.items;
        const uint32_t k = (uint32_t)(w % d.l1), j = 1 + (uint32_t)(w / d.l1), jc = d.ip - j, j2 = 2 * j - 1;
        if (item == 0) {
            CCX(cc, d.ido - 1, j2, k) = C1(ch, 0, k, j);
            CCX(cc, 0, j2 + 1, k) = C1(ch, 0, k, jc);
            continue;
        }
        const uint32_t i = 2 * item - 1, ic = d.ido - i - 2;
        CCX(cc, i, j2 + 1, k) = C1(ch, i, k, j) + C1(ch, i, k, jc);
        CCX(cc, ic, j2, k) = C1(ch, i, k, j) - C1(ch, i, k, jc);
        CCX(cc, i + 1, j2 + 1, k) = C1(ch, i + 1, k, j) + C1(ch, i + 1, k, jc);
        CCX(cc, ic + 1, j2, k) = C1(ch, i + 1, k, jc) - C1(ch, i + 1, k, j);
    }
}
// radbg phase A: CH <- CC (unpacking, :2095-2122)
__global__ __launch_bounds__(kBlock) void radbg_pre_kernel(const float* __restrict__ ccbuf,
                                                           float* __restrict__ chbuf, GenDims d,
                                                           uint64_t transforms) {
    const uint64_t pairs = (uint64_t)(d.ipph - 1) * d.l1 * d.items, plane = d.idl1;
    const uint64_t per_t = pairs + plane, total = transforms * per_t;
    for (uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x; g < total;
         g += (uint64_t)gridDim.x * kBlock) {
        const uint64_t t = g / per_t;
        const float* cc = ccbuf + t * d.n;
        float* ch = chbuf + t * d.n;
        uint64_t w = g % per_t;
        if (w >= pairs) {
            const uint32_t ik = (uint32_t)(w - pairs), i = ik % d.ido, k = ik / d.ido;
            C1(ch, i, k, 0) = CCX(cc, i, 0, k);
            continue;
        }
        const uint32_t item = (uint32_t)(w % d.items);
        w /= d.items;
        const uint32_t k = (uint32_t)(w % d.l1), j = 1 + (uint32_t)(w / d.l1), jc = d.ip - j, j2 = 2 * j - 1;
        if (item == 0) {
            C1(ch, 0, k, j) = 2 * CCX(cc, d.ido - 1, j2, k);
            C1(ch, 0, k, jc) = 2 * CCX(cc, 0, j2 + 1, k);
            continue;
        }
        const uint32_t i = 2 * item - 1, ic = d.ido - i - 2;
        C1(ch, i, k, j) = CCX(cc, i, j2 + 1, k) + CCX(cc, ic, j2, k);
        C1(ch, i, k, jc) = CCX(cc, i, j2 + 1, k) - CCX(cc, ic, j2, k);
        C1(ch, i + 1, k, j) = CCX(cc, i + 1, j2 + 1, k) - CCX(cc, ic + 1, j2, k);
        C1(ch, i + 1, k, jc) = CCX(cc, i + 1, j2 + 1, k) + CCX(cc, ic + 1, j2, k);
    }
}
// radbg phases D+E: CH <- PM of C1, then the output twiddles (:2179-2207)
__global__ __launch_bounds__(kBlock) void radbg_post_kernel(const float* __restrict__ ccbuf,
                                                            float* __restrict__ chbuf,
                                                            const float* __restrict__ wa, GenDims d,
                                                            uint64_t transforms) {
    const uint64_t per_t = (uint64_t)(d.ipph - 1) * d.l1 * d.items, total = transforms * per_t;
    for (uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x; g < total;
         g += (uint64_t)gridDim.x * kBlock) {
        const uint64_t t = g / per_t;
        const float* cc = ccbuf + t * d.n;
        float* ch = chbuf + t * d.n;
        uint32_t w = (uint32_t)(g % per_t);
        const uint32_t item = w % d.items;
        w /= d.items;
        const uint32_t k = w % d.l1, j = 1 + w / d.l1, jc = d.ip - j;
        if (item == 0) {
            C1(ch, 0, k, jc) = C1(cc, 0, k, j) + C1(cc, 0, k, jc);
            C1(ch, 0, k, j) = C1(cc, 0, k, j) - C1(cc, 0, k, jc);
            continue;
        }
        const uint32_t i = 2 * item - 1;
        const float a0 = C1(cc, i, k, j) - C1(cc, i + 1, k, jc), b0 = C1(cc, i, k, j) + C1(cc, i + 1, k, jc);
        const float a1 = C1(cc, i + 1, k, j) + C1(cc, i, k, jc), b1 = C1(cc, i + 1, k, j) - C1(cc, i, k, jc);
        const uint32_t ij = (j - 1) * (d.ido - 1) + i - 1, ijc = (jc - 1) * (d.ido - 1) + i - 1;
        C1(ch, i, k, j) = wa[ij] * a0 - wa[ij + 1] * a1;
        C1(ch, i + 1, k, j) = wa[ij] * a1 + wa[ij + 1] * a0;
        C1(ch, i, k, jc) = wa[ijc] * b0 - wa[ijc + 1] * b1;
        C1(ch, i + 1, k, jc) = wa[ijc] * b1 + wa[ijc + 1] * b0;
    }
}
#undef C1
#undef CCX
#undef C2

// strided tensor row <-> dense row, and the r2c re-packing (general_r2c, pocketfft.hh:3102-3150)
__global__ __launch_bounds__(kBlock) void rfft_gather_kernel(const FftLayout L, float* __restrict__ dense,
                                                             const float* __restrict__ in, uint32_t n) {
    const uint64_t total = L.transforms * n;
    for (uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x; g < total;
         g += (uint64_t)gridDim.x * kBlock) {
        uint64_t t = g / n;
        const uint32_t m = (uint32_t)(g % n);
        int64_t base = (int64_t)L.in_offset;
        for (int a = L.outer_rank - 1; a >= 0; --a) {
            base += (int64_t)(t % L.outer_shape[a]) * L.in_outer_stride[a];
            t /= L.outer_shape[a];
        }
        dense[g] = in[base + (int64_t)m * L.in_axis_stride];
    }
}
// mode 0: real row out (r2r); mode 1: complex n/2+1 bins out (r2c, forward)
__global__ __launch_bounds__(kBlock) void rfft_scatter_kernel(const FftLayout L, float* __restrict__ out,
                                                              const float* __restrict__ dense,
                                                              uint32_t n, int complex_out) {
    const uint32_t no = complex_out ? n / 2 + 1 : n;
    const uint64_t total = L.transforms * no;
    for (uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x; g < total;
         g += (uint64_t)gridDim.x * kBlock) {
        uint64_t t = g / no;
        const uint32_t m = (uint32_t)(g % no);
        const float* row = dense + t * n;
        int64_t base = (int64_t)L.out_offset;
        for (int a = L.outer_rank - 1; a >= 0; --a) {
            base += (int64_t)(t % L.outer_shape[a]) * L.out_outer_stride[a];
            t /= L.outer_shape[a];
        }
        const int64_t o = base + (int64_t)m * L.out_axis_stride;
        if (!complex_out) {
            out[o] = row[m];
        } else {  // bin 0 = (r0, 0); bin m = (r[2m-1], r[2m]); last bin of an even n = (r[n-1], 0)
            float re, im;
            if (m == 0) { re = row[0]; im = 0.0f; }
            else if (2 * m < n) { re = row[2 * m - 1]; im = row[2 * m]; }
            else { re = row[n - 1]; im = 0.0f; }
            out[2 * o] = re;
            out[2 * o + 1] = im;
        }
    }
}
// fftblue::exec_r (pocketfft.hh:2434-2457): real row <-> the complex line Bluestein transforms
__global__ __launch_bounds__(kBlock) void rfft_blue_in_kernel(float2* __restrict__ c,
                                                              const float* __restrict__ dense,
                                                              uint64_t transforms, uint32_t n, int r2hc) {
    const uint64_t total = transforms * n;
    for (uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x; g < total;
         g += (uint64_t)gridDim.x * kBlock) {
        const uint64_t t = g / n;
        const uint32_t m = (uint32_t)(g % n);
        const float* row = dense + t * n;
        if (r2hc) {
            c[g] = mk(row[m], 0.0f * row[0]);
        } else {  // halfcomplex -> Hermitian-symmetric complex line
            float re, im;
            if (m == 0) { re = row[0]; im = row[0] * 0.0f; }
            else if (2 * m < n) { re = row[2 * m - 1]; im = row[2 * m]; }
            else if (2 * m == n) { re = row[n - 1]; im = 0.0f * row[0]; }
            else { const uint32_t q = n - m; re = row[2 * q - 1]; im = -row[2 * q]; }
            c[g] = mk(re, im);
        }
    }
}
__global__ __launch_bounds__(kBlock) void rfft_blue_out_kernel(float* __restrict__ dense,
                                                               const float2* __restrict__ c,
                                                               uint64_t transforms, uint32_t n, int r2hc) {
    const uint64_t total = transforms * n;
    for (uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x; g < total;
         g += (uint64_t)gridDim.x * kBlock) {
        const uint64_t t = g / n;
        const uint32_t m = (uint32_t)(g % n);
        const float2* row = c + t * n;
        if (!r2hc) dense[g] = row[m].x;
        else if (m == 0) dense[g] = row[0].x;
        else dense[g] = (m & 1) ? row[(m + 1) / 2].x : row[m / 2].y;  // r1 i1 r2 i2 ...
    }
}

inline unsigned blocks_for(uint64_t total) {
    uint64_t b = (total + kBlock - 1) / kBlock;
    if (b > 16384) b = 16384;
    return (unsigned)(b ? b : 1);
}

}  // namespace

int rfft_plan_factors(uint64_t n, uint32_t* fact) {  // rfftp::factorize, pocketfft.hh:2277-2297
    int nf = 0;
    uint64_t len = n;
    if (len <= 1) return 0;
    while ((len % 4) == 0) { fact[nf++] = 4; len >>= 2; }
    if ((len % 2) == 0) {
        len >>= 1;
        fact[nf++] = 2;
        const uint32_t t = fact[0];
        fact[0] = fact[nf - 1];
        fact[nf - 1] = t;
    }
    for (uint64_t d = 3; d * d <= len; d += 2)
        while ((len % d) == 0) {
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
bool rfft_supported(uint64_t n) {
    if (n < 1 || n > (1ull << 30)) return false;
    uint32_t fact[64];
    const int nf = rfft_plan_factors(n, fact);
    if (nf < 0) return false;
    for (int i = 0; i < nf; ++i)
        if (fact[i] > 5 && fact[i] > 4096) return false;  // generic radix: one thread per column
    return true;
}
uint64_t rfft_twiddle_count(uint64_t n) {
    uint32_t fact[64];
    const int nf = rfft_plan_factors(n, fact);
    uint64_t total = 0, l1 = 1;
    for (int k = 0; k < nf; ++k) {
        total += (uint64_t)(fact[k] - 1) * (n / (l1 * fact[k]) - 1);
        if (fact[k] > 5) total += 2ull * fact[k];
        l1 *= fact[k];
    }
    return total;
}
void rfft_twiddle_fill(uint64_t n, const float* w, float* out) {  // comp_twiddle, :2311-2345
    uint32_t fact[64];
    const int nf = rfft_plan_factors(n, fact);
    uint64_t off = 0, l1 = 1;
    for (int k = 0; k < nf; ++k) {
        const uint64_t ip = fact[k], ido = n / (l1 * ip);
        for (uint64_t j = 1; j < ip; ++j)
            for (uint64_t i = 1; i <= (ido - 1) / 2; ++i) {
                out[off + (j - 1) * (ido - 1) + 2 * i - 2] = w[2 * (j * l1 * i)];
                out[off + (j - 1) * (ido - 1) + 2 * i - 1] = w[2 * (j * l1 * i) + 1];
            }
        off += (ip - 1) * (ido - 1);
        if (ip > 5) {  // "special factors required by *g functions": the ip-th roots of unity
            float* t = out + off;
            t[0] = 1.0f;
            t[1] = 0.0f;
            for (uint64_t i = 2, ic = 2 * ip - 2; i <= ic; i += 2, ic -= 2) {
                const uint64_t src = i / 2 * (n / ip);
                t[i] = w[2 * src];
                t[i + 1] = w[2 * src + 1];
                t[ic] = w[2 * src];
                t[ic + 1] = -w[2 * src + 1];
            }
            off += 2 * ip;
        }
        l1 *= ip;
    }
}

hipError_t launch_rfft_gather(const FftLayout& L, float* dense, const float* in, uint64_t n, hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(rfft_gather_kernel, dim3(blocks_for(L.transforms * n)), dim3(kBlock), 0, s, L, dense,
                       in, (uint32_t)n);
    return hipGetLastError();
}
hipError_t launch_rfft_scatter(const FftLayout& L, float* out, const float* dense, uint64_t n,
                               bool complex_out, hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(rfft_scatter_kernel, dim3(blocks_for(L.transforms * n)), dim3(kBlock), 0, s, L, out,
                       dense, (uint32_t)n, complex_out ? 1 : 0);
    return hipGetLastError();
}
hipError_t launch_rfft_blue_in(float2* c, const float* dense, uint64_t transforms, uint64_t n, bool r2hc,
                               hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(rfft_blue_in_kernel, dim3(blocks_for(transforms * n)), dim3(kBlock), 0, s, c, dense,
                       transforms, (uint32_t)n, r2hc ? 1 : 0);
    return hipGetLastError();
}
hipError_t launch_rfft_blue_out(float* dense, const float2* c, uint64_t transforms, uint64_t n, bool r2hc,
                                hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(rfft_blue_out_kernel, dim3(blocks_for(transforms * n)), dim3(kBlock), 0, s, dense, c,
                       transforms, (uint32_t)n, r2hc ? 1 : 0);
    return hipGetLastError();
}

// rfftp::exec with fct == 1 (pocketfft.hh:2228-2272) on dense rows; the result ends in *result
// (one of a, b).  tw: rfft_twiddle_fill layout.
hipError_t launch_rfft_passes(uint64_t n, bool r2hc, uint64_t transforms, float* a, float* b,
                              const float* tw, float** result, hipStream_t s) {
    uint32_t fact[64];
    const int nf = rfft_plan_factors(n, fact);
    *result = a;
    if (nf <= 0) return nf == 0 ? hipSuccess : hipErrorInvalidValue;
    uint64_t off[64], o = 0, l1 = 1;
    for (int k = 0; k < nf; ++k) {
        off[k] = o;
        o += (uint64_t)(fact[k] - 1) * (n / (l1 * fact[k]) - 1);
        if (fact[k] > 5) o += 2ull * fact[k];
        l1 *= fact[k];
    }
    auto generic = [&](int k, uint64_t ip, uint64_t ido, uint64_t pl1, float* cc, float* ch, bool fwd) {
        GenDims d;
        d.n = (uint32_t)n;
        d.ido = (uint32_t)ido;
        d.ip = (uint32_t)ip;
        d.l1 = (uint32_t)pl1;
        d.ipph = (uint32_t)((ip + 1) / 2);
        d.idl1 = (uint32_t)(ido * pl1);
        d.items = (uint32_t)(1 + (ido - 1) / 2);
        const float* wa = tw + off[k];
        const float* cs = wa + (ip - 1) * (ido - 1);
        const uint64_t pairs = transforms * (d.ipph - 1) * pl1 * d.items, cols = transforms * d.idl1;
        if (fwd) {  // result ends in cc
            hipLaunchKernelGGL(radfg_pre_kernel, dim3(blocks_for(pairs)), dim3(kBlock), 0, s, cc, wa, d, transforms);
            hipLaunchKernelGGL((radg_mix_kernel<true>), dim3(blocks_for(cols)), dim3(kBlock), 0, s,
                               (const float*)cc, ch, cs, d, transforms);
            hipLaunchKernelGGL(radfg_post_kernel, dim3(blocks_for(pairs + cols)), dim3(kBlock), 0, s, cc,
                               (const float*)ch, d, transforms);
        } else {    // result ends in ch
            hipLaunchKernelGGL(radbg_pre_kernel, dim3(blocks_for(pairs + cols)), dim3(kBlock), 0, s,
                               (const float*)cc, ch, d, transforms);
            hipLaunchKernelGGL((radg_mix_kernel<false>), dim3(blocks_for(cols)), dim3(kBlock), 0, s,
                               (const float*)ch, cc, cs, d, transforms);
            hipLaunchKernelGGL(radbg_sum_kernel, dim3(blocks_for(cols)), dim3(kBlock), 0, s, ch, d, transforms);
            hipLaunchKernelGGL(radbg_post_kernel, dim3(blocks_for(pairs)), dim3(kBlock), 0, s,
                               (const float*)cc, ch, wa, d, transforms);
        }
    };
    float *p1 = a, *p2 = b;
    (void)hipGetLastError();
#define JST_RPASS(IP, DIR)                                                                            \
    hipLaunchKernelGGL((rfft_pass_kernel<IP, DIR>), dim3(blocks_for(transforms * pl1 * (1 + (ido - 1) / 2))), \
                       dim3(kBlock), 0, s, (const float*)p1, p2, tw + off[k], transforms, (uint32_t)n,  \
                       (uint32_t)ido, (uint32_t)pl1)
    if (r2hc) {
        uint64_t cur = n;
        for (int k1 = 0; k1 < nf; ++k1) {
            const int k = nf - k1 - 1;
            const uint64_t ip = fact[k], ido = n / cur;
            cur /= ip;
            const uint64_t pl1 = cur;
            switch (ip) {
                case 2: JST_RPASS(2, true); break;
                case 3: JST_RPASS(3, true); break;
                case 4: JST_RPASS(4, true); break;
                case 5: JST_RPASS(5, true); break;
                default: {  // radfg leaves its result in the source buffer (the reference swaps twice)
                    generic(k, ip, ido, pl1, p1, p2, true);
                    float* u = p1; p1 = p2; p2 = u;
                } break;
            }
            float* t = p1; p1 = p2; p2 = t;
        }
    } else {
        uint64_t pl1 = 1;
        for (int k = 0; k < nf; ++k) {
            const uint64_t ip = fact[k], ido = n / (ip * pl1);
            switch (ip) {
                case 2: JST_RPASS(2, false); break;
                case 3: JST_RPASS(3, false); break;
                case 4: JST_RPASS(4, false); break;
                case 5: JST_RPASS(5, false); break;
                default: generic(k, ip, ido, pl1, p1, p2, false); break;
            }
            float* t = p1; p1 = p2; p2 = t;
            pl1 *= ip;
        }
    }
#undef JST_RPASS
    *result = p1;
    return hipGetLastError();
}

// pocketfft_r's plan choice (pocketfft.hh:2508-2527): 0 = rfftp, else the Bluestein length.
uint64_t rfft_bluestein_size(uint64_t n) { return fft_bluestein_size_scaled(n, 0.5); }

}  // namespace jst::kernels
