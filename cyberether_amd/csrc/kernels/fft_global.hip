// fft_global.hip -- general-length FFT: one Stockham pass per launch, through HBM.
//
// Covers what the LDS-resident kernels (fft_lds.hh) do not: lengths above 16384 points and
// lengths with factors 3 and 5 (the Filter block's convolution size S + taps - 1 is rarely a power
// of two: 160000 = 8*8*4*5*5*5*5 for BASELINE config 3).  Same plan, butterflies and twiddle table
// as pocketfft's cfftp (factor order pocketfft.hh:1476-1497; pass2/3/4/5/8 :843-1223), so results
// are bit-identical to the reference CPU path; the passes ping-pong between two dense scratch
// tensors, the first reading the (strided) input and the last writing the (strided) output.
// One thread per butterfly: writes are fully coalesced (u + c*N/ip), reads are contiguous runs of
// ido elements.  Traffic is nf x 16 B per sample -- correctness-first; a fused LDS multi-pass
// version for 65536 points is future work.
#include "device_math.hh"
#include "kernels.hh"

namespace jst::kernels {

using namespace jst::dev;

namespace {

constexpr int kBlock = 256;

// pocketfft pass3 (pocketfft.hh:873-923) without the output twiddles
template <bool FWD>
__device__ __forceinline__ void butterfly3(float2 (&x)[3]) {
    constexpr float tw1r = -0.5f;
    constexpr float tw1i = (FWD ? -1 : 1) * 0.8660254037844386467637231707529362f;
    const float2 t0 = x[0], t1 = cadd(x[1], x[2]), t2 = csub(x[1], x[2]);
    x[0] = cadd(t0, t1);
    const float2 ca = mk(t0.x + t1.x * tw1r, t0.y + t1.y * tw1r);
    const float2 cb = mk(-t2.y * tw1i, t2.x * tw1i);
    x[1] = cadd(ca, cb);
    x[2] = csub(ca, cb);
}
// pocketfft pass5 (pocketfft.hh:976-1050) without the output twiddles
template <bool FWD>
__device__ __forceinline__ void butterfly5(float2 (&x)[5]) {
    constexpr float tw1r = 0.3090169943749474241022934171828191f;
    constexpr float tw1i = (FWD ? -1 : 1) * 0.9510565162951535721164393333793821f;
    constexpr float tw2r = -0.8090169943749474241022934171828191f;
    constexpr float tw2i = (FWD ? -1 : 1) * 0.5877852522924731291687059546390728f;
    const float2 t0 = x[0];
    const float2 t1 = cadd(x[1], x[4]), t4 = csub(x[1], x[4]);
    const float2 t2 = cadd(x[2], x[3]), t3 = csub(x[2], x[3]);
    x[0] = mk(t0.x + t1.x + t2.x, t0.y + t1.y + t2.y);
    const float2 ca = mk(t0.x + tw1r * t1.x + tw2r * t2.x, t0.y + tw1r * t1.y + tw2r * t2.y);
    float2 cb;
    cb.y = tw1i * t4.x + tw2i * t3.x;
    cb.x = -(tw1i * t4.y + tw2i * t3.y);
    const float2 da = mk(t0.x + tw2r * t1.x + tw1r * t2.x, t0.y + tw2r * t1.y + tw1r * t2.y);
    float2 db;
    db.y = tw2i * t4.x - tw1i * t3.x;
    db.x = -(tw2i * t4.y - tw1i * t3.y);
    x[1] = cadd(ca, cb);
    x[4] = csub(ca, cb);
    x[2] = cadd(da, db);
    x[3] = csub(da, db);
}
template <int IP, bool FWD>
__device__ __forceinline__ void butterfly_any(float2 (&x)[IP]) {
    if constexpr (IP == 3) butterfly3<FWD>(x);
    else if constexpr (IP == 5) butterfly5<FWD>(x);
    else butterfly<IP, FWD>(x);
}

struct PassIo {
    // source: element (t, p) at src[src_base(t) + p * src_stride]; dense scratch: base = t*N
    const float2* src;
    float2* dst;
    int64_t src_stride, dst_stride;
    int32_t src_strided, dst_strided;  // 1: use the FftLayout outer decode (tensor), 0: dense
};

template <int IP, bool FWD>
__global__ __launch_bounds__(kBlock) void fft_gpass_kernel(const FftLayout L, const PassIo io,
                                                           const float2* __restrict__ W,
                                                           uint64_t n, uint64_t l1, uint64_t ido) {
    const uint64_t but = n / IP, total = L.transforms * but;
    for (uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x; g < total;
         g += (uint64_t)gridDim.x * kBlock) {
        const uint64_t t = g / but, u = g % but;
        const uint64_t i = u % ido, k = u / ido;
        int64_t sbase = (int64_t)(t * n), dbase = (int64_t)(t * n);
        if (io.src_strided | io.dst_strided) {
            int64_t in_base = (int64_t)L.in_offset, out_base = (int64_t)L.out_offset;
            uint64_t rem = t;
            for (int a = L.outer_rank - 1; a >= 0; --a) {
                const uint64_t c = rem % L.outer_shape[a];
                rem /= L.outer_shape[a];
                in_base += (int64_t)c * L.in_outer_stride[a];
                out_base += (int64_t)c * L.out_outer_stride[a];
            }
            if (io.src_strided) sbase = in_base;
            if (io.dst_strided) dbase = out_base;
        }
        float2 x[IP];
#pragma unroll
        for (int b = 0; b < IP; ++b)
            x[b] = io.src[sbase + (int64_t)(i + ido * ((uint64_t)b + IP * k)) * io.src_stride];
        butterfly_any<IP, FWD>(x);
        if (i != 0) {
#pragma unroll
            for (int c = 1; c < IP; ++c) x[c] = special_mul<FWD>(x[c], W[(uint64_t)c * l1 * i]);
        }
#pragma unroll
        for (int c = 0; c < IP; ++c)
            io.dst[dbase + (int64_t)(u + (uint64_t)c * but) * io.dst_stride] = x[c];
    }
}

template <bool FWD>
hipError_t launch_pass(int ip, const FftLayout& L, const PassIo& io, const float2* W, uint64_t n,
                       uint64_t l1, uint64_t ido, hipStream_t s) {
    const uint64_t total = L.transforms * (n / (uint64_t)ip);
    uint64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 16384) blocks = 16384;
    if (blocks == 0) return hipSuccess;
    (void)hipGetLastError();
#define JST_GPASS(IP) \
    hipLaunchKernelGGL((fft_gpass_kernel<IP, FWD>), dim3((unsigned)blocks), dim3(kBlock), 0, s, L, io, W, n, l1, ido)
    switch (ip) {
        case 2: JST_GPASS(2); break;
        case 3: JST_GPASS(3); break;
        case 4: JST_GPASS(4); break;
        case 5: JST_GPASS(5); break;
        case 8: JST_GPASS(8); break;
        default: return hipErrorInvalidValue;
    }
#undef JST_GPASS
    return hipGetLastError();
}

}  // namespace

int fft_plan_factors(uint64_t n, uint32_t* fact) {
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
    if (len > 1) fact[nf++] = (uint32_t)len;
    for (int i = 0; i < nf; ++i)
        if (fact[i] != 2 && fact[i] != 3 && fact[i] != 4 && fact[i] != 5 && fact[i] != 8) return -1;
    return nf;
}

bool fft_global_supported(uint64_t n) {
    uint32_t fact[64];
    return n >= 1 && n <= (1ull << 31) && fft_plan_factors(n, fact) >= 0;
}

hipError_t launch_fft_c2c_global(uint64_t n, bool forward, const FftLayout& L, const float2* W,
                                 const float2* in, float2* out, float2* scratch_a,
                                 float2* scratch_b, hipStream_t s) {
    uint32_t fact[64];
    const int nf = fft_plan_factors(n, fact);
    if (nf < 0) return hipErrorInvalidValue;
    if (nf == 0) {  // n == 1: copy through a degenerate radix... handled by the LDS path
        return hipErrorInvalidValue;
    }
    uint64_t l1 = 1;
    const float2* src = in;
    for (int p = 0; p < nf; ++p) {
        const uint64_t ip = fact[p], ido = n / (l1 * ip);
        const bool first = p == 0, last = p == nf - 1;
        float2* dst = last ? out : ((p & 1) ? scratch_b : scratch_a);
        PassIo io;
        io.src = src;
        io.dst = dst;
        io.src_strided = first ? 1 : 0;
        io.dst_strided = last ? 1 : 0;
        io.src_stride = first ? L.in_axis_stride : 1;
        io.dst_stride = last ? L.out_axis_stride : 1;
        const hipError_t e = forward ? launch_pass<true>((int)ip, L, io, W, n, l1, ido, s)
                                     : launch_pass<false>((int)ip, L, io, W, n, l1, ido, s);
        if (e != hipSuccess) return e;
        src = dst;
        l1 *= ip;
    }
    return hipSuccess;
}

}  // namespace jst::kernels
