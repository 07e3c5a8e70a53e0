// fft_radix.hh -- the odd-radix butterflies of pocketfft's cfftp (pass3/5/7/11, pocketfft.hh:873-1312)
// restated for registers, and the compile-time radix dispatch shared by the pass-per-launch
// (fft_global.hip) and LDS-tiled (fft_tiled.hip) kernels.  Radix 2/4/8 live in device_math.hh.
#pragma once

#include "device_math.hh"

namespace jst::dev {

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
// pocketfft pass7 (pocketfft.hh:1047-1122) without the output twiddles
template <bool FWD>
__device__ __forceinline__ void butterfly7(float2 (&x)[7]) {
    constexpr float sg = FWD ? -1.0f : 1.0f;
    constexpr float tw1r = 0.6234898018587335305250048840042398f,
                    tw1i = sg * 0.7818314824680298087084445266740578f,
                    tw2r = -0.2225209339563144042889025644967948f,
                    tw2i = sg * 0.9749279121818236070181316829939312f,
                    tw3r = -0.9009688679024191262361023195074451f,
                    tw3i = sg * 0.433883739117558120475768332848359f;
    const float2 t1 = x[0];
    const float2 t2 = cadd(x[1], x[6]), t7 = csub(x[1], x[6]);
    const float2 t3 = cadd(x[2], x[5]), t6 = csub(x[2], x[5]);
    const float2 t4 = cadd(x[3], x[4]), t5 = csub(x[3], x[4]);
    x[0] = mk(t1.x + t2.x + t3.x + t4.x, t1.y + t2.y + t3.y + t4.y);
#define JST_STEP7(u1, u2, x1, x2, x3, y1, y2, y3)                            \
    {                                                                        \
        float2 ca, cb;                                                       \
        ca.x = t1.x + x1 * t2.x + x2 * t3.x + x3 * t4.x;                     \
        ca.y = t1.y + x1 * t2.y + x2 * t3.y + x3 * t4.y;                     \
        cb.y = y1 * t7.x y2 * t6.x y3 * t5.x;                                \
        cb.x = -(y1 * t7.y y2 * t6.y y3 * t5.y);                             \
        x[u1] = cadd(ca, cb);                                                \
        x[u2] = csub(ca, cb);                                                \
    }
    JST_STEP7(1, 6, tw1r, tw2r, tw3r, +tw1i, +tw2i, +tw3i)
    JST_STEP7(2, 5, tw2r, tw3r, tw1r, +tw2i, -tw3i, -tw1i)
    JST_STEP7(3, 4, tw3r, tw1r, tw2r, +tw3i, -tw1i, +tw2i)
#undef JST_STEP7
}
// pocketfft pass11 (pocketfft.hh:1226-1312) without the output twiddles
template <bool FWD>
__device__ __forceinline__ void butterfly11(float2 (&x)[11]) {
    constexpr float sg = FWD ? -1.0f : 1.0f;
    constexpr float tw1r = 0.8412535328311811688618116489193677f,
                    tw1i = sg * 0.5406408174555975821076359543186917f,
                    tw2r = 0.4154150130018864255292741492296232f,
                    tw2i = sg * 0.9096319953545183714117153830790285f,
                    tw3r = -0.1423148382732851404437926686163697f,
                    tw3i = sg * 0.9898214418809327323760920377767188f,
                    tw4r = -0.6548607339452850640569250724662936f,
                    tw4i = sg * 0.7557495743542582837740358439723444f,
                    tw5r = -0.9594929736144973898903680570663277f,
                    tw5i = sg * 0.2817325568414296977114179153466169f;
    const float2 t1 = x[0];
    const float2 t2 = cadd(x[1], x[10]), t11 = csub(x[1], x[10]);
    const float2 t3 = cadd(x[2], x[9]), t10 = csub(x[2], x[9]);
    const float2 t4 = cadd(x[3], x[8]), t9 = csub(x[3], x[8]);
    const float2 t5 = cadd(x[4], x[7]), t8 = csub(x[4], x[7]);
    const float2 t6 = cadd(x[5], x[6]), t7 = csub(x[5], x[6]);
    x[0] = mk(t1.x + t2.x + t3.x + t4.x + t5.x + t6.x, t1.y + t2.y + t3.y + t4.y + t5.y + t6.y);
#define JST_STEP11(u1, u2, x1, x2, x3, x4, x5, y1, y2, y3, y4, y5)                           \
    {                                                                                        \
        float2 ca, cb;                                                                       \
        ca.x = t1.x + t2.x * x1 + t3.x * x2 + t4.x * x3 + t5.x * x4 + t6.x * x5;             \
        ca.y = t1.y + t2.y * x1 + t3.y * x2 + t4.y * x3 + t5.y * x4 + t6.y * x5;             \
        cb.y = y1 * t11.x y2 * t10.x y3 * t9.x y4 * t8.x y5 * t7.x;                          \
        cb.x = -(y1 * t11.y y2 * t10.y y3 * t9.y y4 * t8.y y5 * t7.y);                       \
        x[u1] = cadd(ca, cb);                                                                \
        x[u2] = csub(ca, cb);                                                                \
    }
    JST_STEP11(1, 10, tw1r, tw2r, tw3r, tw4r, tw5r, +tw1i, +tw2i, +tw3i, +tw4i, +tw5i)
    JST_STEP11(2, 9, tw2r, tw4r, tw5r, tw3r, tw1r, +tw2i, +tw4i, -tw5i, -tw3i, -tw1i)
    JST_STEP11(3, 8, tw3r, tw5r, tw2r, tw1r, tw4r, +tw3i, -tw5i, -tw2i, +tw1i, +tw4i)
    JST_STEP11(4, 7, tw4r, tw3r, tw1r, tw5r, tw2r, +tw4i, -tw3i, +tw1i, +tw5i, -tw2i)
    JST_STEP11(5, 6, tw5r, tw1r, tw4r, tw2r, tw3r, +tw5i, -tw1i, +tw4i, -tw2i, +tw3i)
#undef JST_STEP11
}
template <int IP, bool FWD>
__device__ __forceinline__ void butterfly_any(float2 (&x)[IP]) {
    if constexpr (IP == 3) butterfly3<FWD>(x);
    else if constexpr (IP == 5) butterfly5<FWD>(x);
    else if constexpr (IP == 7) butterfly7<FWD>(x);
    else if constexpr (IP == 11) butterfly11<FWD>(x);
    else butterfly<IP, FWD>(x);
}

}  // namespace jst::dev
