/*
 * jst_oracle.c -- CPU restatement of the CyberEther (Jetstream) hot-path modules.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (cyberether_amd/, bench.py's
 * timed GPU leg) may call into this file.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and there only as the checker / CPU baseline.
 *
 * Every function restates, in plain C with IEEE-754 binary32/binary64 arithmetic and
 * NO fused multiply-add (build with -ffp-contract=off), the arithmetic of one reference
 * CPU module.  Citations are relative to /root/reference (CyberEther 1.9.1).
 *
 * Parity pinning: tests/test_oracle_*.py check this file against
 *   (1) the known-answer vectors in the reference's own module tests (tests/golden/*.json,
 *       transcribed with file:line citations), and
 *   (2) oracle/_ref/libref_pocketfft.so -- the reference's vendored pocketfft.hh compiled
 *       in place from /root/reference (bit-exact comparison of the FFT restatement).
 */
#define _POSIX_C_SOURCE 199309L /* clock_gettime for the bench driver at the end of this file */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define JST_PI 3.14159265358979323846 /* include/jetstream/types.hh:52-53 */

#define ORACLE_MAX_RANK 8

/* ------------------------------------------------------------------------------------------
 * Strided N-ary traversal.
 * Restates include/jetstream/tools/automatic_iterator.hh:108-343: elements are visited in
 * row-major order over the (common) shape; every tensor has its own element strides.  All
 * the specialised 1D/contiguous/2D/3D iterators there produce the same visiting order as the
 * generic coordinate-odometer at :207-231, which is what is restated here.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t rank;
    uint64_t shape[ORACLE_MAX_RANK];
    uint64_t coord[ORACLE_MAX_RANK];
} odometer_t;

static uint64_t odo_init(odometer_t* o, uint32_t rank, const uint64_t* shape) {
    uint64_t size = 1;
    o->rank = rank;
    for (uint32_t i = 0; i < rank; ++i) {
        o->shape[i] = shape[i];
        o->coord[i] = 0;
        size *= shape[i];
    }
    return size;
}

static uint64_t odo_offset(const odometer_t* o, const uint64_t* stride) {
    uint64_t off = 0;
    for (uint32_t i = 0; i < o->rank; ++i) off += o->coord[i] * stride[i];
    return off;
}

static void odo_step(odometer_t* o) {
    for (uint32_t axis = o->rank; axis-- > 0;) {
        if (o->coord[axis] + 1 < o->shape[axis]) {
            o->coord[axis]++;
            return;
        }
        o->coord[axis] = 0;
    }
}

/* ------------------------------------------------------------------------------------------
 * Window (Blackman).  src/domains/dsp/window/module_impl_native_cpu.cc:20-37.
 * F64 evaluation, symmetric (N-1) denominator, imag = +0; N == 1 -> (1, 0).
 * out: interleaved CF32[n].
 * ---------------------------------------------------------------------------------------- */
void jst_oracle_window(float* out, uint64_t n) {
    if (n == 1) {
        out[0] = 1.0f;
        out[1] = 0.0f;
        return;
    }
    for (uint64_t i = 0; i < n; ++i) {
        const double tap = 0.42 - 0.50 * cos(2.0 * JST_PI * i / (n - 1)) +
                           0.08 * cos(4.0 * JST_PI * i / (n - 1));
        out[2 * i] = (float)tap;
        out[2 * i + 1] = 0.0f;
    }
}

/* ------------------------------------------------------------------------------------------
 * Invert.  src/domains/dsp/invert/module_impl_native_cpu.cc:79-103.
 * Flat index 'index' runs over the (contiguous-order) element sequence; the coordinate on the
 * resolved sample axis is (index / inner) % length.  Even length: negate odd coordinates
 * (unary minus flips the sign of BOTH parts, so +0 imag becomes -0).  Odd length: multiply by
 * the F64-evaluated phasor exp(j*2*pi*floor(length/2)*n/length) cast to F32, using the
 * std::complex<float> product (re = a*c - b*d, im = a*d + b*c; see multiply below).
 * in/out interleaved CF32, contiguous, 'count' elements.  f32 variant promotes to (x, +0).
 * ---------------------------------------------------------------------------------------- */
static void cmul_f32(float a, float b, float c, float d, float* re, float* im);

void jst_oracle_invert_cf32(const float* in, float* out, uint64_t count, uint64_t inner,
                            uint64_t length) {
    for (uint64_t index = 0; index < count; ++index) {
        const uint64_t coord = (index / inner) % length;
        const float re = in[2 * index], im = in[2 * index + 1];
        if ((length & 1ull) == 0) {
            if (coord & 1ull) {
                out[2 * index] = -re;
                out[2 * index + 1] = -im;
            } else {
                out[2 * index] = re;
                out[2 * index + 1] = im;
            }
        } else {
            const double phase =
                2.0 * JST_PI * (double)(length / 2) * (double)coord / (double)length;
            cmul_f32(re, im, (float)cos(phase), (float)sin(phase), &out[2 * index],
                     &out[2 * index + 1]);
        }
    }
}

void jst_oracle_invert_f32(const float* in, float* out, uint64_t count, uint64_t inner,
                           uint64_t length) {
    for (uint64_t index = 0; index < count; ++index) {
        const uint64_t coord = (index / inner) % length;
        const float re = in[index], im = 0.0f;
        if ((length & 1ull) == 0) {
            if (coord & 1ull) {
                out[2 * index] = -re;
                out[2 * index + 1] = -im;
            } else {
                out[2 * index] = re;
                out[2 * index + 1] = im;
            }
        } else {
            const double phase =
                2.0 * JST_PI * (double)(length / 2) * (double)coord / (double)length;
            cmul_f32(re, im, (float)cos(phase), (float)sin(phase), &out[2 * index],
                     &out[2 * index + 1]);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Multiply (broadcast, strided).  src/domains/core/multiply/module_impl.cc:10-84 (broadcast
 * views: stride 0 on expanded axes) and module_impl_native_cpu.cc:86-100 (c = a * b).
 * CF32 product is std::complex<float>::operator*, i.e. libgcc __mulsc3: the four products
 * ac, bd, ad, bc are rounded individually, re = ac - bd, im = ad + bc, and only if BOTH
 * results are NaN is the C99 Annex G infinity-recovery path taken.  Restated here for finite
 * inputs plus the recovery path (so inf/NaN inputs also match).
 * Strides are in ELEMENTS (src/memory/tensor.cc:94-109).
 * ---------------------------------------------------------------------------------------- */
static void cmul_f32(float a, float b, float c, float d, float* re, float* im) {
    float ac = a * c, bd = b * d, ad = a * d, bc = b * c;
    float x = ac - bd, y = ad + bc;
    if (isnan(x) && isnan(y)) {
        int recalc = 0;
        if (isinf(a) || isinf(b)) {
            a = copysignf(isinf(a) ? 1.0f : 0.0f, a);
            b = copysignf(isinf(b) ? 1.0f : 0.0f, b);
            if (isnan(c)) c = copysignf(0.0f, c);
            if (isnan(d)) d = copysignf(0.0f, d);
            recalc = 1;
        }
        if (isinf(c) || isinf(d)) {
            c = copysignf(isinf(c) ? 1.0f : 0.0f, c);
            d = copysignf(isinf(d) ? 1.0f : 0.0f, d);
            if (isnan(a)) a = copysignf(0.0f, a);
            if (isnan(b)) b = copysignf(0.0f, b);
            recalc = 1;
        }
        if (!recalc && (isinf(ac) || isinf(bd) || isinf(ad) || isinf(bc))) {
            if (isnan(a)) a = copysignf(0.0f, a);
            if (isnan(b)) b = copysignf(0.0f, b);
            if (isnan(c)) c = copysignf(0.0f, c);
            if (isnan(d)) d = copysignf(0.0f, d);
            recalc = 1;
        }
        if (recalc) {
            x = INFINITY * (a * c - b * d);
            y = INFINITY * (a * d + b * c);
        }
    }
    *re = x;
    *im = y;
}

void jst_oracle_multiply_cf32(uint32_t rank, const uint64_t* shape, const float* a,
                              const uint64_t* sa, const float* b, const uint64_t* sb, float* c,
                              const uint64_t* sc) {
    odometer_t o;
    const uint64_t size = odo_init(&o, rank, shape);
    for (uint64_t i = 0; i < size; ++i) {
        const uint64_t ia = odo_offset(&o, sa), ib = odo_offset(&o, sb), ic = odo_offset(&o, sc);
        cmul_f32(a[2 * ia], a[2 * ia + 1], b[2 * ib], b[2 * ib + 1], &c[2 * ic], &c[2 * ic + 1]);
        odo_step(&o);
    }
}

void jst_oracle_multiply_f32(uint32_t rank, const uint64_t* shape, const float* a,
                             const uint64_t* sa, const float* b, const uint64_t* sb, float* c,
                             const uint64_t* sc) {
    odometer_t o;
    const uint64_t size = odo_init(&o, rank, shape);
    for (uint64_t i = 0; i < size; ++i) {
        c[odo_offset(&o, sc)] = a[odo_offset(&o, sa)] * b[odo_offset(&o, sb)];
        odo_step(&o);
    }
}

/* ------------------------------------------------------------------------------------------
 * FFT, complex-to-complex, single precision, UNNORMALISED in both directions.
 * The reference calls pocketfft::c2c(..., fct = 1.0f) (src/domains/dsp/fft/
 * module_impl_native_cpu.cc:125-140) on its vendored src/domains/dsp/fft/pocketfft.hh
 * (in-tree, BSD-3; so no unpinned third-party dependency).  For lengths whose only prime
 * factor is 2 the plan is cfftp (pocketfft.hh:819-1560):
 *   factorize()   :1476-1497  -> 8s first, then 4s, then one 2 moved to the FRONT
 *   comp_twiddle():1513-1535  -> tw[(j-1)*(ido-1)+i-1] = twiddle[j*l1*i]
 *   sincos_2pibyn :296-372    -> twiddles from two F64 tables multiplied in F64, cast to F32
 *   pass_all      :1420-1468  -> Stockham passes ping-ponging between c and ch
 *   pass2/4/8     :843-872, :929-975, :1141-1223 with special_mul<fwd> (:266-272, conj for fwd)
 * plus pass3 :873-923 and pass5 :976-1050 for lengths with factors 3 and 5 (the Filter block's
 * convolution sizes, e.g. 160000 = 8*8*4*5*5*5*5).  Lengths with any other prime factor (pass7,
 * pass11, passg, Bluestein) are checked against oracle/_ref only.  Verified BIT-EXACT against
 * oracle/_ref for every 2^m, m in 0..16, and a sweep of 2^a 3^b 5^c lengths, both directions
 * (tests/test_oracle_fft.py).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    double r, i;
} c64;
typedef struct {
    float r, i;
} c32;

typedef struct {
    uint64_t N, mask, shift;
    c64 *v1, *v2;
} sincos_t;

static c64 sincos_calc(uint64_t x, uint64_t n, double ang) { /* pocketfft.hh:303-339 */
    c64 o;
    x <<= 3;
    if (x < 4 * n) {
        if (x < 2 * n) {
            if (x < n) {
                o.r = cos((double)x * ang);
                o.i = sin((double)x * ang);
                return o;
            }
            o.r = sin((double)(2 * n - x) * ang);
            o.i = cos((double)(2 * n - x) * ang);
            return o;
        } else {
            x -= 2 * n;
            if (x < n) {
                o.r = -sin((double)x * ang);
                o.i = cos((double)x * ang);
                return o;
            }
            o.r = -cos((double)(2 * n - x) * ang);
            o.i = sin((double)(2 * n - x) * ang);
            return o;
        }
    } else {
        x = 8 * n - x;
        if (x < 2 * n) {
            if (x < n) {
                o.r = cos((double)x * ang);
                o.i = -sin((double)x * ang);
                return o;
            }
            o.r = sin((double)(2 * n - x) * ang);
            o.i = -cos((double)(2 * n - x) * ang);
            return o;
        } else {
            x -= 2 * n;
            if (x < n) {
                o.r = -sin((double)x * ang);
                o.i = -cos((double)x * ang);
                return o;
            }
            o.r = -cos((double)(2 * n - x) * ang);
            o.i = -sin((double)(2 * n - x) * ang);
            return o;
        }
    }
}

static void sincos_init(sincos_t* s, uint64_t n) { /* pocketfft.hh:342-360 */
    const long double pi = 3.141592653589793238462643383279502884197L;
    const double ang = (double)(0.25L * pi / n);
    const uint64_t nval = (n + 2) / 2;
    s->N = n;
    s->shift = 1;
    while ((((uint64_t)1) << s->shift) * (((uint64_t)1) << s->shift) < nval) ++s->shift;
    s->mask = (((uint64_t)1) << s->shift) - 1;
    const uint64_t n1 = s->mask + 1;
    const uint64_t n2 = (nval + s->mask) / (s->mask + 1);
    s->v1 = (c64*)malloc(n1 * sizeof(c64));
    s->v2 = (c64*)malloc(n2 * sizeof(c64));
    s->v1[0].r = 1.0;
    s->v1[0].i = 0.0;
    for (uint64_t i = 1; i < n1; ++i) s->v1[i] = sincos_calc(i, n, ang);
    s->v2[0].r = 1.0;
    s->v2[0].i = 0.0;
    for (uint64_t i = 1; i < n2; ++i) s->v2[i] = sincos_calc(i * (s->mask + 1), n, ang);
}

static c32 sincos_get(const sincos_t* s, uint64_t idx) { /* pocketfft.hh:362-372 */
    c32 o;
    if (2 * idx <= s->N) {
        const c64 x1 = s->v1[idx & s->mask], x2 = s->v2[idx >> s->shift];
        o.r = (float)(x1.r * x2.r - x1.i * x2.i);
        o.i = (float)(x1.r * x2.i + x1.i * x2.r);
        return o;
    }
    idx = s->N - idx;
    const c64 x1 = s->v1[idx & s->mask], x2 = s->v2[idx >> s->shift];
    o.r = (float)(x1.r * x2.r - x1.i * x2.i);
    o.i = -(float)(x1.r * x2.i + x1.i * x2.r);
    return o;
}

static void sincos_free(sincos_t* s) {
    free(s->v1);
    free(s->v2);
}

/* Fills tw[n] with exp(+j*2*pi*k/n) exactly as pocketfft's sincos_2pibyn<float>(n)[k].
 * Exposed so tests can pin the product's twiddle generator against the same table. */
void jst_oracle_fft_twiddles(float* tw, uint64_t n) {
    sincos_t s;
    sincos_init(&s, n);
    for (uint64_t k = 0; k < n; ++k) {
        const c32 w = sincos_get(&s, k);
        tw[2 * k] = w.r;
        tw[2 * k + 1] = w.i;
    }
    sincos_free(&s);
}

/* Factor list (pocketfft.hh:1476-1497): 8s, 4s, one 2 moved to the FRONT, then odd divisors in
 * increasing order.  Returns the count (any prime may appear: 7 and 11 have dedicated passes,
 * larger ones use passg).  Whether this plan or Bluestein is used: jst_oracle_fft_bluestein_size. */
int jst_oracle_fft_factors(uint64_t n, uint32_t* fact) {
    int nf = 0;
    uint64_t len = n;
    if (len <= 1) return 0;
    while ((len & 7) == 0) {
        fact[nf++] = 8;
        len >>= 3;
    }
    while ((len & 3) == 0) {
        fact[nf++] = 4;
        len >>= 2;
    }
    if ((len & 1) == 0) {
        len >>= 1;
        fact[nf++] = 2;
        const uint32_t t = fact[0];
        fact[0] = fact[nf - 1];
        fact[nf - 1] = t;
    }
    for (uint64_t divisor = 3; divisor * divisor <= len; divisor += 2)
        while ((len % divisor) == 0) {
            fact[nf++] = (uint32_t)divisor;
            len /= divisor;
        }
    if (len > 1) fact[nf++] = (uint32_t)len;
    return nf;
}

static inline c32 cadd(c32 a, c32 b) {
    c32 o = {a.r + b.r, a.i + b.i};
    return o;
}
static inline c32 csub(c32 a, c32 b) {
    c32 o = {a.r - b.r, a.i - b.i};
    return o;
}
/* special_mul<fwd> (pocketfft.hh:266-272): forward multiplies by conj(w). */
static inline c32 special_mul(c32 v, c32 w, int fwd) {
    c32 o;
    if (fwd) {
        o.r = v.r * w.r + v.i * w.i;
        o.i = v.i * w.r - v.r * w.i;
    } else {
        o.r = v.r * w.r - v.i * w.i;
        o.i = v.r * w.i + v.i * w.r;
    }
    return o;
}
/* ROTX90<fwd> (pocketfft.hh:290-291): fwd -> multiply by -j, bwd -> by +j. */
static inline c32 rotx90(c32 a, int fwd) {
    c32 o;
    if (fwd) {
        o.r = a.i;
        o.i = -a.r;
    } else {
        o.r = -a.i;
        o.i = a.r;
    }
    return o;
}
/* ROTX45 / ROTX135 (pocketfft.hh:1124-1139). */
static inline c32 rotx45(c32 a, int fwd) {
    const float hsqt2 = (float)0.707106781186547524400844362104849L;
    c32 o;
    if (fwd) {
        o.r = hsqt2 * (a.r + a.i);
        o.i = hsqt2 * (a.i - a.r);
    } else {
        o.r = hsqt2 * (a.r - a.i);
        o.i = hsqt2 * (a.i + a.r);
    }
    return o;
}
static inline c32 rotx135(c32 a, int fwd) {
    const float hsqt2 = (float)0.707106781186547524400844362104849L;
    c32 o;
    if (fwd) {
        o.r = hsqt2 * (a.i - a.r);
        o.i = hsqt2 * (-a.r - a.i);
    } else {
        o.r = hsqt2 * (-a.r - a.i);
        o.i = hsqt2 * (a.r - a.i);
    }
    return o;
}

#define CC(a, b, c) cc[(a) + ido * ((b) + ip * (c))]
#define CH(a, b, c) ch[(a) + ido * ((b) + l1 * (c))]
#define WA(x, i) wa[(i)-1 + (x) * (ido - 1)]

static void pass2(uint64_t ido, uint64_t l1, const c32* cc, c32* ch, const c32* wa, int fwd) {
    const uint64_t ip = 2; /* pocketfft.hh:843-872 */
    for (uint64_t k = 0; k < l1; ++k) {
        CH(0, k, 0) = cadd(CC(0, 0, k), CC(0, 1, k));
        CH(0, k, 1) = csub(CC(0, 0, k), CC(0, 1, k));
        for (uint64_t i = 1; i < ido; ++i) {
            CH(i, k, 0) = cadd(CC(i, 0, k), CC(i, 1, k));
            CH(i, k, 1) = special_mul(csub(CC(i, 0, k), CC(i, 1, k)), WA(0, i), fwd);
        }
    }
}

static void pass4(uint64_t ido, uint64_t l1, const c32* cc, c32* ch, const c32* wa, int fwd) {
    const uint64_t ip = 4; /* pocketfft.hh:929-975 */
    for (uint64_t k = 0; k < l1; ++k) {
        for (uint64_t i = 0; i < ido; ++i) {
            c32 t1, t2, t3, t4;
            const c32 cc0 = CC(i, 0, k), cc1 = CC(i, 1, k), cc2 = CC(i, 2, k), cc3 = CC(i, 3, k);
            t2 = cadd(cc0, cc2);
            t1 = csub(cc0, cc2);
            t3 = cadd(cc1, cc3);
            t4 = csub(cc1, cc3);
            t4 = rotx90(t4, fwd);
            if (i == 0) {
                CH(0, k, 0) = cadd(t2, t3);
                CH(0, k, 2) = csub(t2, t3);
                CH(0, k, 1) = cadd(t1, t4);
                CH(0, k, 3) = csub(t1, t4);
            } else {
                CH(i, k, 0) = cadd(t2, t3);
                CH(i, k, 1) = special_mul(cadd(t1, t4), WA(0, i), fwd);
                CH(i, k, 2) = special_mul(csub(t2, t3), WA(1, i), fwd);
                CH(i, k, 3) = special_mul(csub(t1, t4), WA(2, i), fwd);
            }
        }
    }
}

static void pass8(uint64_t ido, uint64_t l1, const c32* cc, c32* ch, const c32* wa, int fwd) {
    const uint64_t ip = 8; /* pocketfft.hh:1141-1223 */
    for (uint64_t k = 0; k < l1; ++k) {
        for (uint64_t i = 0; i < ido; ++i) {
            c32 a0, a1, a2, a3, a4, a5, a6, a7, t;
            /* PM(a1,a5,CC1,CC5); PM(a3,a7,CC3,CC7) */
            a1 = cadd(CC(i, 1, k), CC(i, 5, k));
            a5 = csub(CC(i, 1, k), CC(i, 5, k));
            a3 = cadd(CC(i, 3, k), CC(i, 7, k));
            a7 = csub(CC(i, 3, k), CC(i, 7, k));
            /* The i==0 and i>0 bodies order ROTX90(a7) differently relative to
             * PMINPLACE(a1,a3), but those touch disjoint values: same arithmetic. */
            a7 = rotx90(a7, fwd);
            t = a1; /* PMINPLACE(a1,a3) */
            a1 = cadd(a1, a3);
            a3 = csub(t, a3);
            a3 = rotx90(a3, fwd);
            t = a5; /* PMINPLACE(a5,a7) */
            a5 = cadd(a5, a7);
            a7 = csub(t, a7);
            a5 = rotx45(a5, fwd);
            a7 = rotx135(a7, fwd);
            a0 = cadd(CC(i, 0, k), CC(i, 4, k));
            a4 = csub(CC(i, 0, k), CC(i, 4, k));
            a2 = cadd(CC(i, 2, k), CC(i, 6, k));
            a6 = csub(CC(i, 2, k), CC(i, 6, k));
            if (i == 0) {
                /* pocketfft.hh:1166-1172 */
                const c32 s02 = cadd(a0, a2), d02 = csub(a0, a2);
                CH(0, k, 0) = cadd(s02, a1);
                CH(0, k, 4) = csub(s02, a1);
                CH(0, k, 2) = cadd(d02, a3);
                CH(0, k, 6) = csub(d02, a3);
                a6 = rotx90(a6, fwd);
                const c32 s46 = cadd(a4, a6), d46 = csub(a4, a6);
                CH(0, k, 1) = cadd(s46, a5);
                CH(0, k, 5) = csub(s46, a5);
                CH(0, k, 3) = cadd(d46, a7);
                CH(0, k, 7) = csub(d46, a7);
            } else {
                /* pocketfft.hh:1210-1220 */
                t = a0; /* PMINPLACE(a0,a2) */
                a0 = cadd(a0, a2);
                a2 = csub(t, a2);
                CH(i, k, 0) = cadd(a0, a1);
                CH(i, k, 4) = special_mul(csub(a0, a1), WA(3, i), fwd);
                CH(i, k, 2) = special_mul(cadd(a2, a3), WA(1, i), fwd);
                CH(i, k, 6) = special_mul(csub(a2, a3), WA(5, i), fwd);
                a6 = rotx90(a6, fwd);
                t = a4; /* PMINPLACE(a4,a6) */
                a4 = cadd(a4, a6);
                a6 = csub(t, a6);
                CH(i, k, 1) = special_mul(cadd(a4, a5), WA(0, i), fwd);
                CH(i, k, 5) = special_mul(csub(a4, a5), WA(4, i), fwd);
                CH(i, k, 3) = special_mul(cadd(a6, a7), WA(2, i), fwd);
                CH(i, k, 7) = special_mul(csub(a6, a7), WA(6, i), fwd);
            }
        }
    }
}
static void pass3(uint64_t ido, uint64_t l1, const c32* cc, c32* ch, const c32* wa, int fwd) {
    const uint64_t ip = 3; /* pocketfft.hh:873-923 */
    const float tw1r = -0.5f;
    const float tw1i = (fwd ? -1 : 1) * (float)0.8660254037844386467637231707529362L;
    for (uint64_t k = 0; k < l1; ++k)
        for (uint64_t i = 0; i < ido; ++i) {
            const c32 t0 = CC(i, 0, k);
            const c32 t1 = cadd(CC(i, 1, k), CC(i, 2, k)), t2 = csub(CC(i, 1, k), CC(i, 2, k));
            CH(i, k, 0) = cadd(t0, t1);
            c32 ca = {t0.r + t1.r * tw1r, t0.i + t1.i * tw1r};
            c32 cb = {-t2.i * tw1i, t2.r * tw1i};
            if (i == 0) {
                CH(0, k, 1) = cadd(ca, cb);
                CH(0, k, 2) = csub(ca, cb);
            } else {
                CH(i, k, 1) = special_mul(cadd(ca, cb), WA(0, i), fwd);
                CH(i, k, 2) = special_mul(csub(ca, cb), WA(1, i), fwd);
            }
        }
}

static void pass5(uint64_t ido, uint64_t l1, const c32* cc, c32* ch, const c32* wa, int fwd) {
    const uint64_t ip = 5; /* pocketfft.hh:976-1050 */
    const float tw1r = (float)0.3090169943749474241022934171828191L;
    const float tw1i = (fwd ? -1 : 1) * (float)0.9510565162951535721164393333793821L;
    const float tw2r = (float)-0.8090169943749474241022934171828191L;
    const float tw2i = (fwd ? -1 : 1) * (float)0.5877852522924731291687059546390728L;
    for (uint64_t k = 0; k < l1; ++k)
        for (uint64_t i = 0; i < ido; ++i) {
            const c32 t0 = CC(i, 0, k);
            const c32 t1 = cadd(CC(i, 1, k), CC(i, 4, k)), t4 = csub(CC(i, 1, k), CC(i, 4, k));
            const c32 t2 = cadd(CC(i, 2, k), CC(i, 3, k)), t3 = csub(CC(i, 2, k), CC(i, 3, k));
            c32 o0 = {t0.r + t1.r + t2.r, t0.i + t1.i + t2.i};
            CH(i, k, 0) = o0;
            /* PARTSTEP5(1,4, tw1r,tw2r,+tw1i,+tw2i) */
            c32 ca = {t0.r + tw1r * t1.r + tw2r * t2.r, t0.i + tw1r * t1.i + tw2r * t2.i};
            c32 cb;
            cb.i = +tw1i * t4.r + tw2i * t3.r;
            cb.r = -(+tw1i * t4.i + tw2i * t3.i);
            /* PARTSTEP5(2,3, tw2r,tw1r,+tw2i,-tw1i) */
            c32 da = {t0.r + tw2r * t1.r + tw1r * t2.r, t0.i + tw2r * t1.i + tw1r * t2.i};
            c32 db;
            db.i = +tw2i * t4.r - tw1i * t3.r;
            db.r = -(+tw2i * t4.i - tw1i * t3.i);
            if (i == 0) {
                CH(0, k, 1) = cadd(ca, cb);
                CH(0, k, 4) = csub(ca, cb);
                CH(0, k, 2) = cadd(da, db);
                CH(0, k, 3) = csub(da, db);
            } else {
                CH(i, k, 1) = special_mul(cadd(ca, cb), WA(0, i), fwd);
                CH(i, k, 4) = special_mul(csub(ca, cb), WA(3, i), fwd);
                CH(i, k, 2) = special_mul(cadd(da, db), WA(1, i), fwd);
                CH(i, k, 3) = special_mul(csub(da, db), WA(2, i), fwd);
            }
        }
}
/* pass7 (pocketfft.hh:1047-1122).  Sums are left to right, exactly as the macros expand. */
static void pass7(uint64_t ido, uint64_t l1, const c32* cc, c32* ch, const c32* wa, int fwd) {
    const uint64_t ip = 7;
    const float sg = fwd ? -1.0f : 1.0f;
    const float tw1r = (float)0.6234898018587335305250048840042398L,
                tw1i = sg * (float)0.7818314824680298087084445266740578L,
                tw2r = (float)-0.2225209339563144042889025644967948L,
                tw2i = sg * (float)0.9749279121818236070181316829939312L,
                tw3r = (float)-0.9009688679024191262361023195074451L,
                tw3i = sg * (float)0.433883739117558120475768332848359L;
    for (uint64_t k = 0; k < l1; ++k)
        for (uint64_t i = 0; i < ido; ++i) {
            const c32 t1 = CC(i, 0, k);
            const c32 t2 = cadd(CC(i, 1, k), CC(i, 6, k)), t7 = csub(CC(i, 1, k), CC(i, 6, k));
            const c32 t3 = cadd(CC(i, 2, k), CC(i, 5, k)), t6 = csub(CC(i, 2, k), CC(i, 5, k));
            const c32 t4 = cadd(CC(i, 3, k), CC(i, 4, k)), t5 = csub(CC(i, 3, k), CC(i, 4, k));
            c32 o0 = {t1.r + t2.r + t3.r + t4.r, t1.i + t2.i + t3.i + t4.i};
            CH(i, k, 0) = o0;
#define JST_STEP7(u1, u2, x1, x2, x3, y1, y2, y3)                                            \
    {                                                                                        \
        c32 ca, cb;                                                                          \
        ca.r = t1.r + x1 * t2.r + x2 * t3.r + x3 * t4.r;                                     \
        ca.i = t1.i + x1 * t2.i + x2 * t3.i + x3 * t4.i;                                     \
        cb.i = y1 * t7.r y2 * t6.r y3 * t5.r;                                                \
        cb.r = -(y1 * t7.i y2 * t6.i y3 * t5.i);                                             \
        if (i == 0) {                                                                        \
            CH(0, k, u1) = cadd(ca, cb);                                                     \
            CH(0, k, u2) = csub(ca, cb);                                                     \
        } else {                                                                             \
            CH(i, k, u1) = special_mul(cadd(ca, cb), WA(u1 - 1, i), fwd);                    \
            CH(i, k, u2) = special_mul(csub(ca, cb), WA(u2 - 1, i), fwd);                    \
        }                                                                                    \
    }
            JST_STEP7(1, 6, tw1r, tw2r, tw3r, +tw1i, +tw2i, +tw3i)
            JST_STEP7(2, 5, tw2r, tw3r, tw1r, +tw2i, -tw3i, -tw1i)
            JST_STEP7(3, 4, tw3r, tw1r, tw2r, +tw3i, -tw1i, +tw2i)
#undef JST_STEP7
        }
}

/* pass11 (pocketfft.hh:1226-1312). */
static void pass11(uint64_t ido, uint64_t l1, const c32* cc, c32* ch, const c32* wa, int fwd) {
    const uint64_t ip = 11;
    const float sg = fwd ? -1.0f : 1.0f;
    const float tw1r = (float)0.8412535328311811688618116489193677L,
                tw1i = sg * (float)0.5406408174555975821076359543186917L,
                tw2r = (float)0.4154150130018864255292741492296232L,
                tw2i = sg * (float)0.9096319953545183714117153830790285L,
                tw3r = (float)-0.1423148382732851404437926686163697L,
                tw3i = sg * (float)0.9898214418809327323760920377767188L,
                tw4r = (float)-0.6548607339452850640569250724662936L,
                tw4i = sg * (float)0.7557495743542582837740358439723444L,
                tw5r = (float)-0.9594929736144973898903680570663277L,
                tw5i = sg * (float)0.2817325568414296977114179153466169L;
    for (uint64_t k = 0; k < l1; ++k)
        for (uint64_t i = 0; i < ido; ++i) {
            const c32 t1 = CC(i, 0, k);
            const c32 t2 = cadd(CC(i, 1, k), CC(i, 10, k)), t11 = csub(CC(i, 1, k), CC(i, 10, k));
            const c32 t3 = cadd(CC(i, 2, k), CC(i, 9, k)), t10 = csub(CC(i, 2, k), CC(i, 9, k));
            const c32 t4 = cadd(CC(i, 3, k), CC(i, 8, k)), t9 = csub(CC(i, 3, k), CC(i, 8, k));
            const c32 t5 = cadd(CC(i, 4, k), CC(i, 7, k)), t8 = csub(CC(i, 4, k), CC(i, 7, k));
            const c32 t6 = cadd(CC(i, 5, k), CC(i, 6, k)), t7 = csub(CC(i, 5, k), CC(i, 6, k));
            c32 o0 = {t1.r + t2.r + t3.r + t4.r + t5.r + t6.r, t1.i + t2.i + t3.i + t4.i + t5.i + t6.i};
            CH(i, k, 0) = o0;
#define JST_STEP11(u1, u2, x1, x2, x3, x4, x5, y1, y2, y3, y4, y5)                            \
    {                                                                                        \
        c32 ca, cb;                                                                          \
        ca.r = t1.r + t2.r * x1 + t3.r * x2 + t4.r * x3 + t5.r * x4 + t6.r * x5;             \
        ca.i = t1.i + t2.i * x1 + t3.i * x2 + t4.i * x3 + t5.i * x4 + t6.i * x5;             \
        cb.i = y1 * t11.r y2 * t10.r y3 * t9.r y4 * t8.r y5 * t7.r;                          \
        cb.r = -(y1 * t11.i y2 * t10.i y3 * t9.i y4 * t8.i y5 * t7.i);                       \
        if (i == 0) {                                                                        \
            CH(0, k, u1) = cadd(ca, cb);                                                     \
            CH(0, k, u2) = csub(ca, cb);                                                     \
        } else {                                                                             \
            CH(i, k, u1) = special_mul(cadd(ca, cb), WA(u1 - 1, i), fwd);                    \
            CH(i, k, u2) = special_mul(csub(ca, cb), WA(u2 - 1, i), fwd);                    \
        }                                                                                    \
    }
            JST_STEP11(1, 10, tw1r, tw2r, tw3r, tw4r, tw5r, +tw1i, +tw2i, +tw3i, +tw4i, +tw5i)
            JST_STEP11(2, 9, tw2r, tw4r, tw5r, tw3r, tw1r, +tw2i, +tw4i, -tw5i, -tw3i, -tw1i)
            JST_STEP11(3, 8, tw3r, tw5r, tw2r, tw1r, tw4r, +tw3i, -tw5i, -tw2i, +tw1i, +tw4i)
            JST_STEP11(4, 7, tw4r, tw3r, tw1r, tw5r, tw2r, +tw4i, -tw3i, +tw1i, +tw5i, -tw2i)
            JST_STEP11(5, 6, tw5r, tw1r, tw4r, tw2r, tw3r, +tw5i, -tw1i, +tw4i, -tw2i, +tw3i)
#undef JST_STEP11
        }
}
#undef CC
#undef CH
#undef WA

/* passg (pocketfft.hh:1314-1421): generic odd radix.  Per (i,k) the computation only touches
 * CC(i,*,k), so it is restated butterfly by butterfly with the same expression grouping; the
 * result is what the reference leaves in `cc` (it swaps once more afterwards), written here to
 * `ch` in CH(i,k,j) order.  csarr[j] = twiddle[j*l1*ido] (comp_twiddle, :1526-1531). */
static void passg(uint64_t ido, uint64_t ip, uint64_t l1, const c32* cc, c32* ch, const c32* wa,
                  const c32* csarr, int fwd) {
    const uint64_t ipph = (ip + 1) / 2;
    c32* wal = (c32*)malloc(ip * sizeof(c32));
    c32* h = (c32*)malloc(ip * sizeof(c32));
    c32* x = (c32*)malloc(ip * sizeof(c32));
    wal[0].r = 1.0f;
    wal[0].i = 0.0f;
    for (uint64_t i = 1; i < ip; ++i) {
        wal[i].r = csarr[i].r;
        wal[i].i = fwd ? -csarr[i].i : csarr[i].i;
    }
    for (uint64_t k = 0; k < l1; ++k)
        for (uint64_t i = 0; i < ido; ++i) {
            h[0] = cc[i + ido * (0 + ip * k)];
            for (uint64_t j = 1, jc = ip - 1; j < ipph; ++j, --jc) {
                const c32 a = cc[i + ido * (j + ip * k)], b = cc[i + ido * (jc + ip * k)];
                h[j] = cadd(a, b);
                h[jc] = csub(a, b);
            }
            c32 tmp = h[0];
            for (uint64_t j = 1; j < ipph; ++j) {
                tmp.r += h[j].r;
                tmp.i += h[j].i;
            }
            x[0] = tmp;
            for (uint64_t l = 1, lc = ip - 1; l < ipph; ++l, --lc) {
                c32 xl, xlc;
                xl.r = h[0].r + wal[l].r * h[1].r + wal[2 * l].r * h[2].r;
                xl.i = h[0].i + wal[l].r * h[1].i + wal[2 * l].r * h[2].i;
                xlc.r = -wal[l].i * h[ip - 1].i - wal[2 * l].i * h[ip - 2].i;
                xlc.i = wal[l].i * h[ip - 1].r + wal[2 * l].i * h[ip - 2].r;
                uint64_t iwal = 2 * l;
                uint64_t j = 3, jc = ip - 3;
                for (; j < ipph - 1; j += 2, jc -= 2) {
                    iwal += l;
                    if (iwal > ip) iwal -= ip;
                    const c32 xwal = wal[iwal];
                    iwal += l;
                    if (iwal > ip) iwal -= ip;
                    const c32 xwal2 = wal[iwal];
                    xl.r += h[j].r * xwal.r + h[j + 1].r * xwal2.r;
                    xl.i += h[j].i * xwal.r + h[j + 1].i * xwal2.r;
                    xlc.r -= h[jc].i * xwal.i + h[jc - 1].i * xwal2.i;
                    xlc.i += h[jc].r * xwal.i + h[jc - 1].r * xwal2.i;
                }
                for (; j < ipph; ++j, --jc) {
                    iwal += l;
                    if (iwal > ip) iwal -= ip;
                    const c32 xwal = wal[iwal];
                    xl.r += h[j].r * xwal.r;
                    xl.i += h[j].i * xwal.r;
                    xlc.r -= h[jc].i * xwal.i;
                    xlc.i += h[jc].r * xwal.i;
                }
                x[l] = xl;
                x[lc] = xlc;
            }
            ch[i + ido * (k + l1 * 0)] = x[0];
            for (uint64_t j = 1, jc = ip - 1; j < ipph; ++j, --jc) {
                const c32 s1 = cadd(x[j], x[jc]), s2 = csub(x[j], x[jc]);
                if (i == 0) {
                    ch[i + ido * (k + l1 * j)] = s1;
                    ch[i + ido * (k + l1 * jc)] = s2;
                } else {
                    ch[i + ido * (k + l1 * j)] = special_mul(s1, wa[(j - 1) * (ido - 1) + i - 1], fwd);
                    ch[i + ido * (k + l1 * jc)] = special_mul(s2, wa[(jc - 1) * (ido - 1) + i - 1], fwd);
                }
            }
        }
    free(wal);
    free(h);
    free(x);
}

/* cfftp plan (pocketfft.hh:1476-1547): factors + per-pass twiddles (+ csarr for ip > 11). */
typedef struct {
    uint64_t n;
    int nf;
    uint32_t fact[64];
    c32* mem;
    const c32* tw[64];
    const c32* tws[64];
} cfftp_t;

static int cfftp_init(cfftp_t* p, uint64_t n) {
    memset(p, 0, sizeof(*p));
    p->n = n;
    if (n == 0) return -1;
    if (n == 1) return 0;
    p->nf = jst_oracle_fft_factors(n, p->fact);
    if (p->nf < 0) return -1;
    uint64_t twsize = 0, l1 = 1;
    for (int k = 0; k < p->nf; ++k) {
        const uint64_t ip = p->fact[k], ido = n / (l1 * ip);
        twsize += (ip - 1) * (ido - 1);
        if (ip > 11) twsize += ip;
        l1 *= ip;
    }
    sincos_t s;
    sincos_init(&s, n);
    p->mem = (c32*)malloc((twsize + 1) * sizeof(c32));
    uint64_t memofs = 0;
    l1 = 1;
    for (int k = 0; k < p->nf; ++k) {
        const uint64_t ip = p->fact[k], ido = n / (l1 * ip);
        p->tw[k] = p->mem + memofs;
        for (uint64_t j = 1; j < ip; ++j)
            for (uint64_t i = 1; i < ido; ++i)
                p->mem[memofs + (j - 1) * (ido - 1) + i - 1] = sincos_get(&s, j * l1 * i);
        memofs += (ip - 1) * (ido - 1);
        if (ip > 11) {
            p->tws[k] = p->mem + memofs;
            for (uint64_t j = 0; j < ip; ++j) p->mem[memofs + j] = sincos_get(&s, j * l1 * ido);
            memofs += ip;
        }
        l1 *= ip;
    }
    sincos_free(&s);
    return 0;
}
static void cfftp_free(cfftp_t* p) { free(p->mem); }

/* pass_all with fct == 1 (pocketfft.hh:1423-1471): c is transformed in place, ch is scratch. */
static void cfftp_exec(const cfftp_t* p, c32* c, c32* ch, int forward) {
    const uint64_t n = p->n;
    if (n == 1) return;
    c32 *p1 = c, *p2 = ch;
    uint64_t l1 = 1;
    for (int k = 0; k < p->nf; ++k) {
        const uint64_t ip = p->fact[k], l2 = ip * l1, ido = n / l2;
        if (ip == 8) pass8(ido, l1, p1, p2, p->tw[k], forward);
        else if (ip == 4) pass4(ido, l1, p1, p2, p->tw[k], forward);
        else if (ip == 2) pass2(ido, l1, p1, p2, p->tw[k], forward);
        else if (ip == 3) pass3(ido, l1, p1, p2, p->tw[k], forward);
        else if (ip == 5) pass5(ido, l1, p1, p2, p->tw[k], forward);
        else if (ip == 7) pass7(ido, l1, p1, p2, p->tw[k], forward);
        else if (ip == 11) pass11(ido, l1, p1, p2, p->tw[k], forward);
        else passg(ido, ip, l1, p1, p2, p->tw[k], p->tws[k], forward);
        c32* t = p1;
        p1 = p2;
        p2 = t;
        l1 = l2;
    }
    if (p1 != c) memcpy(c, p1, n * sizeof(c32));
}

/* util::largest_prime_factor / cost_guess / good_size_cmplx (pocketfft.hh:372-428). */
static uint64_t largest_prime_factor(uint64_t n) {
    uint64_t res = 1;
    while ((n & 1) == 0) {
        res = 2;
        n >>= 1;
    }
    for (uint64_t x = 3; x * x <= n; x += 2)
        while ((n % x) == 0) {
            res = x;
            n /= x;
        }
    if (n > 1) res = n;
    return res;
}
static double cost_guess(uint64_t n) {
    const double lfp = 1.1;
    const uint64_t ni = n;
    double result = 0.;
    while ((n & 1) == 0) {
        result += 2;
        n >>= 1;
    }
    for (uint64_t x = 3; x * x <= n; x += 2)
        while ((n % x) == 0) {
            result += (x <= 5) ? (double)x : lfp * (double)x;
            n /= x;
        }
    if (n > 1) result += (n <= 5) ? (double)n : lfp * (double)n;
    return result * (double)ni;
}
static uint64_t good_size_cmplx(uint64_t n) {
    if (n <= 12) return n;
    uint64_t bestfac = 2 * n;
    for (uint64_t f11 = 1; f11 < bestfac; f11 *= 11)
        for (uint64_t f117 = f11; f117 < bestfac; f117 *= 7)
            for (uint64_t f1175 = f117; f1175 < bestfac; f1175 *= 5) {
                uint64_t x = f1175;
                while (x < n) x *= 2;
                for (;;) {
                    if (x < n)
                        x *= 3;
                    else if (x > n) {
                        if (x < bestfac) bestfac = x;
                        if (x & 1) break;
                        x >>= 1;
                    } else
                        return n;
                }
            }
    return bestfac;
}
/* pocketfft_c's plan choice (pocketfft.hh:2472-2489): 0 = cfftp of n; otherwise the Bluestein
 * convolution length n2. */
uint64_t jst_oracle_fft_bluestein_size(uint64_t n) {
    if (n == 0) return 0;
    const uint64_t lpf = (n < 50) ? 0 : largest_prime_factor(n);
    if (lpf * lpf <= n) return 0;
    const double comp1 = cost_guess(n);
    double comp2 = 2 * cost_guess(good_size_cmplx(2 * n - 1));
    comp2 *= 1.5;
    return (comp2 < comp1) ? good_size_cmplx(2 * n - 1) : 0;
}

/* fftblue (pocketfft.hh:2362-2432). */
typedef struct {
    uint64_t n, n2;
    cfftp_t plan;
    c32 *bk, *bkf;
} fftblue_t;
static int fftblue_init(fftblue_t* b, uint64_t n, uint64_t n2) {
    b->n = n;
    b->n2 = n2;
    if (cfftp_init(&b->plan, n2) != 0) return -1;
    b->bk = (c32*)malloc(n * sizeof(c32));
    b->bkf = (c32*)malloc((n2 / 2 + 1) * sizeof(c32));
    sincos_t tmp;
    sincos_init(&tmp, 2 * n);
    b->bk[0].r = 1.0f;
    b->bk[0].i = 0.0f;
    uint64_t coeff = 0;
    for (uint64_t m = 1; m < n; ++m) {
        coeff += 2 * m - 1;
        if (coeff >= 2 * n) coeff -= 2 * n;
        b->bk[m] = sincos_get(&tmp, coeff);
    }
    sincos_free(&tmp);
    c32* tbkf = (c32*)malloc(n2 * sizeof(c32));
    c32* scratch = (c32*)malloc(n2 * sizeof(c32));
    const float xn2 = 1.0f / (float)n2;
    tbkf[0].r = b->bk[0].r * xn2;
    tbkf[0].i = b->bk[0].i * xn2;
    for (uint64_t m = 1; m < n; ++m) {
        c32 v = {b->bk[m].r * xn2, b->bk[m].i * xn2};
        tbkf[m] = tbkf[n2 - m] = v;
    }
    for (uint64_t m = n; m <= (n2 - n); ++m) {
        tbkf[m].r = 0.0f;
        tbkf[m].i = 0.0f;
    }
    cfftp_exec(&b->plan, tbkf, scratch, 1);
    for (uint64_t i = 0; i < n2 / 2 + 1; ++i) b->bkf[i] = tbkf[i];
    free(tbkf);
    free(scratch);
    return 0;
}
static void fftblue_free(fftblue_t* b) {
    cfftp_free(&b->plan);
    free(b->bk);
    free(b->bkf);
}
static void fftblue_exec(const fftblue_t* b, c32* c, c32* akf, c32* scratch, int fwd) {
    const uint64_t n = b->n, n2 = b->n2;
    for (uint64_t m = 0; m < n; ++m) akf[m] = special_mul(c[m], b->bk[m], fwd);
    const c32 zero = {akf[0].r * 0.0f, akf[0].i * 0.0f}; /* akf[0]*T0(0): NaN/-0 propagate */
    for (uint64_t m = n; m < n2; ++m) akf[m] = zero;
    cfftp_exec(&b->plan, akf, scratch, 1);
    akf[0] = special_mul(akf[0], b->bkf[0], !fwd);
    for (uint64_t m = 1; m < (n2 + 1) / 2; ++m) {
        akf[m] = special_mul(akf[m], b->bkf[m], !fwd);
        akf[n2 - m] = special_mul(akf[n2 - m], b->bkf[m], !fwd);
    }
    if ((n2 & 1) == 0) akf[n2 / 2] = special_mul(akf[n2 / 2], b->bkf[n2 / 2], !fwd);
    cfftp_exec(&b->plan, akf, scratch, 0);
    for (uint64_t m = 0; m < n; ++m) {
        const c32 v = special_mul(akf[m], b->bk[m], fwd);
        c[m].r = v.r * 1.0f; /* *fct with fct == 1 */
        c[m].i = v.i * 1.0f;
    }
}

/* in/out: interleaved CF32, 'batch' contiguous rows of n; any n >= 1 (plan chosen like
 * pocketfft_c).  Returns 0, or -1 for n == 0. */
int jst_oracle_fft_c2c(const float* in, float* out, uint64_t n, uint64_t batch, int forward) {
    if (n == 0) return -1;
    if (n == 1) {
        memcpy(out, in, batch * 2 * sizeof(float));
        return 0;
    }
    const uint64_t n2 = jst_oracle_fft_bluestein_size(n);
    if (n2 != 0) {
        fftblue_t b;
        if (fftblue_init(&b, n, n2) != 0) return -1;
        c32* c = (c32*)malloc(n * sizeof(c32));
        c32* akf = (c32*)malloc(n2 * sizeof(c32));
        c32* scratch = (c32*)malloc(n2 * sizeof(c32));
        for (uint64_t r = 0; r < batch; ++r) {
            memcpy(c, in + 2 * r * n, n * sizeof(c32));
            fftblue_exec(&b, c, akf, scratch, forward);
            memcpy(out + 2 * r * n, c, n * sizeof(c32));
        }
        free(c);
        free(akf);
        free(scratch);
        fftblue_free(&b);
        return 0;
    }
    cfftp_t p;
    if (cfftp_init(&p, n) != 0) return -1;
    c32* c = (c32*)malloc(n * sizeof(c32));
    c32* ch = (c32*)malloc(n * sizeof(c32));
    for (uint64_t r = 0; r < batch; ++r) {
        memcpy(c, in + 2 * r * n, n * sizeof(c32));
        cfftp_exec(&p, c, ch, forward);
        memcpy(out + 2 * r * n, c, n * sizeof(c32));
    }
    free(c);
    free(ch);
    cfftp_free(&p);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Real transforms: pocketfft's rfftp (FFTPACK halfcomplex), pocketfft.hh:1553-2358, for plans with
 * radices 2, 3, 4, 5 (the generic radfg/radbg are not restated), plus fftblue::exec_r and the
 * pocketfft_r plan choice (:2362-2457, 2508-2530).  The FFT module's F32 paths:
 * r2r_fftpack(forward, forward) and r2c (module_impl_native_cpu.cc:142-167).
 * ---------------------------------------------------------------------------------------- */
#define RWA(x, i) wa[(i) + (x) * (ido - 1)]
#define MULPM(a, b, c, d, e, f) { a = (c) * (e) + (d) * (f); b = (c) * (f) - (d) * (e); }

static void radf2(uint64_t ido, uint64_t l1, const float* cc, float* ch, const float* wa) {
#define CC(a, b, c) cc[(a) + ido * ((b) + l1 * (c))]
#define CH(a, b, c) ch[(a) + ido * ((b) + 2 * (c))]
    for (uint64_t k = 0; k < l1; k++) {
        CH(0, 0, k) = CC(0, k, 0) + CC(0, k, 1);
        CH(ido - 1, 1, k) = CC(0, k, 0) - CC(0, k, 1);
    }
    if ((ido & 1) == 0)
        for (uint64_t k = 0; k < l1; k++) {
            CH(0, 1, k) = -CC(ido - 1, k, 1);
            CH(ido - 1, 0, k) = CC(ido - 1, k, 0);
        }
    if (ido <= 2) return;
    for (uint64_t k = 0; k < l1; k++)
        for (uint64_t i = 2; i < ido; i += 2) {
            const uint64_t ic = ido - i;
            float tr2, ti2;
            MULPM(tr2, ti2, RWA(0, i - 2), RWA(0, i - 1), CC(i - 1, k, 1), CC(i, k, 1))
            CH(i - 1, 0, k) = CC(i - 1, k, 0) + tr2;
            CH(ic - 1, 1, k) = CC(i - 1, k, 0) - tr2;
            CH(i, 0, k) = ti2 + CC(i, k, 0);
            CH(ic, 1, k) = ti2 - CC(i, k, 0);
        }
#undef CC
#undef CH
}
#define REARRANGE(rx, ix, ry, iy) { const float t1 = rx + ry, t2 = ry - rx, t3 = ix + iy, t4 = ix - iy; rx = t1; ix = t3; ry = t4; iy = t2; }
static void radf3(uint64_t ido, uint64_t l1, const float* cc, float* ch, const float* wa) {
    const float taur = -0.5f, taui = (float)0.8660254037844386467637231707529362L;
#define CC(a, b, c) cc[(a) + ido * ((b) + l1 * (c))]
#define CH(a, b, c) ch[(a) + ido * ((b) + 3 * (c))]
    for (uint64_t k = 0; k < l1; k++) {
        const float cr2 = CC(0, k, 1) + CC(0, k, 2);
        CH(0, 0, k) = CC(0, k, 0) + cr2;
        CH(0, 2, k) = taui * (CC(0, k, 2) - CC(0, k, 1));
        CH(ido - 1, 1, k) = CC(0, k, 0) + taur * cr2;
    }
    if (ido == 1) return;
    for (uint64_t k = 0; k < l1; k++)
        for (uint64_t i = 2; i < ido; i += 2) {
            const uint64_t ic = ido - i;
            float di2, di3, dr2, dr3;
            MULPM(dr2, di2, RWA(0, i - 2), RWA(0, i - 1), CC(i - 1, k, 1), CC(i, k, 1))
            MULPM(dr3, di3, RWA(1, i - 2), RWA(1, i - 1), CC(i - 1, k, 2), CC(i, k, 2))
            REARRANGE(dr2, di2, dr3, di3)
            CH(i - 1, 0, k) = CC(i - 1, k, 0) + dr2;
            CH(i, 0, k) = CC(i, k, 0) + di2;
            const float tr2 = CC(i - 1, k, 0) + taur * dr2, ti2 = CC(i, k, 0) + taur * di2;
            const float tr3 = taui * dr3, ti3 = taui * di3;
            CH(i - 1, 2, k) = tr2 + tr3;
            CH(ic - 1, 1, k) = tr2 - tr3;
            CH(i, 2, k) = ti3 + ti2;
            CH(ic, 1, k) = ti3 - ti2;
        }
#undef CC
#undef CH
}
static void radf4(uint64_t ido, uint64_t l1, const float* cc, float* ch, const float* wa) {
    const float hsqt2 = (float)0.707106781186547524400844362104849L;
#define CC(a, b, c) cc[(a) + ido * ((b) + l1 * (c))]
#define CH(a, b, c) ch[(a) + ido * ((b) + 4 * (c))]
    for (uint64_t k = 0; k < l1; k++) {
        const float tr1 = CC(0, k, 3) + CC(0, k, 1);
        CH(0, 2, k) = CC(0, k, 3) - CC(0, k, 1);
        const float tr2 = CC(0, k, 0) + CC(0, k, 2);
        CH(ido - 1, 1, k) = CC(0, k, 0) - CC(0, k, 2);
        CH(0, 0, k) = tr2 + tr1;
        CH(ido - 1, 3, k) = tr2 - tr1;
    }
    if ((ido & 1) == 0)
        for (uint64_t k = 0; k < l1; k++) {
            const float ti1 = -hsqt2 * (CC(ido - 1, k, 1) + CC(ido - 1, k, 3));
            const float tr1 = hsqt2 * (CC(ido - 1, k, 1) - CC(ido - 1, k, 3));
            CH(ido - 1, 0, k) = CC(ido - 1, k, 0) + tr1;
            CH(ido - 1, 2, k) = CC(ido - 1, k, 0) - tr1;
            CH(0, 3, k) = ti1 + CC(ido - 1, k, 2);
            CH(0, 1, k) = ti1 - CC(ido - 1, k, 2);
        }
    if (ido <= 2) return;
    for (uint64_t k = 0; k < l1; k++)
        for (uint64_t i = 2; i < ido; i += 2) {
            const uint64_t ic = ido - i;
            float ci2, ci3, ci4, cr2, cr3, cr4;
            MULPM(cr2, ci2, RWA(0, i - 2), RWA(0, i - 1), CC(i - 1, k, 1), CC(i, k, 1))
            MULPM(cr3, ci3, RWA(1, i - 2), RWA(1, i - 1), CC(i - 1, k, 2), CC(i, k, 2))
            MULPM(cr4, ci4, RWA(2, i - 2), RWA(2, i - 1), CC(i - 1, k, 3), CC(i, k, 3))
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
        }
#undef CC
#undef CH
}
static void radf5(uint64_t ido, uint64_t l1, const float* cc, float* ch, const float* wa) {
    const float tr11 = (float)0.3090169943749474241022934171828191L, ti11 = (float)0.9510565162951535721164393333793821L,
                tr12 = (float)-0.8090169943749474241022934171828191L, ti12 = (float)0.5877852522924731291687059546390728L;
#define CC(a, b, c) cc[(a) + ido * ((b) + l1 * (c))]
#define CH(a, b, c) ch[(a) + ido * ((b) + 5 * (c))]
    for (uint64_t k = 0; k < l1; k++) {
        const float cr2 = CC(0, k, 4) + CC(0, k, 1), ci5 = CC(0, k, 4) - CC(0, k, 1);
        const float cr3 = CC(0, k, 3) + CC(0, k, 2), ci4 = CC(0, k, 3) - CC(0, k, 2);
        CH(0, 0, k) = CC(0, k, 0) + cr2 + cr3;
        CH(ido - 1, 1, k) = CC(0, k, 0) + tr11 * cr2 + tr12 * cr3;
        CH(0, 2, k) = ti11 * ci5 + ti12 * ci4;
        CH(ido - 1, 3, k) = CC(0, k, 0) + tr12 * cr2 + tr11 * cr3;
        CH(0, 4, k) = ti12 * ci5 - ti11 * ci4;
    }
    if (ido == 1) return;
    for (uint64_t k = 0; k < l1; ++k)
        for (uint64_t i = 2, ic = ido - 2; i < ido; i += 2, ic -= 2) {
            float di2, di3, di4, di5, dr2, dr3, dr4, dr5;
            MULPM(dr2, di2, RWA(0, i - 2), RWA(0, i - 1), CC(i - 1, k, 1), CC(i, k, 1))
            MULPM(dr3, di3, RWA(1, i - 2), RWA(1, i - 1), CC(i - 1, k, 2), CC(i, k, 2))
            MULPM(dr4, di4, RWA(2, i - 2), RWA(2, i - 1), CC(i - 1, k, 3), CC(i, k, 3))
            MULPM(dr5, di5, RWA(3, i - 2), RWA(3, i - 1), CC(i - 1, k, 4), CC(i, k, 4))
            REARRANGE(dr2, di2, dr5, di5)
            REARRANGE(dr3, di3, dr4, di4)
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
static void radb2(uint64_t ido, uint64_t l1, const float* cc, float* ch, const float* wa) {
#define CC(a, b, c) cc[(a) + ido * ((b) + 2 * (c))]
#define CH(a, b, c) ch[(a) + ido * ((b) + l1 * (c))]
    for (uint64_t k = 0; k < l1; k++) {
        CH(0, k, 0) = CC(0, 0, k) + CC(ido - 1, 1, k);
        CH(0, k, 1) = CC(0, 0, k) - CC(ido - 1, 1, k);
    }
    if ((ido & 1) == 0)
        for (uint64_t k = 0; k < l1; k++) {
            CH(ido - 1, k, 0) = 2 * CC(ido - 1, 0, k);
            CH(ido - 1, k, 1) = -2 * CC(0, 1, k);
        }
    if (ido <= 2) return;
    for (uint64_t k = 0; k < l1; ++k)
        for (uint64_t i = 2; i < ido; i += 2) {
            const uint64_t ic = ido - i;
            CH(i - 1, k, 0) = CC(i - 1, 0, k) + CC(ic - 1, 1, k);
            const float tr2 = CC(i - 1, 0, k) - CC(ic - 1, 1, k);
            const float ti2 = CC(i, 0, k) + CC(ic, 1, k);
            CH(i, k, 0) = CC(i, 0, k) - CC(ic, 1, k);
            MULPM(CH(i, k, 1), CH(i - 1, k, 1), RWA(0, i - 2), RWA(0, i - 1), ti2, tr2)
        }
#undef CC
#undef CH
}
static void radb3(uint64_t ido, uint64_t l1, const float* cc, float* ch, const float* wa) {
    const float taur = -0.5f, taui = (float)0.8660254037844386467637231707529362L;
#define CC(a, b, c) cc[(a) + ido * ((b) + 3 * (c))]
#define CH(a, b, c) ch[(a) + ido * ((b) + l1 * (c))]
    for (uint64_t k = 0; k < l1; k++) {
        const float tr2 = 2 * CC(ido - 1, 1, k);
        const float cr2 = CC(0, 0, k) + taur * tr2;
        CH(0, k, 0) = CC(0, 0, k) + tr2;
        const float ci3 = 2 * taui * CC(0, 2, k);
        CH(0, k, 2) = cr2 + ci3;
        CH(0, k, 1) = cr2 - ci3;
    }
    if (ido == 1) return;
    for (uint64_t k = 0; k < l1; k++)
        for (uint64_t i = 2, ic = ido - 2; i < ido; i += 2, ic -= 2) {
            const float tr2 = CC(i - 1, 2, k) + CC(ic - 1, 1, k), ti2 = CC(i, 2, k) - CC(ic, 1, k);
            const float cr2 = CC(i - 1, 0, k) + taur * tr2, ci2 = CC(i, 0, k) + taur * ti2;
            CH(i - 1, k, 0) = CC(i - 1, 0, k) + tr2;
            CH(i, k, 0) = CC(i, 0, k) + ti2;
            const float cr3 = taui * (CC(i - 1, 2, k) - CC(ic - 1, 1, k)), ci3 = taui * (CC(i, 2, k) + CC(ic, 1, k));
            const float dr3 = cr2 + ci3, dr2 = cr2 - ci3, di2 = ci2 + cr3, di3 = ci2 - cr3;
            MULPM(CH(i, k, 1), CH(i - 1, k, 1), RWA(0, i - 2), RWA(0, i - 1), di2, dr2)
            MULPM(CH(i, k, 2), CH(i - 1, k, 2), RWA(1, i - 2), RWA(1, i - 1), di3, dr3)
        }
#undef CC
#undef CH
}
static void radb4(uint64_t ido, uint64_t l1, const float* cc, float* ch, const float* wa) {
    const float sqrt2 = (float)1.414213562373095048801688724209698L;
#define CC(a, b, c) cc[(a) + ido * ((b) + 4 * (c))]
#define CH(a, b, c) ch[(a) + ido * ((b) + l1 * (c))]
    for (uint64_t k = 0; k < l1; k++) {
        const float tr2 = CC(0, 0, k) + CC(ido - 1, 3, k), tr1 = CC(0, 0, k) - CC(ido - 1, 3, k);
        const float tr3 = 2 * CC(ido - 1, 1, k), tr4 = 2 * CC(0, 2, k);
        CH(0, k, 0) = tr2 + tr3;
        CH(0, k, 2) = tr2 - tr3;
        CH(0, k, 3) = tr1 + tr4;
        CH(0, k, 1) = tr1 - tr4;
    }
    if ((ido & 1) == 0)
        for (uint64_t k = 0; k < l1; k++) {
            const float ti1 = CC(0, 3, k) + CC(0, 1, k), ti2 = CC(0, 3, k) - CC(0, 1, k);
            const float tr2 = CC(ido - 1, 0, k) + CC(ido - 1, 2, k), tr1 = CC(ido - 1, 0, k) - CC(ido - 1, 2, k);
            CH(ido - 1, k, 0) = tr2 + tr2;
            CH(ido - 1, k, 1) = sqrt2 * (tr1 - ti1);
            CH(ido - 1, k, 2) = ti2 + ti2;
            CH(ido - 1, k, 3) = -sqrt2 * (tr1 + ti1);
        }
    if (ido <= 2) return;
    for (uint64_t k = 0; k < l1; ++k)
        for (uint64_t i = 2; i < ido; i += 2) {
            const uint64_t ic = ido - i;
            const float tr2 = CC(i - 1, 0, k) + CC(ic - 1, 3, k), tr1 = CC(i - 1, 0, k) - CC(ic - 1, 3, k);
            const float ti1 = CC(i, 0, k) + CC(ic, 3, k), ti2 = CC(i, 0, k) - CC(ic, 3, k);
            const float tr4 = CC(i, 2, k) + CC(ic, 1, k), ti3 = CC(i, 2, k) - CC(ic, 1, k);
            const float tr3 = CC(i - 1, 2, k) + CC(ic - 1, 1, k), ti4 = CC(i - 1, 2, k) - CC(ic - 1, 1, k);
            CH(i - 1, k, 0) = tr2 + tr3;
            const float cr3 = tr2 - tr3;
            CH(i, k, 0) = ti2 + ti3;
            const float ci3 = ti2 - ti3;
            const float cr4 = tr1 + tr4, cr2 = tr1 - tr4, ci2 = ti1 + ti4, ci4 = ti1 - ti4;
            MULPM(CH(i, k, 1), CH(i - 1, k, 1), RWA(0, i - 2), RWA(0, i - 1), ci2, cr2)
            MULPM(CH(i, k, 2), CH(i - 1, k, 2), RWA(1, i - 2), RWA(1, i - 1), ci3, cr3)
            MULPM(CH(i, k, 3), CH(i - 1, k, 3), RWA(2, i - 2), RWA(2, i - 1), ci4, cr4)
        }
#undef CC
#undef CH
}
static void radb5(uint64_t ido, uint64_t l1, const float* cc, float* ch, const float* wa) {
    const float tr11 = (float)0.3090169943749474241022934171828191L, ti11 = (float)0.9510565162951535721164393333793821L,
                tr12 = (float)-0.8090169943749474241022934171828191L, ti12 = (float)0.5877852522924731291687059546390728L;
#define CC(a, b, c) cc[(a) + ido * ((b) + 5 * (c))]
#define CH(a, b, c) ch[(a) + ido * ((b) + l1 * (c))]
    for (uint64_t k = 0; k < l1; k++) {
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
    }
    if (ido == 1) return;
    for (uint64_t k = 0; k < l1; ++k)
        for (uint64_t i = 2, ic = ido - 2; i < ido; i += 2, ic -= 2) {
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
            MULPM(CH(i, k, 1), CH(i - 1, k, 1), RWA(0, i - 2), RWA(0, i - 1), di2, dr2)
            MULPM(CH(i, k, 2), CH(i - 1, k, 2), RWA(1, i - 2), RWA(1, i - 1), di3, dr3)
            MULPM(CH(i, k, 3), CH(i - 1, k, 3), RWA(2, i - 2), RWA(2, i - 1), di4, dr4)
            MULPM(CH(i, k, 4), CH(i - 1, k, 4), RWA(3, i - 2), RWA(3, i - 1), di5, dr5)
        }
#undef CC
#undef CH
}
#undef REARRANGE
#undef MULPM
#undef RWA

/* radfg (pocketfft.hh:1753-1893): generic radix, forward.  Works on BOTH arrays like the
 * reference (cc is modified in place, the result ends in cc: the caller swaps once more). */
static void radfg(uint64_t ido, uint64_t ip, uint64_t l1, float* cc, float* ch, const float* wa,
                  const float* csarr) {
    const uint64_t cdim = ip, ipph = (ip + 1) / 2, idl1 = ido * l1;
#define CC(a, b, c) cc[(a) + ido * ((b) + cdim * (c))]
#define CH(a, b, c) ch[(a) + ido * ((b) + l1 * (c))]
#define C1(a, b, c) cc[(a) + ido * ((b) + l1 * (c))]
#define C2(a, b) cc[(a) + idl1 * (b)]
#define CH2(a, b) ch[(a) + idl1 * (b)]
    if (ido > 1) {
        for (uint64_t j = 1, jc = ip - 1; j < ipph; ++j, --jc) {
            const uint64_t is = (j - 1) * (ido - 1), is2 = (jc - 1) * (ido - 1);
            for (uint64_t k = 0; k < l1; ++k) {
                uint64_t idij = is, idij2 = is2;
                for (uint64_t i = 1; i <= ido - 2; i += 2) {
                    const float t1 = C1(i, k, j), t2 = C1(i + 1, k, j), t3 = C1(i, k, jc), t4 = C1(i + 1, k, jc);
                    const float x1 = wa[idij] * t1 + wa[idij + 1] * t2, x2 = wa[idij] * t2 - wa[idij + 1] * t1,
                                x3 = wa[idij2] * t3 + wa[idij2 + 1] * t4, x4 = wa[idij2] * t4 - wa[idij2 + 1] * t3;
                    C1(i, k, j) = x3 + x1;
                    C1(i + 1, k, jc) = x3 - x1;
                    C1(i + 1, k, j) = x2 + x4;
                    C1(i, k, jc) = x2 - x4;
                    idij += 2;
                    idij2 += 2;
                }
            }
        }
    }
    for (uint64_t j = 1, jc = ip - 1; j < ipph; ++j, --jc)
        for (uint64_t k = 0; k < l1; ++k) { /* MPINPLACE(C1(0,k,jc), C1(0,k,j)) */
            const float t = C1(0, k, jc);
            C1(0, k, jc) = C1(0, k, jc) - C1(0, k, j);
            C1(0, k, j) = t + C1(0, k, j);
        }
    for (uint64_t l = 1, lc = ip - 1; l < ipph; ++l, --lc) {
        for (uint64_t ik = 0; ik < idl1; ++ik) {
            CH2(ik, l) = C2(ik, 0) + csarr[2 * l] * C2(ik, 1) + csarr[4 * l] * C2(ik, 2);
            CH2(ik, lc) = csarr[2 * l + 1] * C2(ik, ip - 1) + csarr[4 * l + 1] * C2(ik, ip - 2);
        }
        uint64_t iang = 2 * l, j = 3, jc = ip - 3;
        for (; j < ipph - 3; j += 4, jc -= 4) {
            iang += l; if (iang >= ip) iang -= ip;
            const float ar1 = csarr[2 * iang], ai1 = csarr[2 * iang + 1];
            iang += l; if (iang >= ip) iang -= ip;
            const float ar2 = csarr[2 * iang], ai2 = csarr[2 * iang + 1];
            iang += l; if (iang >= ip) iang -= ip;
            const float ar3 = csarr[2 * iang], ai3 = csarr[2 * iang + 1];
            iang += l; if (iang >= ip) iang -= ip;
            const float ar4 = csarr[2 * iang], ai4 = csarr[2 * iang + 1];
            for (uint64_t ik = 0; ik < idl1; ++ik) {
                CH2(ik, l) += ar1 * C2(ik, j) + ar2 * C2(ik, j + 1) + ar3 * C2(ik, j + 2) + ar4 * C2(ik, j + 3);
                CH2(ik, lc) += ai1 * C2(ik, jc) + ai2 * C2(ik, jc - 1) + ai3 * C2(ik, jc - 2) + ai4 * C2(ik, jc - 3);
            }
        }
        for (; j < ipph - 1; j += 2, jc -= 2) {
            iang += l; if (iang >= ip) iang -= ip;
            const float ar1 = csarr[2 * iang], ai1 = csarr[2 * iang + 1];
            iang += l; if (iang >= ip) iang -= ip;
            const float ar2 = csarr[2 * iang], ai2 = csarr[2 * iang + 1];
            for (uint64_t ik = 0; ik < idl1; ++ik) {
                CH2(ik, l) += ar1 * C2(ik, j) + ar2 * C2(ik, j + 1);
                CH2(ik, lc) += ai1 * C2(ik, jc) + ai2 * C2(ik, jc - 1);
            }
        }
        for (; j < ipph; ++j, --jc) {
            iang += l; if (iang >= ip) iang -= ip;
            const float ar = csarr[2 * iang], ai = csarr[2 * iang + 1];
            for (uint64_t ik = 0; ik < idl1; ++ik) {
                CH2(ik, l) += ar * C2(ik, j);
                CH2(ik, lc) += ai * C2(ik, jc);
            }
        }
    }
    for (uint64_t ik = 0; ik < idl1; ++ik) CH2(ik, 0) = C2(ik, 0);
    for (uint64_t j = 1; j < ipph; ++j)
        for (uint64_t ik = 0; ik < idl1; ++ik) CH2(ik, 0) += C2(ik, j);
    for (uint64_t k = 0; k < l1; ++k)
        for (uint64_t i = 0; i < ido; ++i) CC(i, 0, k) = CH(i, k, 0);
    for (uint64_t j = 1, jc = ip - 1; j < ipph; ++j, --jc) {
        const uint64_t j2 = 2 * j - 1;
        for (uint64_t k = 0; k < l1; ++k) {
            CC(ido - 1, j2, k) = CH(0, k, j);
            CC(0, j2 + 1, k) = CH(0, k, jc);
        }
    }
    if (ido == 1) return;
    for (uint64_t j = 1, jc = ip - 1; j < ipph; ++j, --jc) {
        const uint64_t j2 = 2 * j - 1;
        for (uint64_t k = 0; k < l1; ++k)
            for (uint64_t i = 1, ic = ido - i - 2; i <= ido - 2; i += 2, ic -= 2) {
                CC(i, j2 + 1, k) = CH(i, k, j) + CH(i, k, jc);
                CC(ic, j2, k) = CH(i, k, j) - CH(i, k, jc);
                CC(i + 1, j2 + 1, k) = CH(i + 1, k, j) + CH(i + 1, k, jc);
                CC(ic + 1, j2, k) = CH(i + 1, k, jc) - CH(i + 1, k, j);
            }
    }
#undef CC
#undef CH
#undef C1
#undef C2
#undef CH2
}
/* radbg (pocketfft.hh:2076-2208): generic radix, backward; result in ch, cc used as scratch. */
static void radbg(uint64_t ido, uint64_t ip, uint64_t l1, float* cc, float* ch, const float* wa,
                  const float* csarr) {
    const uint64_t cdim = ip, ipph = (ip + 1) / 2, idl1 = ido * l1;
#define CC(a, b, c) cc[(a) + ido * ((b) + cdim * (c))]
#define CH(a, b, c) ch[(a) + ido * ((b) + l1 * (c))]
#define C1(a, b, c) cc[(a) + ido * ((b) + l1 * (c))]
#define C2(a, b) cc[(a) + idl1 * (b)]
#define CH2(a, b) ch[(a) + idl1 * (b)]
    for (uint64_t k = 0; k < l1; ++k)
        for (uint64_t i = 0; i < ido; ++i) CH(i, k, 0) = CC(i, 0, k);
    for (uint64_t j = 1, jc = ip - 1; j < ipph; ++j, --jc) {
        const uint64_t j2 = 2 * j - 1;
        for (uint64_t k = 0; k < l1; ++k) {
            CH(0, k, j) = 2 * CC(ido - 1, j2, k);
            CH(0, k, jc) = 2 * CC(0, j2 + 1, k);
        }
    }
    if (ido != 1) {
        for (uint64_t j = 1, jc = ip - 1; j < ipph; ++j, --jc) {
            const uint64_t j2 = 2 * j - 1;
            for (uint64_t k = 0; k < l1; ++k)
                for (uint64_t i = 1, ic = ido - i - 2; i <= ido - 2; i += 2, ic -= 2) {
                    CH(i, k, j) = CC(i, j2 + 1, k) + CC(ic, j2, k);
                    CH(i, k, jc) = CC(i, j2 + 1, k) - CC(ic, j2, k);
                    CH(i + 1, k, j) = CC(i + 1, j2 + 1, k) - CC(ic + 1, j2, k);
                    CH(i + 1, k, jc) = CC(i + 1, j2 + 1, k) + CC(ic + 1, j2, k);
                }
        }
    }
    for (uint64_t l = 1, lc = ip - 1; l < ipph; ++l, --lc) {
        for (uint64_t ik = 0; ik < idl1; ++ik) {
            C2(ik, l) = CH2(ik, 0) + csarr[2 * l] * CH2(ik, 1) + csarr[4 * l] * CH2(ik, 2);
            C2(ik, lc) = csarr[2 * l + 1] * CH2(ik, ip - 1) + csarr[4 * l + 1] * CH2(ik, ip - 2);
        }
        uint64_t iang = 2 * l, j = 3, jc = ip - 3;
        for (; j < ipph - 3; j += 4, jc -= 4) {
            iang += l; if (iang > ip) iang -= ip;
            const float ar1 = csarr[2 * iang], ai1 = csarr[2 * iang + 1];
            iang += l; if (iang > ip) iang -= ip;
            const float ar2 = csarr[2 * iang], ai2 = csarr[2 * iang + 1];
            iang += l; if (iang > ip) iang -= ip;
            const float ar3 = csarr[2 * iang], ai3 = csarr[2 * iang + 1];
            iang += l; if (iang > ip) iang -= ip;
            const float ar4 = csarr[2 * iang], ai4 = csarr[2 * iang + 1];
            for (uint64_t ik = 0; ik < idl1; ++ik) {
                C2(ik, l) += ar1 * CH2(ik, j) + ar2 * CH2(ik, j + 1) + ar3 * CH2(ik, j + 2) + ar4 * CH2(ik, j + 3);
                C2(ik, lc) += ai1 * CH2(ik, jc) + ai2 * CH2(ik, jc - 1) + ai3 * CH2(ik, jc - 2) + ai4 * CH2(ik, jc - 3);
            }
        }
        for (; j < ipph - 1; j += 2, jc -= 2) {
            iang += l; if (iang > ip) iang -= ip;
            const float ar1 = csarr[2 * iang], ai1 = csarr[2 * iang + 1];
            iang += l; if (iang > ip) iang -= ip;
            const float ar2 = csarr[2 * iang], ai2 = csarr[2 * iang + 1];
            for (uint64_t ik = 0; ik < idl1; ++ik) {
                C2(ik, l) += ar1 * CH2(ik, j) + ar2 * CH2(ik, j + 1);
                C2(ik, lc) += ai1 * CH2(ik, jc) + ai2 * CH2(ik, jc - 1);
            }
        }
        for (; j < ipph; ++j, --jc) {
            iang += l; if (iang > ip) iang -= ip;
            const float war = csarr[2 * iang], wai = csarr[2 * iang + 1];
            for (uint64_t ik = 0; ik < idl1; ++ik) {
                C2(ik, l) += war * CH2(ik, j);
                C2(ik, lc) += wai * CH2(ik, jc);
            }
        }
    }
    for (uint64_t j = 1; j < ipph; ++j)
        for (uint64_t ik = 0; ik < idl1; ++ik) CH2(ik, 0) += CH2(ik, j);
    for (uint64_t j = 1, jc = ip - 1; j < ipph; ++j, --jc)
        for (uint64_t k = 0; k < l1; ++k) {
            CH(0, k, jc) = C1(0, k, j) + C1(0, k, jc);
            CH(0, k, j) = C1(0, k, j) - C1(0, k, jc);
        }
    if (ido == 1) return;
    for (uint64_t j = 1, jc = ip - 1; j < ipph; ++j, --jc)
        for (uint64_t k = 0; k < l1; ++k)
            for (uint64_t i = 1; i <= ido - 2; i += 2) {
                CH(i, k, j) = C1(i, k, j) - C1(i + 1, k, jc);
                CH(i, k, jc) = C1(i, k, j) + C1(i + 1, k, jc);
                CH(i + 1, k, j) = C1(i + 1, k, j) + C1(i, k, jc);
                CH(i + 1, k, jc) = C1(i + 1, k, j) - C1(i, k, jc);
            }
    for (uint64_t j = 1; j < ip; ++j) {
        const uint64_t is = (j - 1) * (ido - 1);
        for (uint64_t k = 0; k < l1; ++k) {
            uint64_t idij = is;
            for (uint64_t i = 1; i <= ido - 2; i += 2) {
                const float t1 = CH(i, k, j), t2 = CH(i + 1, k, j);
                CH(i, k, j) = wa[idij] * t1 - wa[idij + 1] * t2;
                CH(i + 1, k, j) = wa[idij] * t2 + wa[idij + 1] * t1;
                idij += 2;
            }
        }
    }
#undef CC
#undef CH
#undef C1
#undef C2
#undef CH2
}

/* rfftp::factorize (:2277-2297): 4s, a lone 2 moved to the front, odd divisors ascending. */
int jst_oracle_rfft_factors(uint64_t n, uint32_t* fact) {
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
        while ((len % d) == 0) { fact[nf++] = (uint32_t)d; len /= d; }
    if (len > 1) fact[nf++] = (uint32_t)len;
    return nf;
}
/* pocketfft_r's plan choice (:2508-2527): 0 = rfftp, else the Bluestein length. */
uint64_t jst_oracle_rfft_bluestein_size(uint64_t n) {
    if (n == 0) return 0;
    const uint64_t lpf = (n < 50) ? 0 : largest_prime_factor(n);
    if (lpf * lpf <= n) return 0;
    const double comp1 = 0.5 * cost_guess(n);
    double comp2 = 2 * cost_guess(good_size_cmplx(2 * n - 1));
    comp2 *= 1.5;
    return (comp2 < comp1) ? good_size_cmplx(2 * n - 1) : 0;
}

typedef struct {
    uint64_t n;
    int nf;
    uint32_t fact[64];
    float* mem;
    const float* tw[64];
    const float* tws[64];
} rfftp_t;
static int rfftp_init(rfftp_t* p, uint64_t n) {
    memset(p, 0, sizeof(*p));
    p->n = n;
    if (n == 0) return -1;
    if (n == 1) return 0;
    p->nf = jst_oracle_rfft_factors(n, p->fact);
    uint64_t twsz = 0, l1 = 1;
    for (int k = 0; k < p->nf; ++k) {
        const uint64_t ip = p->fact[k], ido = n / (l1 * ip);
        twsz += (ip - 1) * (ido - 1);
        if (ip > 5) twsz += 2 * ip;
        l1 *= ip;
    }
    p->mem = (float*)calloc(twsz + 1, sizeof(float));
    sincos_t s;
    sincos_init(&s, n);
    float* ptr = p->mem;
    l1 = 1;
    for (int k = 0; k < p->nf; ++k) { /* comp_twiddle :2311-2345 */
        const uint64_t ip = p->fact[k], ido = n / (l1 * ip);
        if (k < p->nf - 1) {
            p->tw[k] = ptr;
            for (uint64_t j = 1; j < ip; ++j)
                for (uint64_t i = 1; i <= (ido - 1) / 2; ++i) {
                    const c32 w = sincos_get(&s, j * l1 * i);
                    ptr[(j - 1) * (ido - 1) + 2 * i - 2] = w.r;
                    ptr[(j - 1) * (ido - 1) + 2 * i - 1] = w.i;
                }
            ptr += (ip - 1) * (ido - 1);
        }
        if (ip > 5) { /* special factors required by the *g functions */
            float* t = ptr;
            p->tws[k] = ptr;
            ptr += 2 * ip;
            t[0] = 1.0f;
            t[1] = 0.0f;
            for (uint64_t i = 2, ic = 2 * ip - 2; i <= ic; i += 2, ic -= 2) {
                const c32 w = sincos_get(&s, i / 2 * (n / ip));
                t[i] = w.r;
                t[i + 1] = w.i;
                t[ic] = w.r;
                t[ic + 1] = -w.i;
            }
        }
        l1 *= ip;
    }
    sincos_free(&s);
    return 0;
}
static void rfftp_free(rfftp_t* p) { free(p->mem); }
/* rfftp::exec with fct == 1 (:2228-2272): c transformed in place, ch scratch. */
static void rfftp_exec(const rfftp_t* p, float* c, float* ch, int r2hc) {
    const uint64_t n = p->n;
    if (n == 1) return;
    float *p1 = c, *p2 = ch;
    if (r2hc) {
        uint64_t l1 = n;
        for (int k1 = 0; k1 < p->nf; ++k1) {
            const int k = p->nf - k1 - 1;
            const uint64_t ip = p->fact[k], ido = n / l1;
            l1 /= ip;
            if (ip == 4) radf4(ido, l1, p1, p2, p->tw[k]);
            else if (ip == 2) radf2(ido, l1, p1, p2, p->tw[k]);
            else if (ip == 3) radf3(ido, l1, p1, p2, p->tw[k]);
            else if (ip == 5) radf5(ido, l1, p1, p2, p->tw[k]);
            else { radfg(ido, ip, l1, p1, p2, p->tw[k], p->tws[k]); float* u = p1; p1 = p2; p2 = u; }
            float* t = p1; p1 = p2; p2 = t;
        }
    } else {
        uint64_t l1 = 1;
        for (int k = 0; k < p->nf; ++k) {
            const uint64_t ip = p->fact[k], ido = n / (ip * l1);
            if (ip == 4) radb4(ido, l1, p1, p2, p->tw[k]);
            else if (ip == 2) radb2(ido, l1, p1, p2, p->tw[k]);
            else if (ip == 3) radb3(ido, l1, p1, p2, p->tw[k]);
            else if (ip == 5) radb5(ido, l1, p1, p2, p->tw[k]);
            else radbg(ido, ip, l1, p1, p2, p->tw[k], p->tws[k]);
            float* t = p1; p1 = p2; p2 = t;
            l1 *= ip;
        }
    }
    if (p1 != c) memcpy(c, p1, n * sizeof(float));
}
/* One real line in place: halfcomplex forward (r2hc) or backward; Bluestein via fftblue::exec_r
 * (:2434-2457).  Returns 0, -2 when the plan needs radfg/radbg. */
static int rfft_line(float* c, uint64_t n, int r2hc, const rfftp_t* rp, const fftblue_t* bp,
                     float* scratch, c32* ctmp, c32* akf, c32* cscr) {
    if (bp) {
        if (r2hc) {
            const float zero = 0.0f * c[0];
            for (uint64_t m = 0; m < n; ++m) { ctmp[m].r = c[m]; ctmp[m].i = zero; }
            fftblue_exec(bp, ctmp, akf, cscr, 1);
            c[0] = ctmp[0].r;
            memcpy(c + 1, &ctmp[1].r, (n - 1) * sizeof(float));
        } else {
            ctmp[0].r = c[0];
            ctmp[0].i = c[0] * 0.0f;
            memcpy(&ctmp[1].r, c + 1, (n - 1) * sizeof(float));
            if ((n & 1) == 0) ctmp[n / 2].i = 0.0f * c[0];
            for (uint64_t m = 1; 2 * m < n; ++m) { ctmp[n - m].r = ctmp[m].r; ctmp[n - m].i = -ctmp[m].i; }
            fftblue_exec(bp, ctmp, akf, cscr, 0);
            for (uint64_t m = 0; m < n; ++m) c[m] = ctmp[m].r;
        }
        return 0;
    }
    rfftp_exec(rp, c, scratch, r2hc);
    return 0;
}
/* r2r_fftpack(real2hermitian = forward, forward) over dense rows (module_impl_native_cpu.cc:155-165). */
int jst_oracle_fft_r2r(const float* in, float* out, uint64_t n, uint64_t batch, int forward) {
    if (n == 0) return -1;
    const uint64_t n2 = jst_oracle_rfft_bluestein_size(n);
    rfftp_t rp;
    fftblue_t bp;
    int rc = 0;
    if (n2) { if (fftblue_init(&bp, n, n2) != 0) return -1; }
    else if ((rc = rfftp_init(&rp, n)) != 0) return rc;
    float* scratch = (float*)malloc((n + 1) * sizeof(float));
    c32* ctmp = (c32*)malloc((n + 1) * sizeof(c32));
    c32* akf = (c32*)malloc(((n2 ? n2 : 1)) * sizeof(c32));
    c32* cscr = (c32*)malloc(((n2 ? n2 : 1)) * sizeof(c32));
    for (uint64_t r = 0; r < batch; ++r) {
        float* row = out + r * n;
        memcpy(row, in + r * n, n * sizeof(float));
        rfft_line(row, n, forward, n2 ? NULL : &rp, n2 ? &bp : NULL, scratch, ctmp, akf, cscr);
    }
    free(scratch); free(ctmp); free(akf); free(cscr);
    if (n2) fftblue_free(&bp); else rfftp_free(&rp);
    return 0;
}
/* r2c forward (general_r2c, pocketfft.hh:3102-3150): n reals -> n/2+1 complex per row. */
int jst_oracle_fft_r2c(const float* in, float* out, uint64_t n, uint64_t batch) {
    if (n == 0) return -1;
    const uint64_t n2 = jst_oracle_rfft_bluestein_size(n), no = n / 2 + 1;
    rfftp_t rp;
    fftblue_t bp;
    int rc = 0;
    if (n2) { if (fftblue_init(&bp, n, n2) != 0) return -1; }
    else if ((rc = rfftp_init(&rp, n)) != 0) return rc;
    float* t = (float*)malloc((n + 1) * sizeof(float));
    float* scratch = (float*)malloc((n + 1) * sizeof(float));
    c32* ctmp = (c32*)malloc((n + 1) * sizeof(c32));
    c32* akf = (c32*)malloc(((n2 ? n2 : 1)) * sizeof(c32));
    c32* cscr = (c32*)malloc(((n2 ? n2 : 1)) * sizeof(c32));
    for (uint64_t r = 0; r < batch; ++r) {
        memcpy(t, in + r * n, n * sizeof(float));
        rfft_line(t, n, 1, n2 ? NULL : &rp, n2 ? &bp : NULL, scratch, ctmp, akf, cscr);
        float* o = out + 2 * r * no;
        o[0] = t[0];
        o[1] = 0.0f;
        uint64_t i = 1, ii = 1;
        for (; i < n - 1; i += 2, ++ii) { o[2 * ii] = t[i]; o[2 * ii + 1] = t[i + 1]; }
        if (i < n) { o[2 * ii] = t[i]; o[2 * ii + 1] = 0.0f; }
    }
    free(t); free(scratch); free(ctmp); free(akf); free(cscr);
    if (n2) fftblue_free(&bp); else rfftp_free(&rp);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Amplitude.  src/domains/dsp/amplitude/module_impl.cc:44-60 (coefficient) and
 * module_impl_native_cpu.cc:73-99 (kernels); Backend::ApproxLog10 is
 * include/jetstream/backend/devices/cpu/helpers.hh:59-74 (frexpf + cubic, separate mul/add).
 * ---------------------------------------------------------------------------------------- */
float jst_oracle_approx_log10(float x) {
    float y, f;
    int e;
    f = frexpf(fabsf(x), &e);
    y = 1.23149591368684f;
    y *= f;
    y += -4.11852516267426f;
    y *= f;
    y += 6.02197014179219f;
    y *= f;
    y += -3.13396450166353f;
    y += e;
    return y * 0.3010299956639812f;
}

/* bulk form for the sweep against the reference's own inline function (oracle/ref_helpers.cc, tests/test_oracle_ref_helpers.py) */
void jst_oracle_approx_log10_bits(uint32_t first, uint64_t count, float* out) {
    for (uint64_t i = 0; i < count; ++i) {
        const uint32_t b = first + (uint32_t)i;
        float x;
        memcpy(&x, &b, 4);
        out[i] = jst_oracle_approx_log10(x);
    }
}

float jst_oracle_amplitude_coeff(uint64_t normalization_size) {
    return 20.0f * log10f(1.0f / (float)normalization_size);
}

void jst_oracle_amplitude_cf32(uint32_t rank, const uint64_t* shape, const float* in,
                               const uint64_t* sin_, float* out, const uint64_t* sout,
                               float coeff) {
    odometer_t o;
    const uint64_t size = odo_init(&o, rank, shape);
    for (uint64_t i = 0; i < size; ++i) {
        const uint64_t ii = odo_offset(&o, sin_), io = odo_offset(&o, sout);
        const float re = in[2 * ii], im = in[2 * ii + 1];
        const float mag = sqrtf((re * re) + (im * im));
        out[io] = (mag == 0.0f) ? -INFINITY : 20.0f * jst_oracle_approx_log10(mag) + coeff;
        odo_step(&o);
    }
}

void jst_oracle_amplitude_f32(uint32_t rank, const uint64_t* shape, const float* in,
                              const uint64_t* sin_, float* out, const uint64_t* sout,
                              float coeff) {
    odometer_t o;
    const uint64_t size = odo_init(&o, rank, shape);
    for (uint64_t i = 0; i < size; ++i) {
        const float mag = fabsf(in[odo_offset(&o, sin_)]);
        out[odo_offset(&o, sout)] =
            (mag == 0.0f) ? -INFINITY : 20.0f * jst_oracle_approx_log10(mag) + coeff;
        odo_step(&o);
    }
}

/* ------------------------------------------------------------------------------------------
 * Range.  src/domains/core/range/module_impl.cc:51-62 (coefficients) and
 * module_impl_native_cpu.cc:67-82 (kernel, libm tanhf).
 * ---------------------------------------------------------------------------------------- */
void jst_oracle_range_coeffs(float min, float max, float* scale, float* offset) {
    const float lower = min < max ? min : max;
    const float upper = min < max ? max : min;
    if (lower == upper) {
        *scale = 0.0f;
        *offset = 0.5f;
        return;
    }
    *scale = 1.0f / (upper - lower);
    *offset = -lower * *scale;
}

void jst_oracle_range_f32(uint32_t rank, const uint64_t* shape, const float* in,
                          const uint64_t* sin_, float* out, const uint64_t* sout, float scale,
                          float offset) {
    odometer_t o;
    const uint64_t size = odo_init(&o, rank, shape);
    for (uint64_t i = 0; i < size; ++i) {
        const uint64_t io = odo_offset(&o, sout);
        if (scale == 0.0f) {
            out[io] = 0.5f;
        } else {
            const float normalized = in[odo_offset(&o, sin_)] * scale + offset;
            out[io] = 0.5f + 0.5f * tanhf(4.0f * (normalized - 0.5f));
        }
        odo_step(&o);
    }
}

/* ------------------------------------------------------------------------------------------
 * Spectrogram.  src/domains/visualization/spectrogram/module_impl.cc:104 (decay) and
 * module_impl_native_cpu.cc:61-87 (compute).  State 'bins' is F32 laid out [height][width]
 * (index x + idx*width) although the tensor shape is {width, height} (module_impl.cc:108).
 *
 * The reference computes  index = static_cast<U64>(in * (F32)height)  and accepts
 * 0 < index < height.  That cast is undefined for negative/NaN/huge values; on x86-64 (the
 * reference's CPU target) values in (-1,1) truncate to 0, values <= -1 wrap to >= 2^63, NaN
 * gives 0x8000000000000000 and +inf gives 0 -- all rejected.  The defined-behaviour statement
 * of the same rule, used here and by the HIP kernel, is:
 *     f = in * (float)height;  hit  <=>  (f >= 1.0f && f < (float)height);  index = (u64)f.
 * ---------------------------------------------------------------------------------------- */
float jst_oracle_spectrogram_decay(uint64_t batches) { return powf(0.999f, (float)batches); }

void jst_oracle_spectrogram(float* bins, const float* in, uint64_t batches, uint64_t width,
                            uint64_t height, uint64_t batch_stride, uint64_t elem_stride,
                            float decay) {
    const uint64_t total = width * height;
    for (uint64_t i = 0; i < total; ++i) bins[i] *= decay;
    const float fh = (float)height;
    for (uint64_t b = 0; b < batches; ++b) {
        for (uint64_t x = 0; x < width; ++x) {
            const float f = in[b * batch_stride + x * elem_stride] * fh;
            if (f >= 1.0f && f < fh) {
                const uint64_t index = (uint64_t)f;
                float* val = &bins[x + index * width];
                const float v = *val + 0.02f;
                *val = (1.0f < v) ? 1.0f : v; /* std::min(val + 0.02f, 1.0f) = (b < a) ? b : a */
            }
        }
    }
}

/* Raw x86-64 cast variant, for tests that pin the rule above against the literal cast on the
 * values where the cast is defined (f in [0, 2^63)). */
uint64_t jst_oracle_cast_u64(float f) { return (uint64_t)f; }

/* ------------------------------------------------------------------------------------------
 * Waterfall.  src/domains/visualization/waterfall/ring_state.hh:16-56 (plan + ring state),
 * module_impl_native_cpu.cc:53-78 (row copies).  bins: F32 [height][width] ring.
 * state[0] = writeIndex, state[1] = dirtyRows.
 * ---------------------------------------------------------------------------------------- */
void jst_oracle_waterfall_plan(uint64_t write_index, uint64_t batches, uint64_t height,
                               uint64_t* plan /* sourceRow, destinationRow, rowCount */) {
    const uint64_t retained = batches < height ? batches : height;
    const uint64_t source = batches - retained;
    plan[0] = source;
    plan[1] = (write_index + (source % height)) % height;
    plan[2] = retained;
}

void jst_oracle_waterfall_advance(uint64_t* state, uint64_t batches, uint64_t height) {
    state[0] = (state[0] + (batches % height)) % height;
    const uint64_t room = height - state[1];
    state[1] += batches < room ? batches : room;
}

void jst_oracle_waterfall_dirty_plan(const uint64_t* state, uint64_t height,
                                     uint64_t* plan /* startRow, first, second */) {
    const uint64_t start = (state[0] + height - state[1]) % height;
    const uint64_t first = state[1] < height - start ? state[1] : height - start;
    plan[0] = start;
    plan[1] = first;
    plan[2] = state[1] - first;
}

void jst_oracle_waterfall(float* bins, uint64_t* state, const float* in, uint64_t batches,
                          uint64_t width, uint64_t height, uint64_t batch_stride,
                          uint64_t elem_stride) {
    uint64_t plan[3];
    jst_oracle_waterfall_plan(state[0], batches, height, plan);
    for (uint64_t row = 0; row < plan[2]; ++row) {
        const uint64_t src = plan[0] + row;
        const uint64_t dst = (plan[1] + row) % height;
        for (uint64_t col = 0; col < width; ++col)
            bins[dst * width + col] = in[src * batch_stride + col * elem_stride];
    }
    jst_oracle_waterfall_advance(state, batches, height);
}

/* ------------------------------------------------------------------------------------------
 * Signal generator, cosine CF32 (the CW-tone input generator for the BASELINE configs).
 * src/domains/dsp/signal_generator/module_impl_native_cpu.cc:20-23 (WrapPhase),
 * :159-163 (advancePhase), :222-231 (kernelCosineCF32).  *phase carries across calls.
 * ---------------------------------------------------------------------------------------- */
static double wrap_phase(double value, double period) {
    const double w = fmod(value, period);
    return w < 0.0 ? w + period : w;
}

void jst_oracle_signal_cosine_cf32(float* out, uint64_t count, double amplitude,
                                   double frequency, double sample_rate, double dc_offset,
                                   double* phase) {
    double ph = wrap_phase(*phase, 2.0 * JST_PI);
    for (uint64_t i = 0; i < count; ++i) {
        out[2 * i] = (float)(amplitude * cos(ph) + dc_offset);
        out[2 * i + 1] = (float)(amplitude * sin(ph));
        ph = wrap_phase(ph + 2.0 * JST_PI * frequency / sample_rate, 2.0 * JST_PI);
    }
    *phase = ph;
}

/* All deterministic waveforms (module_impl_native_cpu.cc:159-375).  shape: 0 sine, 1 cosine,
 * 2 square, 3 triangle, 4 sawtooth, 5 dc, 6 chirp.  state[0] = oscillatorPhase, state[1] = chirpTime. */
void jst_oracle_signal(float* out, uint64_t count, int complex_out, int shape, double amplitude,
                       double frequency, double sample_rate, double dc_offset, double chirp_start,
                       double chirp_end, double chirp_duration, double* state) {
    const double period = 2.0 * JST_PI;
    double ph = wrap_phase(state[0], period), tm = state[1];
    for (uint64_t i = 0; i < count; ++i) {
        double re, im = 0.0;
        switch (shape) {
            case 0: re = amplitude * sin(ph) + dc_offset; im = -amplitude * cos(ph); break;
            case 1: case 6: re = amplitude * cos(ph) + dc_offset; im = amplitude * sin(ph); break;
            case 2: re = amplitude * (ph < JST_PI ? 1.0 : -1.0) + dc_offset; break;
            case 3: { const double pv = ph / (2.0 * JST_PI);
                      re = amplitude * (pv < 0.5 ? 4.0 * pv - 1.0 : 3.0 - 4.0 * pv) + dc_offset; break; }
            case 4: { const double pv = ph / (2.0 * JST_PI);
                      re = amplitude * (2.0 * pv - 1.0) + dc_offset; break; }
            default: re = amplitude + dc_offset; break;
        }
        if (complex_out) {
            out[2 * i] = (float)re;
            out[2 * i + 1] = (float)im;
        } else {
            out[i] = (float)re;
        }
        if (shape == 6) { /* advanceChirpPhase :165-190 */
            const double dt = 1.0 / sample_rate;
            const double rate = (chirp_end - chirp_start) / chirp_duration;
            double cycles = 0.0;
            const double until = chirp_duration - tm;
            if (dt < until) {
                cycles = (chirp_start + rate * tm) * dt + 0.5 * rate * dt * dt;
                tm += dt;
            } else {
                cycles = (chirp_start + rate * tm) * until + 0.5 * rate * until * until;
                const double after = dt - until;
                tm = after;
                if (after > 0.0) cycles += (chirp_start + rate * 0.0) * after + 0.5 * rate * after * after;
            }
            ph = wrap_phase(ph + 2.0 * JST_PI * cycles, period);
        } else if (shape != 5) {
            ph = wrap_phase(ph + 2.0 * JST_PI * frequency / sample_rate, period);
        }
    }
    state[0] = ph;
    state[1] = tm;
}

/* ------------------------------------------------------------------------------------------
 * Fold (spectral decimation).  src/domains/dsp/fold/module_impl_native_cpu.cc:103-172.
 * Dense tensors; 'axis' is the fold axis; out[k] = (1/D) * sum_g in[(k + g*size - offset) mod axis]
 * accumulated in F64.  channel_offsets (optional, per coordinate of channel_axis) replace offset.
 * ---------------------------------------------------------------------------------------- */
void jst_oracle_fold_cf32(const float* in, float* out, uint32_t rank, const uint64_t* in_shape,
                          uint64_t fold_axis, uint64_t fold_offset, uint64_t fold_size,
                          int64_t channel_axis, const uint64_t* channel_offsets) {
    uint64_t out_shape[ORACLE_MAX_RANK], in_str[ORACLE_MAX_RANK], out_str[ORACLE_MAX_RANK];
    const uint64_t axis_size = in_shape[fold_axis], decim = axis_size / fold_size;
    uint64_t total_out = 1;
    for (uint32_t d = 0; d < rank; ++d) out_shape[d] = (d == fold_axis) ? fold_size : in_shape[d];
    for (uint32_t d = rank; d-- > 0;) {
        in_str[d] = (d == rank - 1) ? 1 : in_str[d + 1] * in_shape[d + 1];
        out_str[d] = (d == rank - 1) ? 1 : out_str[d + 1] * out_shape[d + 1];
        total_out *= out_shape[d];
    }
    const double divisor = (double)decim;
    const uint64_t scalar_offset = fold_offset % axis_size;
    uint64_t coords[ORACLE_MAX_RANK];
    for (uint64_t oi = 0; oi < total_out; ++oi) {
        uint64_t rem = oi, base = 0;
        for (uint32_t d = 0; d < rank; ++d) {
            coords[d] = rem / out_str[d];
            rem %= out_str[d];
        }
        const uint64_t off = channel_axis < 0 ? scalar_offset
                                              : channel_offsets[coords[channel_axis]] % axis_size;
        for (uint32_t d = 0; d < rank; ++d)
            if (d != fold_axis) base += coords[d] * in_str[d];
        double sr = 0.0, si = 0.0;
        for (uint64_t g = 0; g < decim; ++g) {
            const uint64_t shifted = coords[fold_axis] + g * fold_size;
            const uint64_t ia = shifted >= off ? shifted - off : axis_size - (off - shifted);
            const uint64_t idx = base + ia * in_str[fold_axis];
            sr += (double)in[2 * idx];
            si += (double)in[2 * idx + 1];
        }
        sr /= divisor;
        si /= divisor;
        out[2 * oi] = (float)sr;
        out[2 * oi + 1] = (float)si;
    }
}

/* ------------------------------------------------------------------------------------------
 * Arithmetic, reduction by "add" along one axis: out zeroed, then out += in in row-major input
 * order (src/domains/core/arithmetic/module_impl_native_cpu.cc:98-146): a left-to-right F32 sum
 * starting from +0.  in: dense [outer, r, inner]; out: [outer, inner].  'complex' doubles lanes.
 * ---------------------------------------------------------------------------------------- */
void jst_oracle_arithmetic_add_f32(const float* in, float* out, uint64_t outer, uint64_t r,
                                   uint64_t inner) {
    for (uint64_t o = 0; o < outer; ++o)
        for (uint64_t i = 0; i < inner; ++i) {
            float acc = 0.0f;
            for (uint64_t k = 0; k < r; ++k) acc += in[(o * r + k) * inner + i];
            out[o * inner + i] = acc;
        }
}

/* ------------------------------------------------------------------------------------------
 * Filter taps.  src/domains/dsp/filter_taps/module_impl_native_cpu.cc:46-80.
 * out: CF32 [heads][taps].  std::exp(j*2*pi*n*offset) has real part exactly 0*..., so it is
 * (cos(theta), sin(theta)) with theta = ((2.0*pi)*n)*offset (left-to-right products).
 * ---------------------------------------------------------------------------------------- */
void jst_oracle_filter_taps(float* out, double sample_rate, double bandwidth,
                            const double* center, uint64_t heads, uint64_t taps) {
    const double filter_width = (bandwidth / sample_rate) / 2.0;
    for (uint64_t c = 0; c < heads; ++c) {
        const double filter_offset = center[c] / sample_rate;
        for (uint64_t i = 0; i < taps; ++i) {
            const double fi = (double)i, half = (double)(taps - 1) / 2.0, n = fi - half;
            const double sinc = (n == 0.0) ? (2.0 * filter_width)
                                           : sin(2.0 * JST_PI * filter_width * n) / (JST_PI * n);
            const double win = (taps == 1) ? 1.0
                                           : 0.42 - 0.50 * cos(2.0 * JST_PI * fi / (taps - 1)) +
                                                 0.08 * cos(4.0 * JST_PI * fi / (taps - 1));
            const double theta = ((2.0 * JST_PI) * n) * filter_offset;
            const double sw = sinc * win;
            out[2 * (c * taps + i)] = (float)(sw * cos(theta));
            out[2 * (c * taps + i) + 1] = (float)(sw * sin(theta));
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Phase correction.  src/domains/dsp/phase_correction/module_impl_native_cpu.cc:60-115.
 * Dense CF32; batch/channel coordinates from the flat index; phases[C] (F64) carried across calls.
 * ---------------------------------------------------------------------------------------- */
void jst_oracle_phase_correction(const float* in, float* out, uint64_t count, uint64_t batch_count,
                                 uint64_t batch_inner, uint64_t channel_count,
                                 uint64_t channel_inner, const double* increments, double* phases) {
    float* corr = (float*)malloc(2 * channel_count * batch_count * sizeof(float));
    double* wrapped = (double*)malloc(channel_count * sizeof(double));
    for (uint64_t c = 0; c < channel_count; ++c) wrapped[c] = remainder(increments[c], 2.0 * JST_PI);
    for (uint64_t c = 0; c < channel_count; ++c)
        for (uint64_t b = 0; b < batch_count; ++b) {
            const double ph = phases[c] + wrapped[c] * (double)b;
            corr[2 * (c * batch_count + b)] = (float)cos(ph);
            corr[2 * (c * batch_count + b) + 1] = (float)sin(ph);
        }
    for (uint64_t i = 0; i < count; ++i) {
        const uint64_t b = batch_count == 1 ? 0 : (i / batch_inner) % batch_count;
        const uint64_t c = channel_count == 1 ? 0 : (i / channel_inner) % channel_count;
        const float* k = &corr[2 * (c * batch_count + b)];
        cmul_f32(in[2 * i], in[2 * i + 1], k[0], k[1], &out[2 * i], &out[2 * i + 1]);
    }
    for (uint64_t c = 0; c < channel_count; ++c)
        phases[c] = remainder(phases[c] + wrapped[c] * (double)batch_count, 2.0 * JST_PI);
    free(corr);
    free(wrapped);
}

/* ------------------------------------------------------------------------------------------
 * FM demodulator.  src/domains/dsp/fm/module_impl.cc:108-172 (coefficients, biquads) and
 * module_impl_native_cpu.cc:43-174 (compute).  One lane: 'batches' x 'samples' consecutive
 * samples (batch-major), state carried in fm_state_t across calls.  mode: 0 narrow, 1 wide.
 * Output: narrow F32[batches*samples]; wide F32[batches*samples*2] interleaved (left, right).
 * ---------------------------------------------------------------------------------------- */
typedef struct { float b0, b1, b2, a1, a2; } fm_biquad_t;
typedef struct { float z1, z2; } fm_bqstate_t;
typedef struct {
    float prev_re, prev_im;
    int has_prev;
    float narrow_deemph;
    float pilot_phase, pilot_cos_stage, pilot_sin_stage, pilot_cos, pilot_sin, left_de, right_de;
    fm_bqstate_t sum_notch, diff_notch, sum_filter[3], diff_filter[3];
} fm_state_t;
typedef struct {
    float ref, pilot_inc, pilot_alpha, deemph_alpha;
    fm_biquad_t notch, lp[3];
    int wide, deemph_enabled;
} fm_coeffs_t;

void jst_oracle_fm_coeffs(fm_coeffs_t* k, int wide, int deemph /*0 none,1 50us,2 75us*/, float sample_rate) {
    const float deviation = wide ? 75e3f : 100e3f;
    const float kf = deviation / sample_rate;
    k->wide = wide;
    k->deemph_enabled = deemph != 0;
    k->ref = 1.0f / (2.0f * JST_PI * kf);                 /* F32 * F64 -> F64, then to F32 */
    k->pilot_inc = 2.0f * JST_PI * 19e3f / sample_rate;
    const double sr = sample_rate;
    k->pilot_alpha = (float)(1.0 - exp(-2.0 * JST_PI * 200.0 / sr));
    if (deemph == 0) k->deemph_alpha = 1.0f;
    else k->deemph_alpha = (float)(1.0 - exp(-1.0 / (sr * (deemph == 1 ? 50e-6 : 75e-6))));
    const double pw = 2.0 * JST_PI * 19e3 / sr, pc = cos(pw), ps = sin(pw);
    const double na = ps / (2.0 * 20.0), na0 = 1.0 + na;
    k->notch.b0 = (float)(1.0 / na0);
    k->notch.b1 = (float)(-2.0 * pc / na0);
    k->notch.b2 = k->notch.b0;
    k->notch.a1 = k->notch.b1;
    k->notch.a2 = (float)((1.0 - na) / na0);
    const double q[3] = {0.51763809, 0.70710678, 1.93185165};
    const double w = 2.0 * JST_PI * 15e3 / sr, co = cos(w), si = sin(w);
    for (int s = 0; s < 3; ++s) {
        const double al = si / (2.0 * q[s]), a0 = 1.0 + al;
        k->lp[s].b0 = (float)((1.0 - co) * 0.5 / a0);
        k->lp[s].b1 = (float)((1.0 - co) / a0);
        k->lp[s].b2 = k->lp[s].b0;
        k->lp[s].a1 = (float)(-2.0 * co / a0);
        k->lp[s].a2 = (float)((1.0 - al) / a0);
    }
}

static float fm_biquad(float x, const fm_biquad_t* c, fm_bqstate_t* s) {
    const float y = c->b0 * x + s->z1;
    s->z1 = c->b1 * x - c->a1 * y + s->z2;
    s->z2 = c->b2 * x - c->a2 * y;
    return y;
}
static float fm_lowpass(float x, const fm_coeffs_t* k, fm_bqstate_t* s) {
    for (int i = 0; i < 3; ++i) x = fm_biquad(x, &k->lp[i], &s[i]);
    return x;
}

void jst_oracle_fm_lane(const float* in, float* out, uint64_t count, const fm_coeffs_t* k,
                        fm_state_t* st) {
    const double two_pi = 2.0f * JST_PI; /* the reference compares/subtracts in F64 (F32 op F64) */
    float pr = st->prev_re, pi_ = st->prev_im;
    int has = st->has_prev;
    for (uint64_t n = 0; n < count; ++n) {
        const float cr = in[2 * n], ci = in[2 * n + 1];
        const int fin = isfinite(cr) && isfinite(ci) && isfinite(pr) && isfinite(pi_);
        float d;
        if (!has) d = 0.0f;
        else if (fin) {
            float re, im; /* conj(previous) * current */
            cmul_f32(pr, -pi_, cr, ci, &re, &im);
            d = atan2f(im, re) * k->ref;
        } else d = NAN;
        if (!isfinite(d)) {
            if (!k->wide) out[n] = d;
            else {
                out[2 * n] = d;
                out[2 * n + 1] = d;
                st->pilot_phase += k->pilot_inc;
                if ((double)st->pilot_phase >= two_pi) st->pilot_phase = (float)((double)st->pilot_phase - two_pi);
            }
        } else if (!k->wide) {
            if (!k->deemph_enabled) out[n] = d;
            else {
                st->narrow_deemph += k->deemph_alpha * (d - st->narrow_deemph);
                out[n] = st->narrow_deemph;
            }
        } else {
            const float pc = cosf(st->pilot_phase), ps = sinf(st->pilot_phase);
            st->pilot_cos_stage += k->pilot_alpha * (d * pc - st->pilot_cos_stage);
            st->pilot_sin_stage += k->pilot_alpha * (d * ps - st->pilot_sin_stage);
            st->pilot_cos += k->pilot_alpha * (st->pilot_cos_stage - st->pilot_cos);
            st->pilot_sin += k->pilot_alpha * (st->pilot_sin_stage - st->pilot_sin);
            const float sum = fm_lowpass(fm_biquad(d, &k->notch, &st->sum_notch), k, st->sum_filter);
            const float po = atan2f(st->pilot_cos, st->pilot_sin);
            const float carrier = sinf(2.0f * (st->pilot_phase + po));
            const float diff = fm_lowpass(fm_biquad(2.0f * d * carrier, &k->notch, &st->diff_notch),
                                          k, st->diff_filter);
            float left = sum + diff, right = sum - diff;
            if (k->deemph_enabled) {
                st->left_de += k->deemph_alpha * (left - st->left_de);
                st->right_de += k->deemph_alpha * (right - st->right_de);
                left = st->left_de;
                right = st->right_de;
            }
            out[2 * n] = left;
            out[2 * n + 1] = right;
            st->pilot_phase += k->pilot_inc;
            if ((double)st->pilot_phase >= two_pi) st->pilot_phase = (float)((double)st->pilot_phase - two_pi);
        }
        pr = cr;
        pi_ = ci;
        has = 1;
    }
    st->prev_re = pr;
    st->prev_im = pi_;
    st->has_prev = 1;
}
uint64_t jst_oracle_fm_state_size(void) { return sizeof(fm_state_t); }
uint64_t jst_oracle_fm_coeffs_size(void) { return sizeof(fm_coeffs_t); }

/* ------------------------------------------------------------------------------------------
 * Lineplot compute.  src/domains/visualization/lineplot/module_impl_native_cpu.cc:80-118,
 * normalisation module_impl.cc:91.  avg[] (F32[elements]) is the carried state.
 * ---------------------------------------------------------------------------------------- */
void jst_oracle_lineplot(float* avg, const float* in, uint64_t batches, uint64_t elements,
                         uint64_t batch_stride, uint64_t elem_stride, uint64_t decimation,
                         uint64_t averaging) {
    const float norm = 1.0f / (0.5f * (float)batches);
    const float av = (float)averaging;
    for (uint64_t i = 0; i < elements; ++i) {
        float sum = 0.0f;
        for (uint64_t b = 0; b < batches; ++b) sum += in[b * batch_stride + i * decimation * elem_stride];
        const float amplitude = fminf(fmaxf((sum * norm) - 1.0f, -1.0f), 1.0f);
        avg[i] -= avg[i] / av;
        avg[i] += amplitude / av;
    }
}

/* ------------------------------------------------------------------------------------------
 * AGC: tiled RMS gain with interpolation.  src/domains/dsp/agc/module_impl_native_cpu.cc:20-160
 * (SamplePower :24-33, LimitGainToFiniteRange :35-41, ClampToF32 :43-45, ApplyGain :47-64,
 * LimitGainChange :66-77, ApplyTiledRmsAgc :79-152).  Dense [lanes][samples] arrays.
 * ---------------------------------------------------------------------------------------- */
static double agc_clamp(double v, double lo, double hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }
static double agc_limit_gain(double magnitude, double gain, double limit) {
    return magnitude > limit / gain ? nextafter(limit / magnitude, 0.0) : gain;
}
static float agc_clamp_f32(double v) { return (float)agc_clamp(v, -(double)FLT_MAX, (double)FLT_MAX); }
static double agc_limit_change(double gain, double prev, double min_gain, double max_gain,
                               double max_change) {
    const double q = prev / max_change;
    const double lo = (min_gain < q) ? q : min_gain;
    const double hi = prev > max_gain / max_change ? max_gain : prev * max_change;
    return agc_clamp(gain, lo, hi);
}
static double agc_tile_gain(const float* in, int complex_in, uint64_t start, uint64_t len,
                            double reference, double epsilon, double min_gain, double max_gain) {
    double sum = 0.0;
    for (uint64_t s = 0; s < len; ++s) {
        if (complex_in) {
            const double re = in[2 * (start + s)], im = in[2 * (start + s) + 1];
            sum += re * re + im * im;
        } else {
            const double v = in[start + s];
            sum += v * v;
        }
    }
    const double mean = sum / (double)len;
    return agc_clamp(reference / sqrt(mean + epsilon), min_gain, max_gain);
}
void jst_oracle_agc(const float* in, float* out, int complex_in, uint64_t lanes, uint64_t samples,
                    uint64_t tile, double reference, double epsilon, double min_gain,
                    double max_gain, double max_change) {
    const double max_safe_c = (double)nextafterf(FLT_MAX, 0.0f);
    const uint64_t tiles = 1 + (samples - 1) / tile;
    const uint64_t w = complex_in ? 2 : 1;
    for (uint64_t lane = 0; lane < lanes; ++lane) {
        const float* li = in + lane * samples * w;
        float* lo = out + lane * samples * w;
        double start_gain = agc_tile_gain(li, complex_in, 0, samples < tile ? samples : tile,
                                          reference, epsilon, min_gain, max_gain);
        for (uint64_t t = 0; t < tiles; ++t) {
            const uint64_t ts = t * tile;
            const uint64_t len = tile < samples - ts ? tile : samples - ts;
            double end_gain = start_gain;
            if (t + 1 < tiles) {
                const uint64_t ns = (t + 1) * tile;
                const uint64_t nl = tile < samples - ns ? tile : samples - ns;
                end_gain = agc_limit_change(agc_tile_gain(li, complex_in, ns, nl, reference, epsilon,
                                                          min_gain, max_gain),
                                            start_gain, min_gain, max_gain, max_change);
            }
            const double step = (end_gain - start_gain) / (double)len;
            for (uint64_t s = 0; s < len; ++s) {
                const double gain = start_gain + step * (double)s;
                if (complex_in) {
                    const double re = li[2 * (ts + s)], im = li[2 * (ts + s) + 1];
                    const double g = agc_limit_gain(hypot(re, im), gain, max_safe_c);
                    lo[2 * (ts + s)] = agc_clamp_f32(re * g);
                    lo[2 * (ts + s) + 1] = agc_clamp_f32(im * g);
                } else {
                    const double v = li[ts + s];
                    const double g = agc_limit_gain(fabs(v), gain, (double)FLT_MAX);
                    lo[ts + s] = agc_clamp_f32(v * g);
                }
            }
            start_gain = end_gain;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * CPU baseline driver for bench.py's `cpu_baseline` leg (test/bench infrastructure like the rest
 * of this file).  One pass = the hot path over `rows` batches of n samples with dense loops and
 * caller-provided buffers (no temporaries): Multiply by the static window (x) invert table,
 * FFT -- through `fft`, the REFERENCE's own pocketfft (oracle/_ref, SIMD across batches as in
 * fft/module_impl_native_cpu.cc:125-140) when given, else the restatement above --, Amplitude,
 * Range, Spectrogram.  Timed the way the reference times its own benchmarks
 * (src/benchmark.cc:100-106,175-186: nanobench -- warm-up, epochs of >= 100 ms, MEDIAN per op).
 * ---------------------------------------------------------------------------------------- */
#include <time.h>

typedef int (*jst_oracle_fft_fn)(uint32_t rank, const uint64_t* shape, const int64_t* stride_in,
                                 const int64_t* stride_out, uint64_t axis, int forward,
                                 const float* in, float* out);

void jst_oracle_chain_pass(jst_oracle_fft_fn fft, const float* x, const float* window, uint64_t rows,
                           uint64_t n, float coeff, float scale, float offset, uint64_t height,
                           float decay, float* product, float* spectrum, float* out, float* bins) {
    for (uint64_t r = 0; r < rows; ++r)
        for (uint64_t i = 0; i < n; ++i)
            cmul_f32(x[2 * (r * n + i)], x[2 * (r * n + i) + 1], window[2 * i], window[2 * i + 1],
                     &product[2 * (r * n + i)], &product[2 * (r * n + i) + 1]);
    if (fft) {
        const uint64_t shape[2] = {rows, n};
        const int64_t stride[2] = {(int64_t)(n * 8), 8};
        (void)fft(2, shape, stride, stride, 1, 1, product, spectrum);
    } else {
        (void)jst_oracle_fft_c2c(product, spectrum, n, rows, 1);
    }
    const uint64_t total = rows * n;
    for (uint64_t i = 0; i < total; ++i) {  /* amplitude/module_impl_native_cpu.cc:73-86 + range :67-82 */
        const float re = spectrum[2 * i], im = spectrum[2 * i + 1];
        const float mag = sqrtf((re * re) + (im * im));
        const float db = (mag == 0.0f) ? -INFINITY : 20.0f * jst_oracle_approx_log10(mag) + coeff;
        if (scale == 0.0f) {
            out[i] = 0.5f;
        } else {
            const float normalized = db * scale + offset;
            out[i] = 0.5f + 0.5f * tanhf(4.0f * (normalized - 0.5f));
        }
    }
    jst_oracle_spectrogram(bins, out, rows, n, height, n, 1, decay);
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static int cmp_double(const void* a, const void* b) {
    const double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}

/* Returns the median over `epochs` epochs (each >= min_epoch_s of passes) of samples per second;
 * rates[e] (if given) receives every epoch's figure. */
double jst_oracle_chain_bench(jst_oracle_fft_fn fft, const float* x, const float* window, uint64_t rows,
                              uint64_t n, float coeff, float scale, float offset, uint64_t height,
                              float* product, float* spectrum, float* out, float* bins,
                              double min_epoch_s, uint32_t epochs, double* rates) {
    const float decay = jst_oracle_spectrogram_decay(rows);
    double local[64];
    if (epochs == 0 || epochs > 64) return -1.0;
    jst_oracle_chain_pass(fft, x, window, rows, n, coeff, scale, offset, height, decay, product, spectrum,
                          out, bins); /* warm-up: pages touched, plans built */
    for (uint32_t e = 0; e < epochs; ++e) {
        uint64_t passes = 0;
        const double t0 = now_s();
        double t1;
        do {
            jst_oracle_chain_pass(fft, x, window, rows, n, coeff, scale, offset, height, decay, product,
                                  spectrum, out, bins);
            ++passes;
            t1 = now_s();
        } while (t1 - t0 < min_epoch_s);
        local[e] = (double)(passes * rows * n) / (t1 - t0);
        if (rates) rates[e] = local[e];
    }
    qsort(local, epochs, sizeof(double), cmp_double);
    return (epochs & 1) ? local[epochs / 2] : 0.5 * (local[epochs / 2 - 1] + local[epochs / 2]);
}

/* BASELINE configs[0]: the reference's `benchmark` shape -- ONE batch, FFT -> Amplitude, timed per compute() like
 * src/benchmark.cc:100-106,175-186 drives nanobench (warm-up, epochs of >= min_epoch_s, MEDIAN elapsed per op).
 * One op = fft/module_impl_native_cpu.cc:125-140 (pocketfft c2c, forward) + amplitude/module_impl_native_cpu.cc:73-86
 * on CF32[n].  Returns the median seconds per op over `epochs` epochs; per_op[e] (if given) receives each epoch's. */
double jst_oracle_fft_amplitude_bench(jst_oracle_fft_fn fft, const float* x, uint64_t n, float coeff, float* spectrum,
                                      float* out, double min_epoch_s, uint32_t epochs, double* per_op) {
    double local[64];
    if (epochs == 0 || epochs > 64) return -1.0;
    const uint64_t shape[1] = {n};
    const int64_t stride[1] = {8};
    for (uint32_t e = 0; e <= epochs; ++e) { /* epoch 0 is the warm-up (plan built, pages touched) */
        uint64_t ops = 0;
        const double t0 = now_s();
        double t1;
        do {
            if (fft) (void)fft(1, shape, stride, stride, 0, 1, x, spectrum);
            else (void)jst_oracle_fft_c2c(x, spectrum, n, 1, 1);
            for (uint64_t i = 0; i < n; ++i) {
                const float re = spectrum[2 * i], im = spectrum[2 * i + 1];
                const float mag = sqrtf((re * re) + (im * im));
                out[i] = (mag == 0.0f) ? -INFINITY : 20.0f * jst_oracle_approx_log10(mag) + coeff;
            }
            ++ops;
            t1 = now_s();
        } while (t1 - t0 < (e ? min_epoch_s : 0.02));
        if (e) {
            local[e - 1] = (t1 - t0) / (double)ops;
            if (per_op) per_op[e - 1] = local[e - 1];
        }
    }
    qsort(local, epochs, sizeof(double), cmp_double);
    return (epochs & 1) ? local[epochs / 2] : 0.5 * (local[epochs / 2 - 1] + local[epochs / 2]);
}
