// twx_fft.h — mixed-radix (2/3/4/5 and their products) in-register butterflies and the
// LDS-staged Stockham stages used by every FFT pass of the TWSTFT correlator.
//
// Design: a transform of length N = N1*N2 (5 000 000 = 625*8000 for the 1-s
// window of processing/Octave/godual_ranging.m:25-28) is done as a column pass (length N1,
// W adjacent columns per workgroup) and a row pass (length N2), each an in-LDS Stockham autosort
// FFT whose first stage loads straight from global memory and whose last stage stores straight
// to global memory.  All code here is plain C++ (TWX_HD = __host__ __device__) so the same
// templates are exercised on the CPU by tests/cpu/fft_emul.cpp (thread loop emulation).
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TWX_HD __host__ __device__ __forceinline__
#define TWX_UNROLL _Pragma("unroll")
#else
#define TWX_HD inline
#define TWX_UNROLL
#endif

namespace twx {

// ------------------------------------------------------------------------------------------
// complex helpers
// ------------------------------------------------------------------------------------------
#if defined(__clang__)
// native 2-vectors: complex add/sub/scale map 1:1 onto v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32
// (no register shuffles), and the swizzles of a complex multiply fold into op_sel / neg modifiers
template <typename T> struct cpx_sel;
template <> struct cpx_sel<float> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct cpx_sel<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <typename T> using cpx = typename cpx_sel<T>::type;
template <typename T> TWX_HD cpx<T> mk(T x, T y) { cpx<T> r; r.x = x; r.y = y; return r; }
#define TWX_CPX_OPS(T)                                                                             \
    TWX_HD cpx<T> cmul_g(cpx<T> a, cpx<T> b) { cpx<T> bs; bs.x = -b.y; bs.y = b.x; return a.xx * b + a.yy * bs; }   \
    TWX_HD cpx<T> cmulc_g(cpx<T> a, cpx<T> b) { cpx<T> bs; bs.x = b.y; bs.y = -b.y; return a * b.xx + a.yx * bs; }   \
    TWX_HD cpx<T> cconj(cpx<T> a) { cpx<T> r; r.x = a.x; r.y = -a.y; return r; }                  \
    TWX_HD cpx<T> cscale(cpx<T> a, T s) { return a * s; }                                          \
    TWX_HD T cnorm(cpx<T> a) { return a.x * a.x + a.y * a.y; }                                     \
    TWX_HD cpx<T> mul_mi(cpx<T> a) { return a.yx * cpx<T>{T(1), T(-1)}; }                          \
    TWX_HD cpx<T> mul_pi(cpx<T> a) { return a.yx * cpx<T>{T(-1), T(1)}; }                          \
    /* b + (-i)*d and b + (+i)*d as ONE packed fma on the swizzled operand (exact: the factors are +-1) */ \
    TWX_HD cpx<T> cadd_mi(cpx<T> b, cpx<T> d) { return __builtin_elementwise_fma(d.yx, cpx<T>{T(1), T(-1)}, b); }  \
    TWX_HD cpx<T> cadd_pi(cpx<T> b, cpx<T> d) { return __builtin_elementwise_fma(d.yx, cpx<T>{T(-1), T(1)}, b); }  \
    /* a*(c + i*s): packed mul + packed fma, the rotation folded into the constant pair */       \
    TWX_HD cpx<T> cmul_cs(cpx<T> a, T c, T s) { return __builtin_elementwise_fma(a.yx, cpx<T>{-s, s}, a * c); }
TWX_CPX_OPS(float)
TWX_CPX_OPS(double)
#undef TWX_CPX_OPS
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TWX_NO_ASM_CMUL)
// fp32 complex multiply as exactly two packed instructions: the operand swizzles and the sign live in
// op_sel / neg modifiers (hipcc builds the rotated operand with v_xor + v_mov otherwise: 4 instructions).
// Plain VGPR-to-VGPR VALU ops: no wait states needed around the asm (cdna_hip_programming.md §5.7).
__device__ __forceinline__ cpx<float> cmul_f32_asm(cpx<float> a, cpx<float> b) {
    cpx<float> t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));                  // (a.x b.x, a.x b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(t));   // (-a.y b.y + t.x, a.y b.x + t.y)
    return d;
}
__device__ __forceinline__ cpx<float> cmulc_f32_asm(cpx<float> a, cpx<float> b) {   // a * conj(b)
    cpx<float> t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));                  // (a.x b.x, a.y b.x)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(t));   // (a.y b.y + t.x, -a.x b.y + t.y)
    return d;
}
// (a*b)*c and (a*conj(b))*conj(c) as ONE asm block of four packed instructions: the compiler puts a wait state
// (s_nop) after every asm statement whose result feeds the next instruction, so a chain of two products costs one
// instead of four.
__device__ __forceinline__ cpx<float> cmul3_f32_asm(cpx<float> a, cpx<float> b, cpx<float> c) {
    cpx<float> t, d;
    asm("v_pk_mul_f32 %0, %2, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_mul_f32 %0, %1, %4 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %1, %1, %4, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=&v"(t), "=&v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ cpx<float> cmulc3_f32_asm(cpx<float> a, cpx<float> b, cpx<float> c) {
    cpx<float> t, d;
    asm("v_pk_mul_f32 %0, %2, %3 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]\n\t"
        "v_pk_mul_f32 %0, %1, %4 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %1, %1, %4, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "=&v"(t), "=&v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// a*s and (a*b)*s with s WAVE-UNIFORM: the same packed instructions with the uniform operand left in a scalar register pair
// (op_sel / neg work on SGPR-pair sources as on VGPR pairs; one SGPR source per instruction = the constant-bus limit)
__device__ __forceinline__ cpx<float> cmul_us_f32_asm(cpx<float> a, cpx<float> s) {
    cpx<float> t, d;
    asm("v_pk_mul_f32 %0, %2, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=&v"(t), "=&v"(d) : "v"(a), "s"(s));
    return d;
}
__device__ __forceinline__ cpx<float> cmul3_us_f32_asm(cpx<float> a, cpx<float> b, cpx<float> s) {
    cpx<float> t, d;
    asm("v_pk_mul_f32 %0, %2, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_mul_f32 %0, %1, %4 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %1, %1, %4, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=&v"(t), "=&v"(d) : "v"(a), "v"(b), "s"(s));
    return d;
}
#define TWX_ASM_CMUL 1
#endif
TWX_HD cpx<double> cmul(cpx<double> a, cpx<double> b) { return cmul_g(a, b); }
TWX_HD cpx<double> cmulc(cpx<double> a, cpx<double> b) { return cmulc_g(a, b); }
TWX_HD cpx<double> cmul3(cpx<double> a, cpx<double> b, cpx<double> c) { return cmul_g(cmul_g(a, b), c); }
TWX_HD cpx<double> cmulc3(cpx<double> a, cpx<double> b, cpx<double> c) { return cmulc_g(cmulc_g(a, b), c); }
#if defined(TWX_ASM_CMUL)
__device__ __forceinline__ cpx<float> cmul_us(cpx<float> a, cpx<float> s) { return cmul_us_f32_asm(a, s); }
__device__ __forceinline__ cpx<float> cmul3_us(cpx<float> a, cpx<float> b, cpx<float> s) { return cmul3_us_f32_asm(a, b, s); }
TWX_HD cpx<double> cmul_us(cpx<double> a, cpx<double> s) { return cmul_g(a, s); }
TWX_HD cpx<double> cmul3_us(cpx<double> a, cpx<double> b, cpx<double> s) { return cmul_g(cmul_g(a, b), s); }
__device__ __forceinline__ cpx<float> cmul(cpx<float> a, cpx<float> b) { return cmul_f32_asm(a, b); }
__device__ __forceinline__ cpx<float> cmulc(cpx<float> a, cpx<float> b) { return cmulc_f32_asm(a, b); }
__device__ __forceinline__ cpx<float> cmul3(cpx<float> a, cpx<float> b, cpx<float> c) { return cmul3_f32_asm(a, b, c); }      // (a*b)*c
__device__ __forceinline__ cpx<float> cmulc3(cpx<float> a, cpx<float> b, cpx<float> c) { return cmulc3_f32_asm(a, b, c); }    // a*conj(b)*conj(c)
#else
TWX_HD cpx<float> cmul3(cpx<float> a, cpx<float> b, cpx<float> c) { return cmul_g(cmul_g(a, b), c); }
TWX_HD cpx<float> cmulc3(cpx<float> a, cpx<float> b, cpx<float> c) { return cmulc_g(cmulc_g(a, b), c); }
TWX_HD cpx<float> cmul_us(cpx<float> a, cpx<float> s) { return cmul_g(a, s); }
TWX_HD cpx<float> cmul3_us(cpx<float> a, cpx<float> b, cpx<float> s) { return cmul_g(cmul_g(a, b), s); }
TWX_HD cpx<double> cmul_us(cpx<double> a, cpx<double> s) { return cmul_g(a, s); }
TWX_HD cpx<double> cmul3_us(cpx<double> a, cpx<double> b, cpx<double> s) { return cmul_g(cmul_g(a, b), s); }
TWX_HD cpx<float> cmul(cpx<float> a, cpx<float> b) { return cmul_g(a, b); }
TWX_HD cpx<float> cmulc(cpx<float> a, cpx<float> b) { return cmulc_g(a, b); }
#endif
#else
template <typename T> struct cpx { T x, y; };

template <typename T> TWX_HD cpx<T> mk(T x, T y) { cpx<T> r; r.x = x; r.y = y; return r; }
template <typename T> TWX_HD cpx<T> operator+(cpx<T> a, cpx<T> b) { return mk<T>(a.x + b.x, a.y + b.y); }
template <typename T> TWX_HD cpx<T> operator-(cpx<T> a, cpx<T> b) { return mk<T>(a.x - b.x, a.y - b.y); }
template <typename T> TWX_HD cpx<T> cmul(cpx<T> a, cpx<T> b) {
    return mk<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
template <typename T> TWX_HD cpx<T> cmulc(cpx<T> a, cpx<T> b) {  // a * conj(b)
    return mk<T>(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
template <typename T> TWX_HD cpx<T> cmul3(cpx<T> a, cpx<T> b, cpx<T> c) { return cmul(cmul(a, b), c); }
template <typename T> TWX_HD cpx<T> cmulc3(cpx<T> a, cpx<T> b, cpx<T> c) { return cmulc(cmulc(a, b), c); }
template <typename T> TWX_HD cpx<T> cmul_us(cpx<T> a, cpx<T> s) { return cmul(a, s); }
template <typename T> TWX_HD cpx<T> cmul3_us(cpx<T> a, cpx<T> b, cpx<T> s) { return cmul(cmul(a, b), s); }
template <typename T> TWX_HD cpx<T> cconj(cpx<T> a) { return mk<T>(a.x, -a.y); }
template <typename T> TWX_HD cpx<T> cscale(cpx<T> a, T s) { return mk<T>(a.x * s, a.y * s); }
template <typename T> TWX_HD T cnorm(cpx<T> a) { return a.x * a.x + a.y * a.y; }
// multiply by -i (forward quarter turn e^{-i pi/2}) or +i
template <typename T> TWX_HD cpx<T> mul_mi(cpx<T> a) { return mk<T>(a.y, -a.x); }
template <typename T> TWX_HD cpx<T> mul_pi(cpx<T> a) { return mk<T>(-a.y, a.x); }
template <typename T> TWX_HD cpx<T> cadd_mi(cpx<T> b, cpx<T> d) { return mk<T>(b.x + d.y, b.y - d.x); }
template <typename T> TWX_HD cpx<T> cadd_pi(cpx<T> b, cpx<T> d) { return mk<T>(b.x - d.y, b.y + d.x); }
template <typename T> TWX_HD cpx<T> cmul_cs(cpx<T> a, T c, T s) { return mk<T>(a.x * c - a.y * s, a.x * s + a.y * c); }
#endif

// ------------------------------------------------------------------------------------------
// compile-time trigonometry: cos/sin of 2*pi*num/den, exact octant reduction + Taylor series
// ------------------------------------------------------------------------------------------
constexpr double kPi = 3.14159265358979323846264338327950288;

constexpr double cx_sin_small(double x) {  // |x| <= pi/4
    double x2 = x * x, term = x, sum = x;
    for (int k = 1; k <= 12; ++k) { term *= -x2 / double((2 * k) * (2 * k + 1)); sum += term; }
    return sum;
}
constexpr double cx_cos_small(double x) {
    double x2 = x * x, term = 1.0, sum = 1.0;
    for (int k = 1; k <= 12; ++k) { term *= -x2 / double((2 * k - 1) * (2 * k)); sum += term; }
    return sum;
}
struct cx_cs { double c, s; };
constexpr cx_cs cx_cossin_turn(long long num, long long den) {  // angle = 2*pi*num/den
    long long a = num % den; if (a < 0) a += den;
    long long p = 2 * a, q = den;  // theta = pi * p / q in [0, 2pi)
    double sc = 1.0, ss = 1.0;
    if (p > q) { p = 2 * q - p; ss = -ss; }            // theta -> 2pi - theta
    if (2 * p > q) { p = q - p; sc = -sc; }            // theta -> pi - theta
    bool swap = false;
    if (4 * p > q) { p = q - 2 * p; q = 2 * q; swap = true; }  // theta -> pi/2 - theta
    double x = kPi * double(p) / double(q);
    double c = cx_cos_small(x), s = cx_sin_small(x);
    if (swap) { double t = c; c = s; s = t; }
    return cx_cs{sc * c, ss * s};
}

// multiply by the constant W_R^E = exp(-+ 2*pi*i*E/R)  (forward: minus; INV: plus)
template <typename T, int R, int E, bool INV> TWX_HD cpx<T> cmul_const(cpx<T> a) {
    constexpr int e = ((E % R) + R) % R;
    if constexpr (e == 0) {
        return a;
    } else if constexpr ((4 * e) % R == 0) {
        constexpr int quarter = (4 * e) / R;  // 1,2,3
        if constexpr (quarter == 2) return mk<T>(-a.x, -a.y);
        else if constexpr ((quarter == 1) != INV) return mul_mi(a);
        else return mul_pi(a);
    } else {
        constexpr cx_cs w = cx_cossin_turn(e, R);
        constexpr T c = T(w.c);
        constexpr T s = INV ? T(w.s) : T(-w.s);
        return cmul_cs(a, c, s);
    }
}

// ------------------------------------------------------------------------------------------
// in-register DFT of size R (natural order in, natural order out)
// ------------------------------------------------------------------------------------------
template <int R> struct FirstFactor {
    static constexpr int value = (R % 4 == 0 && R > 4) ? 4 : (R % 5 == 0 && R > 5) ? 5
                               : (R % 2 == 0 && R > 2) ? 2 : (R % 3 == 0 && R > 3) ? 3 : (R % 7 == 0 && R > 7) ? 7 : R;
};

constexpr int cx_gcd(int a, int b) { return b == 0 ? a : cx_gcd(b, a % b); }
constexpr int cx_inv_mod(int a, int m) {                // a^-1 mod m for coprime a, m (m small)
    for (int x = 1; x < m; ++x) if ((a * x) % m == 1) return x;
    return 1;                                           // m == 1
}
#ifndef TWX_PFA
#define TWX_PFA 1
#endif

template <typename T, int R, bool INV, int A = FirstFactor<R>::value> struct Bfly {
    // composite: R = A*B, input r = A*b + a, output q = qb + B*qa
    static constexpr int B = R / A;
    static TWX_HD void run(cpx<T>* v) {
        if constexpr (TWX_PFA && cx_gcd(A, B) == 1) {
            // Coprime factors (20 = 4*5, 10 = 5*2, 12, 14, 15, 18, 21 ...): Good-Thomas prime-factor mapping — with
            // n = (B a + A b) mod R and k = (B u ka + A w kb) mod R, u = B^-1 mod A, w = A^-1 mod B, the cross terms of
            // n*k are multiples of R, so DFT_R = (DFT_A x DFT_B) between two permutations and there are NO twiddle
            // factors between the two layers.  The permutations are compile-time register renamings: a radix-20
            // butterfly loses its 12 constant complex multiplications (a fifth of its instructions).
            constexpr int U = cx_inv_mod(B % A, A), W_ = cx_inv_mod(A % B, B);
            cpx<T> y[A][B];
            TWX_UNROLL
            for (int a = 0; a < A; ++a) {
                cpx<T> t[B];
                TWX_UNROLL
                for (int b = 0; b < B; ++b) t[b] = v[(B * a + A * b) % R];
                Bfly<T, B, INV>::run(t);
                TWX_UNROLL
                for (int kb = 0; kb < B; ++kb) y[a][kb] = t[kb];
            }
            cpx<T> o[R];
            TWX_UNROLL
            for (int kb = 0; kb < B; ++kb) {
                cpx<T> t[A];
                TWX_UNROLL
                for (int a = 0; a < A; ++a) t[a] = y[a][kb];
                Bfly<T, A, INV>::run(t);
                TWX_UNROLL
                for (int ka = 0; ka < A; ++ka) o[(B * U * ka + A * W_ * kb) % R] = t[ka];
            }
            TWX_UNROLL
            for (int i = 0; i < R; ++i) v[i] = o[i];
            return;
        }
        cpx<T> y[A][B];
        TWX_UNROLL
        for (int a = 0; a < A; ++a) {
            cpx<T> t[B];
            TWX_UNROLL
            for (int b = 0; b < B; ++b) t[b] = v[A * b + a];
            Bfly<T, B, INV>::run(t);
            TWX_UNROLL
            for (int qb = 0; qb < B; ++qb) y[a][qb] = t[qb];
        }
        twiddle_rows(y, Idx<0>{});
        TWX_UNROLL
        for (int qb = 0; qb < B; ++qb) {
            cpx<T> t[A];
            TWX_UNROLL
            for (int a = 0; a < A; ++a) t[a] = y[a][qb];
            Bfly<T, A, INV>::run(t);
            TWX_UNROLL
            for (int qa = 0; qa < A; ++qa) v[qb + B * qa] = t[qa];
        }
    }
    template <int I> struct Idx {};
    // y[a][qb] *= W_R^{a*qb}, with compile-time exponents (I enumerates a*B+qb)
    template <int I> static TWX_HD void twiddle_rows(cpx<T> (&y)[A][B], Idx<I>) {
        constexpr int a = I / B, qb = I % B;
        y[a][qb] = cmul_const<T, R, a * qb, INV>(y[a][qb]);
        twiddle_rows(y, Idx<I + 1>{});
    }
    static TWX_HD void twiddle_rows(cpx<T> (&)[A][B], Idx<R>) {}
};

template <typename T, bool INV> struct Bfly<T, 1, INV, 1> {
    static TWX_HD void run(cpx<T>*) {}
};
template <typename T, bool INV> struct Bfly<T, 2, INV, 2> {
    static TWX_HD void run(cpx<T>* v) {
        cpx<T> a = v[0], b = v[1];
        v[0] = a + b; v[1] = a - b;
    }
};
template <typename T, bool INV> struct Bfly<T, 3, INV, 3> {
    static TWX_HD void run(cpx<T>* v) {
        constexpr T s = T(0.86602540378443864676372317075294);  // sin(2pi/3)
        cpx<T> t1 = v[1] + v[2], t2 = v[1] - v[2];
        cpx<T> m = v[0] - cscale(t1, T(0.5));
        cpx<T> st2 = cscale(t2, s);
        v[0] = v[0] + t1;
        v[1] = INV ? cadd_pi(m, st2) : cadd_mi(m, st2);   // m -+ i*s*t2
        v[2] = INV ? cadd_mi(m, st2) : cadd_pi(m, st2);
    }
};
template <typename T, bool INV> struct Bfly<T, 4, INV, 4> {
    static TWX_HD void run(cpx<T>* v) {
        cpx<T> a = v[0] + v[2], b = v[0] - v[2], c = v[1] + v[3], d = v[1] - v[3];
        v[0] = a + c; v[2] = a - c;
        v[1] = INV ? cadd_pi(b, d) : cadd_mi(b, d);
        v[3] = INV ? cadd_mi(b, d) : cadd_pi(b, d);
    }
};
template <typename T, bool INV> struct Bfly<T, 5, INV, 5> {
    static TWX_HD void run(cpx<T>* v) {
        constexpr T c1 = T(0.30901699437494742410229341718282);   // cos(2pi/5)
        constexpr T c2 = T(-0.80901699437494742410229341718282);  // cos(4pi/5)
        constexpr T s1 = T(0.95105651629515357211643933337938);   // sin(2pi/5)
        constexpr T s2 = T(0.58778525229247312916870595463907);   // sin(4pi/5)
        cpx<T> t1 = v[1] + v[4], t2 = v[2] + v[3], t3 = v[1] - v[4], t4 = v[2] - v[3];
        cpx<T> a1 = v[0] + cscale(t1, c1) + cscale(t2, c2);
        cpx<T> a2 = v[0] + cscale(t1, c2) + cscale(t2, c1);
        cpx<T> b1 = cscale(t3, s1) + cscale(t4, s2);
        cpx<T> b2 = cscale(t3, s2) - cscale(t4, s1);
        // forward: y1 = a1 - i*b1, y4 = a1 + i*b1, y2 = a2 - i*b2, y3 = a2 + i*b2
        v[0] = v[0] + t1 + t2;
        v[1] = INV ? cadd_pi(a1, b1) : cadd_mi(a1, b1); v[4] = INV ? cadd_mi(a1, b1) : cadd_pi(a1, b1);
        v[2] = INV ? cadd_pi(a2, b2) : cadd_mi(a2, b2); v[3] = INV ? cadd_mi(a2, b2) : cadd_pi(a2, b2);
    }
};

// radix 7 (prime): pairs (n, 7-n) → three sums a_j and three differences b_j;  y[k] = x0 + sum_j a_j cos(2 pi j k/7)
// -+ i sum_j b_j sin(2 pi j k/7), and y[7-k] is its mirror.  Needed for window lengths with a factor 7 (a native
// 70 Msps x 1 s window is 2^7 5^7 7 samples); 14 = 2*7 and 21 = 3*7 build on it.
template <typename T, bool INV> struct Bfly<T, 7, INV, 7> {
    static TWX_HD void run(cpx<T>* v) {
        constexpr T c1 = T(0.62348980185873353052500488400424), c2 = T(-0.22252093395631440428890256449679),
                    c3 = T(-0.90096886790241912623610231950745);                       // cos(2 pi j/7)
        constexpr T s1 = T(0.78183148246802980870844452667406), s2 = T(0.97492791218182360701813168299393),
                    s3 = T(0.43388373911755812047576833284836);                        // sin(2 pi j/7)
        const cpx<T> a1 = v[1] + v[6], a2 = v[2] + v[5], a3 = v[3] + v[4];
        const cpx<T> b1 = v[1] - v[6], b2 = v[2] - v[5], b3 = v[3] - v[4];
        const cpx<T> x0 = v[0];
        // k = 1: cos(1,2,3 · 2pi/7) = c1 c2 c3, sin = s1 s2 s3;  k = 2: indices 2,4,6 → c2 c3 c1, s2 -s3 -s1;  k = 3: 3,6,9 → c3 c1 c2, s3 -s1 s2
        const cpx<T> p1 = x0 + cscale(a1, c1) + cscale(a2, c2) + cscale(a3, c3);
        const cpx<T> p2 = x0 + cscale(a1, c2) + cscale(a2, c3) + cscale(a3, c1);
        const cpx<T> p3 = x0 + cscale(a1, c3) + cscale(a2, c1) + cscale(a3, c2);
        const cpx<T> q1 = cscale(b1, s1) + cscale(b2, s2) + cscale(b3, s3);
        const cpx<T> q2 = cscale(b1, s2) - cscale(b2, s3) - cscale(b3, s1);
        const cpx<T> q3 = cscale(b1, s3) - cscale(b2, s1) + cscale(b3, s2);
        v[0] = x0 + a1 + a2 + a3;
        // forward: y[k] = p_k - i q_k, y[7-k] = p_k + i q_k
        v[1] = INV ? cadd_pi(p1, q1) : cadd_mi(p1, q1); v[6] = INV ? cadd_mi(p1, q1) : cadd_pi(p1, q1);
        v[2] = INV ? cadd_pi(p2, q2) : cadd_mi(p2, q2); v[5] = INV ? cadd_mi(p2, q2) : cadd_pi(p2, q2);
        v[3] = INV ? cadd_pi(p3, q3) : cadd_mi(p3, q3); v[4] = INV ? cadd_mi(p3, q3) : cadd_pi(p3, q3);
    }
};

// Gather from a small read-only table in global memory at a 32-bit element index: on the device the wave-uniform base is
// pinned in SGPRs and the lane part stays a 32-bit byte offset (`global_load v, v_off, s[base:base+1]`); written as
// tw[idx] the compiler forms a 64-bit VALU address (v_mad_i64_i32) per element.
template <typename V> TWX_HD V tw_load(const V* tw, unsigned idx) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long u = reinterpret_cast<unsigned long long>(tw);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    const __attribute__((address_space(1))) char* g = (const __attribute__((address_space(1))) char*)(((unsigned long long)hi << 32) | lo);
    return *(const __attribute__((address_space(1))) V*)(g + idx * (unsigned)sizeof(V));
#else
    return tw[idx];
#endif
}

// ------------------------------------------------------------------------------------------
// plan: length L as a product of up to four radices (Stockham stage order)
// ------------------------------------------------------------------------------------------
template <int L_, int R0_, int R1_ = 1, int R2_ = 1, int R3_ = 1> struct Plan {
    static constexpr int L = L_;
    static constexpr int S = (R1_ == 1) ? 1 : (R2_ == 1) ? 2 : (R3_ == 1) ? 3 : 4;
    static constexpr int radix(int s) { return s == 0 ? R0_ : s == 1 ? R1_ : s == 2 ? R2_ : R3_; }
    static constexpr int ns(int s) {  // product of the radices before stage s
        int p = 1;
        for (int i = 0; i < s; ++i) p *= radix(i);
        return p;
    }
    static constexpr int rmax() {
        int m = 1;
        for (int i = 0; i < S; ++i) m = radix(i) > m ? radix(i) : m;
        return m;
    }
    static constexpr int rmin() {
        int m = radix(0);
        for (int i = 1; i < S; ++i) m = radix(i) < m ? radix(i) : m;
        return m;
    }
    static constexpr int max_tasks = L / rmin();   // per column
    static_assert(R0_ * R1_ * R2_ * R3_ == L_, "radices must multiply to L");
};

// ------------------------------------------------------------------------------------------
// one tile = W columns of length-L data in LDS, element (p, c) at index pad(p)*W + c
// ------------------------------------------------------------------------------------------
template <class P, typename T, bool INV, int W, int PADQ> struct Tile {
    using C = cpx<T>;
    static constexpr int L = P::L;
    static TWX_HD int pad(int p) { return PADQ > 0 ? p + p / (PADQ > 0 ? PADQ : 1) : p; }
    static constexpr int lds_elems = (PADQ > 0 ? (L + (L - 1) / (PADQ > 0 ? PADQ : 1)) : L) * W;

    template <int s> static constexpr int R() { return P::radix(s); }
    template <int s> static constexpr int tasks() { return (L / P::radix(s)) * W; }  // per tile

    // Stockham indexing for stage s, per-column task j in [0, L/R)
    template <int s> static TWX_HD int in_pos(int j, int r) { return j + r * (L / P::radix(s)); }
    template <int s> static TWX_HD int out_pos(int j, int q) {
        constexpr int Ns = P::ns(s), Rr = P::radix(s);
        if constexpr (Ns == 1) return j * Rr + q;
        else if constexpr (Ns * Rr == L) return j + q * Ns;   // last stage: j < Ns
        else return (j / Ns) * (Ns * Rr) + (j % Ns) + q * Ns;
    }

    // physical (padded) LDS element index of logical position p; strength-reduced forms: the
    // r/q-dependent part is a compile-time multiple whenever the stride is a multiple of PADQ
    template <int s> static TWX_HD int in_base(int j) { return pad(j); }
    template <int s> static TWX_HD int in_idx(int base, int j, int r) {
        constexpr int Tt = L / P::radix(s);
        if constexpr (PADQ == 0) return base + r * Tt;
        else if constexpr (Tt % (PADQ > 0 ? PADQ : 1) == 0) return base + r * (Tt + Tt / (PADQ > 0 ? PADQ : 1));
        else return pad(j + r * Tt);
    }
    template <int s> static TWX_HD int out_base(int j) {
        constexpr int Ns = P::ns(s), Rr = P::radix(s);
        if constexpr (PADQ == 0) return out_pos<s>(j, 0);
        else if constexpr (Ns == 1 && Rr == PADQ) return j * (Rr + 1);
        else if constexpr (Ns * Rr == L) return pad(j);
        else if constexpr (Ns % (PADQ > 0 ? PADQ : 1) == 0) {
            constexpr int NR = Ns * Rr;
            return (j / Ns) * (NR + NR / PADQ) + pad(j % Ns);
        } else return 0;
    }
    template <int s> static TWX_HD int out_idx(int base, int j, int q) {
        constexpr int Ns = P::ns(s), Rr = P::radix(s);
        if constexpr (PADQ == 0) return base + q * Ns;
        else if constexpr (Ns == 1 && Rr == PADQ) return base + q;
        else if constexpr (Ns % (PADQ > 0 ? PADQ : 1) == 0) return base + q * (Ns + Ns / (PADQ > 0 ? PADQ : 1));
        else return pad(out_pos<s>(j, q));
    }

    // stage-s butterfly on v[0..R) (inputs already in natural r order, twiddled)
    template <int s> static TWX_HD void bfly(C* v) { Bfly<T, P::radix(s), INV>::run(v); }

    // read the R inputs of task (j, c) of stage s (s >= 1) from LDS and apply the stage twiddles
    // tw = table of exp(-2*pi*i*m/L), m in [0, L)
    template <int s> static TWX_HD void load_lds(const C* lds, const C* tw, int j, int c, C* v) {
        constexpr int Ns = P::ns(s), Rr = P::radix(s);
        constexpr int step = L / (Ns * Rr);
        const int jm = (Ns * Rr == L) ? j : (j % Ns);
        const int ib = in_base<s>(j);
        TWX_UNROLL
        for (int r = 0; r < Rr; ++r) v[r] = lds[in_idx<s>(ib, j, r) * W + c];
        TWX_UNROLL
        for (int r = 1; r < Rr; ++r) {
            C w = tw_load(tw, (unsigned)(jm * r * step));
            v[r] = INV ? cmulc(v[r], w) : cmul(v[r], w);
        }
    }
    // load_lds in two halves: the twiddle gather touches only the read-only table, so it can be issued BEFORE the
    // workgroup barrier that publishes the previous stage's LDS writes (its latency then overlaps the barrier wait)
    template <int s> static TWX_HD void load_tw(const C* tw, int j, C* twr /*[R-1]*/) {
        constexpr int Ns = P::ns(s), Rr = P::radix(s);
        constexpr int step = L / (Ns * Rr);
        const int jm = (Ns * Rr == L) ? j : (j % Ns);
        TWX_UNROLL
        for (int r = 1; r < Rr; ++r) twr[r - 1] = tw_load(tw, (unsigned)(jm * r * step));
    }
    template <int s> static TWX_HD void load_lds_tw(const C* lds, const C* twr, int j, int c, C* v) {
        constexpr int Rr = P::radix(s);
        const int ib = in_base<s>(j);
        TWX_UNROLL
        for (int r = 0; r < Rr; ++r) v[r] = lds[in_idx<s>(ib, j, r) * W + c];
        TWX_UNROLL
        for (int r = 1; r < Rr; ++r) v[r] = INV ? cmulc(v[r], twr[r - 1]) : cmul(v[r], twr[r - 1]);
    }
    template <int s> static TWX_HD void store_lds(C* lds, int j, int c, const C* v) {
        constexpr int Rr = P::radix(s);
        const int ob = out_base<s>(j);
        TWX_UNROLL
        for (int q = 0; q < Rr; ++q) lds[out_idx<s>(ob, j, q) * W + c] = v[q];
    }
};

// ------------------------------------------------------------------------------------------
// compact per-stage twiddle tables (kept in LDS by the row kernels)
//
// Stage s (radix R_s, Ns = R_0*...*R_{s-1}) needs W_{Ns*R_s}^{jm*r}, jm = j mod Ns.  With the
// mixed-radix digits of jm (x_0 = jm mod R_0, x_1 = (jm / R_0) mod R_1, ...) it factors into
//   prod_d  W_{den_d}^{x_d * r},   den_d = R_s * R_d * R_{d+1} * ... * R_{s-1}
// so every factor comes from a tiny 2-D table tab[s][d][r][x_d] (R_s x R_d entries), stored
// r-major so that consecutive lanes (consecutive x_0) read consecutive entries.
// ------------------------------------------------------------------------------------------
template <class P> struct StageTabs {
    static constexpr int off(int s, int d) {        // entries before table (s, d)
        int o = 0;
        for (int ss = 1; ss < P::S; ++ss)
            for (int dd = 0; dd < ss; ++dd) {
                if (ss == s && dd == d) return o;
                o += P::radix(ss) * P::radix(dd);
            }
        return o;
    }
    static constexpr int total = off(P::S, 0);
    static constexpr int den(int s, int d) {
        int v = P::radix(s);
        for (int i = d; i < s; ++i) v *= P::radix(i);
        return v;
    }
};

template <class P, typename T, bool INV, int PADQ> struct RowTile : Tile<P, T, INV, 1, PADQ> {
    using Base = Tile<P, T, INV, 1, PADQ>;
    using C = cpx<T>;
    // read the inputs of task j of stage s (s >= 1) from LDS and apply twiddles from LDS tables
    template <int s> static TWX_HD void load_lds_tab(const C* lds, const C* tabs, int j, C* v) {
        constexpr int Ns = P::ns(s), Rr = P::radix(s);
        const int jm = (Ns * Rr == P::L) ? j : (j % Ns);
        int x[4]; int rem = jm;
        TWX_UNROLL
        for (int d = 0; d < s; ++d) { x[d] = rem % P::radix(d); rem /= P::radix(d); }
        const int ib = Base::template in_base<s>(j);
        v[0] = lds[Base::template in_idx<s>(ib, j, 0)];
        TWX_UNROLL
        for (int r = 1; r < Rr; ++r) {
            C w = tabs[StageTabs<P>::off(s, 0) + r * P::radix(0) + x[0]];
            TWX_UNROLL
            for (int d = 1; d < s; ++d) w = cmul(w, tabs[StageTabs<P>::off(s, d) + r * P::radix(d) + x[d]]);
            const C u = lds[Base::template in_idx<s>(ib, j, r)];
            v[r] = INV ? cmulc(u, w) : cmul(u, w);
#if defined(__HIP_DEVICE_COMPILE__) && defined(TWX_SCHED_GROUP)
            if (r % TWX_SCHED_GROUP == 0) __builtin_amdgcn_sched_barrier(0);
#endif
        }
    }
    // the same twiddles kept by the caller: tw[r-1] = W_{Ns*R}^{jm*r} (forward sign), r = 1..R-1
    template <int s> static TWX_HD void stage_twiddles(const C* tabs, int j, C* tw) {
        constexpr int Ns = P::ns(s), Rr = P::radix(s);
        const int jm = (Ns * Rr == P::L) ? j : (j % Ns);
        int x[4]; int rem = jm;
        TWX_UNROLL
        for (int d = 0; d < s; ++d) { x[d] = rem % P::radix(d); rem /= P::radix(d); }
        TWX_UNROLL
        for (int r = 1; r < Rr; ++r) {
            C w = tabs[StageTabs<P>::off(s, 0) + r * P::radix(0) + x[0]];
            TWX_UNROLL
            for (int d = 1; d < s; ++d) w = cmul(w, tabs[StageTabs<P>::off(s, d) + r * P::radix(d) + x[d]]);
            tw[r - 1] = w;
        }
    }
};

// ------------------------------------------------------------------------------------------
// RowD: row transform in decimation-in-frequency (forward) / decimation-in-time (inverse) form,
// L = R0 * R * R.  Only the stride-(L/R0) stage exchanges data between all threads of the
// workgroup; the two inner stages work on R0 independent blocks of R*R elements, and each block
// is owned by R lanes of ONE wave (64/R blocks per wave), so the inner exchange needs no
// workgroup barrier.  One workgroup barrier per forward transform, two per inverse transform
// (the Stockham form above needs four).
//
//   forward:  stage 0  thread t=(i1,i2)  x[t + (L/R0) r] → DFT_R0 → ·W_L^{t q0}         → lds(q0,i1,i2)   | barrier
//             stage 1  thread (q0,i2)    lds(q0,r,i2)     → DFT_R  → ·W_{R^2}^{i2 q1}    → lds(q0,q1,i2)   | wave
//             stage 2  thread (q0,q1)    lds(q0,q1,r)     → DFT_R  → X[q0 + R0 q1 + R0 R q2] in v[q2]
//   inverse:  stage A  thread (q0,q1)    v[q2]            → IDFT_R → ·W_{R^2}^{-q1 a}    → lds(q0,q1,a)    | wave
//             stage B  thread (q0,a)     lds(q0,r,a)      → IDFT_R → ·W_L^{-q0(a + R b)} → lds(q0,b,a)     | barrier
//             stage C  thread t=a+R b    lds(r,b,a)       → IDFT_R0 → z[t + (L/R0) c] in v[c]
// lds(q0,i1,i2) = q0*R*(R+1) + i1*(R+1) + i2  (rows padded by one element: stride-(R+1) accesses
// of stages 2/A are bank-conflict free).  Tables (forward sign): ta[q][x] = W_{R^2}^{x q} (R x R),
// tb[q0][x] = W_L^{x q0}, tc[q0][x] = W_{L/R}^{x q0} (R0 x R each).
// ------------------------------------------------------------------------------------------
template <class P, typename T> struct RowD {
    using C = cpx<T>;
    static constexpr int L = P::L;
    static constexpr int R0 = (P::S == 3) ? P::radix(0) : 1;
    static constexpr int R = P::radix(P::S - 1);
    static_assert(P::S == 2 || P::S == 3, "RowD needs L = [R0 *] R * R");
    static_assert(P::radix(P::S - 2) == R && R0 * R * R == L, "RowD needs equal inner radices");
    static constexpr int M = R * R;               // block length = L / R0 = stage-0 / stage-C tasks
    static constexpr int BK = R * (R + 1);        // padded block
    static constexpr int lds_elems = R0 * BK;
    static constexpr int BPW = 64 / R;            // blocks per wave
    static constexpr int NT_BLK = ((R0 + BPW - 1) / BPW) * 64;   // threads needed by the block stages
    static constexpr int NT_MIN = (M > NT_BLK ? ((M + 63) / 64) * 64 : NT_BLK);
    static constexpr int tab_a = 0, tab_b = R * R, tab_c = R * R + R0 * R, tab_total = R * R + 2 * R0 * R;

    static TWX_HD int phys(int q0, int i1, int i2) { return q0 * BK + i1 * (R + 1) + i2; }
    static TWX_HD bool blk_map(int tid, int& q0, int& i) {
        const int w = tid >> 6, l = tid & 63;
        q0 = w * BPW + l / R; i = l % R;
        return l < BPW * R && q0 < R0;
    }
    // ---- forward
    static TWX_HD void f0_twiddle_store(C* lds, const C* tabs, int t, C* v) {     // after DFT_R0 of the loaded inputs
        const int i1 = t / R, i2 = t % R;
        lds[phys(0, i1, i2)] = v[0];
        TWX_UNROLL
        for (int q0 = 1; q0 < R0; ++q0) {
            lds[phys(q0, i1, i2)] = cmul3(v[q0], tabs[tab_b + q0 * R + i2], tabs[tab_c + q0 * R + i1]);
        }
    }
    static TWX_HD void f1(C* lds, const C* tabs, int q0, int i2, C* v) {
        TWX_UNROLL
        for (int r = 0; r < R; ++r) v[r] = lds[phys(q0, r, i2)];
        Bfly<T, R, false>::run(v);
        lds[phys(q0, 0, i2)] = v[0];
        TWX_UNROLL
        for (int q1 = 1; q1 < R; ++q1) lds[phys(q0, q1, i2)] = cmul(v[q1], tabs[tab_a + q1 * R + i2]);
    }
    static TWX_HD void f2(const C* lds, int q0, int q1, C* v) {
        TWX_UNROLL
        for (int r = 0; r < R; ++r) v[r] = lds[phys(q0, q1, r)];
        Bfly<T, R, false>::run(v);
    }
    static TWX_HD int k_of(int q0, int q1, int q2) { return q0 + R0 * q1 + R0 * R * q2; }   // bin held in v[q2] after f2
    // ---- inverse
    static TWX_HD void iA(C* lds, const C* tabs, int q0, int q1, C* v) {
        Bfly<T, R, true>::run(v);
        lds[phys(q0, q1, 0)] = v[0];
        TWX_UNROLL
        for (int a = 1; a < R; ++a) lds[phys(q0, q1, a)] = cmulc(v[a], tabs[tab_a + a * R + q1]);
    }
    // iA in two halves: the register part touches only the (read-only) tables, so it may run before the
    // workgroup barrier that frees the data region; the store part must follow it
    static TWX_HD void iA_pre(const C* tabs, int q1, C* v) {
        Bfly<T, R, true>::run(v);
        TWX_UNROLL
        for (int a = 1; a < R; ++a) v[a] = cmulc(v[a], tabs[tab_a + a * R + q1]);
    }
    static TWX_HD void iA_store(C* lds, int q0, int q1, const C* v) {
        TWX_UNROLL
        for (int a = 0; a < R; ++a) lds[phys(q0, q1, a)] = v[a];
    }
    static TWX_HD void iB(C* lds, const C* tabs, int q0, int a, C* v) {
        TWX_UNROLL
        for (int r = 0; r < R; ++r) v[r] = lds[phys(q0, r, a)];
        Bfly<T, R, true>::run(v);
        if (R0 == 1) {
            TWX_UNROLL
            for (int b = 0; b < R; ++b) lds[phys(q0, b, a)] = v[b];
        } else {
            const C wa = tabs[tab_b + q0 * R + a];
            TWX_UNROLL
            for (int b = 0; b < R; ++b) lds[phys(q0, b, a)] = cmulc3(v[b], wa, tabs[tab_c + q0 * R + b]);
        }
    }
    // stage B with the factors handed over ready-made: lds(q0,b,a) = IDFT(...)[b] * wa * tabs[tab_c + q0 R + b] — the caller has
    // folded everything that does not depend on q0's partner index into them (k_rowd<MID>: conjugated twiddles, the output
    // twiddle W_N^{-k1 (a + R b)} and the interpolation ramp of the phase), so stage C needs one product per output less
    static TWX_HD void iB_folded(C* lds, const C* tabs, int q0, int a, C wa, C* v) {
        TWX_UNROLL
        for (int r = 0; r < R; ++r) v[r] = lds[phys(q0, r, a)];
        Bfly<T, R, true>::run(v);
        TWX_UNROLL
        for (int b = 0; b < R; ++b) lds[phys(q0, b, a)] = cmul3(v[b], wa, tabs[tab_c + q0 * R + b]);
    }
    // R0 == 1: stage B without the write-back — thread (block, a) ends with v[b] = z[a + R b]
    static TWX_HD void iB_keep(const C* lds, int q0, int a, C* v) {
        TWX_UNROLL
        for (int r = 0; r < R; ++r) v[r] = lds[phys(q0, r, a)];
        Bfly<T, R, true>::run(v);
    }
    static TWX_HD void iC(const C* lds, int t, C* v) {                            // → v[c] = z[t + M c]
        const int b = t / R, a = t % R;
        TWX_UNROLL
        for (int r = 0; r < R0; ++r) v[r] = lds[phys(r, b, a)];
        Bfly<T, R0, true>::run(v);
    }
};

}  // namespace twx
