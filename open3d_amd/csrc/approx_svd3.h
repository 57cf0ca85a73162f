// Approximate 3x3 SVD and the pseudo-inverse solve built on it, as the
// reference's EstimateColorGradients uses them
// (cpp/open3d/core/linalg/kernel/SVD3x3.h: svd3x3 :86-1129 / :1131-2168 -- the
// float and double specialisations are the same text -- and solve_svd3x3
// :2170-2215; McAdams, Selle, Tamstorf, Teran, Sifakis 2011):
//
//   S = A^T A;  4 sweeps of Jacobi conjugation on S in the fixed pair order
//   (1,2), (2,3), (3,1), the rotation accumulated as a quaternion, each angle
//   from the unnormalised (ch, sh) = (s_pp - s_qq, s_pq / 2) with a clamp to
//   pi/8 when ch^2 <= (3 + 2 sqrt 2) sh^2;  V = R(q / |q|);  B = A V;  columns
//   of B and V sorted by decreasing norm with a sign flip on every swap;
//   B = U Sigma by three Givens rotations (QR), U accumulated from I.
//
// Bit parity with the reference needs more than the algorithm:
//   * every intermediate is a union {scalar f; unsigned int ui;} and the
//     conditionals are 32-bit masks on .ui. For float that is a plain select;
//     for double .ui aliases only the LOW half of the mantissa, so a "select"
//     keeps the high half of whatever the register held, an XOR-swap exchanges
//     low halves only, and (-2.0).ui & mask is always -2.0. The reference's
//     Float64 path is therefore not an SVD of its input in general (it yields
//     NaN on ~9 % of random matrices); it is reproduced as it is, and
//     O3DMI_EXACT_COLOR_GRADIENTS=1 selects an exact solve instead
//     (normals.hip);
//   * registers are reused in a fixed order and stale contents leak through
//     the masks, so each statement writes the register the reference writes;
//   * 1 / sqrt(x) is (scalar)(1.0 / sqrt(x)) (float sqrt, float64 division),
//     with one Newton step for the quaternion and the Givens angles;
//   * 1e-20 and 5.828... are double literals (products / comparisons formed in
//     float64 and narrowed), 1e-12 is narrowed to the scalar first.
// The three Jacobi conjugations and the three Givens steps are one function
// each, called with the registers in the roles the unrolled reference gives
// them.

#pragma once

#include <hip/hip_runtime.h>

#include <cmath>

namespace o3dmi {
namespace svd3 {

template <typename T>
union Reg {
    T f;
    unsigned int ui;
};

constexpr unsigned int kOne = 1065353216u;              // 1.0f
constexpr unsigned int kSinePiOver8 = 1053028117u;      // sin(pi/8) as float
constexpr unsigned int kCosinePiOver8 = 1064076127u;    // cos(pi/8) as float
constexpr double kTiny = 1.e-20;
constexpr double kFourGammaSquared = 5.8284273147583007813;

template <typename T>
__host__ __device__ __forceinline__ T Rsqrt(T x) {
    return (T)(1.0 / sqrt(x));
}
template <typename T>
__host__ __device__ __forceinline__ T Max(T a, T b) {  // std::max
    return (a < b) ? b : a;
}

// One Jacobi conjugation. (a, b, c) = (s_pp, s_qp, s_qq) of the rotated pair,
// d = the third diagonal entry, (e, g) = the two off-diagonal entries that mix.
// kAxis: which quaternion component takes +sh (1: z, 2: x, 3: y).
template <typename T, int kAxis>
__host__ __device__ __forceinline__ void JacobiConjugation(Reg<T>& a, Reg<T>& b, Reg<T>& c, Reg<T>& d,
                              Reg<T>& e, Reg<T>& g, Reg<T>& qs,
                              Reg<T>& qx, Reg<T>& qy, Reg<T>& qz,
                              Reg<T>& cs, Reg<T>& sn, Reg<T>& ch, Reg<T>& sh,
                              Reg<T>& t1, Reg<T>& t2, Reg<T>& t3,
                              Reg<T>& t4, Reg<T>& t5) {
    sh.f = b.f * 0.5f;
    t5.f = a.f - c.f;

    t2.f = sh.f * sh.f;
    t1.ui = (t2.f >= kTiny) ? 0xffffffff : 0;
    sh.ui = t1.ui & sh.ui;
    ch.ui = t1.ui & t5.ui;
    t2.ui = ~t1.ui & kOne;
    ch.ui = ch.ui | t2.ui;

    t1.f = sh.f * sh.f;
    t2.f = ch.f * ch.f;
    t3.f = t1.f + t2.f;
    t4.f = Rsqrt(t3.f);

    sh.f = t4.f * sh.f;
    ch.f = t4.f * ch.f;
    t1.f = kFourGammaSquared * t1.f;
    t1.ui = (t2.f <= t1.f) ? 0xffffffff : 0;

    t2.ui = kSinePiOver8 & t1.ui;
    sh.ui = ~t1.ui & sh.ui;
    sh.ui = sh.ui | t2.ui;
    t2.ui = kCosinePiOver8 & t1.ui;
    ch.ui = ~t1.ui & ch.ui;
    ch.ui = ch.ui | t2.ui;

    t1.f = sh.f * sh.f;
    t2.f = ch.f * ch.f;
    cs.f = t2.f - t1.f;
    sn.f = ch.f * sh.f;
    sn.f = sn.f + sn.f;

    // the conjugation itself
    t3.f = t1.f + t2.f;
    d.f = d.f * t3.f;
    e.f = e.f * t3.f;
    g.f = g.f * t3.f;
    d.f = d.f * t3.f;

    t1.f = sn.f * e.f;
    t2.f = sn.f * g.f;
    e.f = cs.f * e.f;
    g.f = cs.f * g.f;
    e.f = t2.f + e.f;
    g.f = g.f - t1.f;

    t2.f = sn.f * sn.f;
    t1.f = c.f * t2.f;
    t3.f = a.f * t2.f;
    t4.f = cs.f * cs.f;
    a.f = a.f * t4.f;
    c.f = c.f * t4.f;
    a.f = a.f + t1.f;
    c.f = c.f + t3.f;
    t4.f = t4.f - t2.f;
    t2.f = b.f + b.f;
    b.f = b.f * t4.f;
    t4.f = cs.f * sn.f;
    t2.f = t2.f * t4.f;
    t5.f = t5.f * t4.f;
    a.f = a.f + t2.f;
    b.f = b.f - t5.f;
    c.f = c.f - t2.f;

    // cumulative rotation, as a quaternion
    t1.f = sh.f * qx.f;
    t2.f = sh.f * qy.f;
    t3.f = sh.f * qz.f;
    sh.f = sh.f * qs.f;

    qs.f = ch.f * qs.f;
    qx.f = ch.f * qx.f;
    qy.f = ch.f * qy.f;
    qz.f = ch.f * qz.f;

    if (kAxis == 1) {
        qz.f = qz.f + sh.f;
        qs.f = qs.f - t3.f;
        qx.f = qx.f + t2.f;
        qy.f = qy.f - t1.f;
    } else if (kAxis == 2) {
        qx.f = qx.f + sh.f;
        qs.f = qs.f - t1.f;
        qy.f = qy.f + t3.f;
        qz.f = qz.f - t2.f;
    } else {
        qy.f = qy.f + sh.f;
        qs.f = qs.f - t2.f;
        qz.f = qz.f + t1.f;
        qx.f = qx.f - t3.f;
    }
}

// Row r of A times V: (x, y, z) <- (x, y, z) V, through t1..3.
template <typename T>
__host__ __device__ __forceinline__ void RowTimesV(Reg<T>& x, Reg<T>& y, Reg<T>& z, const Reg<T>& v11,
                      const Reg<T>& v21, const Reg<T>& v31,
                      const Reg<T>& v12, const Reg<T>& v22,
                      const Reg<T>& v32, const Reg<T>& v13,
                      const Reg<T>& v23, const Reg<T>& v33, Reg<T>& t1,
                      Reg<T>& t2, Reg<T>& t3) {
    t2.f = y.f;
    t3.f = z.f;
    y.f = v12.f * x.f;
    z.f = v13.f * x.f;
    x.f = v11.f * x.f;
    t1.f = v21.f * t2.f;
    x.f = x.f + t1.f;
    t1.f = v31.f * t3.f;
    x.f = x.f + t1.f;
    t1.f = v22.f * t2.f;
    y.f = y.f + t1.f;
    t1.f = v32.f * t3.f;
    y.f = y.f + t1.f;
    t1.f = v23.f * t2.f;
    z.f = z.f + t1.f;
    t1.f = v33.f * t3.f;
    z.f = z.f + t1.f;
}

template <typename T>
__host__ __device__ __forceinline__ void MaskedSwap(Reg<T>& x, Reg<T>& y, Reg<T>& t5,
                       const Reg<T>& t4) {
    t5.ui = x.ui ^ y.ui;
    t5.ui = t5.ui & t4.ui;
    x.ui = x.ui ^ t5.ui;
    y.ui = y.ui ^ t5.ui;
}

// Conditional swap of columns i and j of B and V (and of their squared norms
// ni, nj) when ni < nj, then the sign flip of column `neg` (n1..n3 = the
// column the reference multiplies).
template <typename T>
__host__ __device__ __forceinline__ void SortColumns(Reg<T>& ai1, Reg<T>& aj1, Reg<T>& ai2, Reg<T>& aj2,
                        Reg<T>& ai3, Reg<T>& aj3, Reg<T>& vi1, Reg<T>& vj1,
                        Reg<T>& vi2, Reg<T>& vj2, Reg<T>& vi3, Reg<T>& vj3,
                        Reg<T>& ni, Reg<T>& nj, Reg<T>& n1, Reg<T>& n2,
                        Reg<T>& n3, Reg<T>& w1, Reg<T>& w2, Reg<T>& w3,
                        Reg<T>& t4, Reg<T>& t5) {
    t4.ui = (ni.f < nj.f) ? 0xffffffff : 0;
    MaskedSwap(ai1, aj1, t5, t4);
    MaskedSwap(ai2, aj2, t5, t4);
    MaskedSwap(ai3, aj3, t5, t4);
    MaskedSwap(vi1, vj1, t5, t4);
    MaskedSwap(vi2, vj2, t5, t4);
    MaskedSwap(vi3, vj3, t5, t4);
    MaskedSwap(ni, nj, t5, t4);
    t5.f = -2.f;
    t5.ui = t5.ui & t4.ui;
    t4.f = 1.f;
    t4.f = t4.f + t5.f;
    n1.f = n1.f * t4.f;
    n2.f = n2.f * t4.f;
    n3.f = n3.f * t4.f;
    w1.f = w1.f * t4.f;
    w2.f = w2.f * t4.f;
    w3.f = w3.f * t4.f;
}

template <typename T>
__host__ __device__ __forceinline__ void Rotate(Reg<T>& x, Reg<T>& y, const Reg<T>& cs, const Reg<T>& sn,
                   Reg<T>& t1, Reg<T>& t2) {
    t1.f = sn.f * x.f;
    t2.f = sn.f * y.f;
    x.f = cs.f * x.f;
    y.f = cs.f * y.f;
    x.f = x.f + t2.f;
    y.f = y.f - t1.f;
}

// One Givens step of the QR factorisation: zero `below` against `pivot`,
// rotating the three row pairs of B and the three column pairs of U.
template <typename T>
__host__ __device__ __forceinline__ void GivensQR(const Reg<T>& pivot, const Reg<T>& below, Reg<T>& r1x,
                     Reg<T>& r1y, Reg<T>& r2x, Reg<T>& r2y, Reg<T>& r3x,
                     Reg<T>& r3y, Reg<T>& u1x, Reg<T>& u1y, Reg<T>& u2x,
                     Reg<T>& u2y, Reg<T>& u3x, Reg<T>& u3y, T gsmall_number,
                     Reg<T>& cs, Reg<T>& sn, Reg<T>& ch, Reg<T>& sh,
                     Reg<T>& t1, Reg<T>& t2, Reg<T>& t3,
                     Reg<T>& t4, Reg<T>& t5) {
    sh.f = below.f * below.f;
    sh.ui = (sh.f >= gsmall_number) ? 0xffffffff : 0;
    sh.ui = sh.ui & below.ui;

    t5.f = 0.f;
    ch.f = t5.f - pivot.f;
    ch.f = Max(ch.f, pivot.f);
    ch.f = Max(ch.f, gsmall_number);
    t5.ui = (pivot.f >= t5.f) ? 0xffffffff : 0;

    t1.f = ch.f * ch.f;
    t2.f = sh.f * sh.f;
    t2.f = t1.f + t2.f;
    t1.f = Rsqrt(t2.f);

    t4.f = t1.f * 0.5f;
    t3.f = t1.f * t4.f;
    t3.f = t1.f * t3.f;
    t3.f = t2.f * t3.f;
    t1.f = t1.f + t4.f;
    t1.f = t1.f - t3.f;
    t1.f = t1.f * t2.f;

    ch.f = ch.f + t1.f;

    t1.ui = ~t5.ui & sh.ui;
    t2.ui = ~t5.ui & ch.ui;
    ch.ui = t5.ui & ch.ui;
    sh.ui = t5.ui & sh.ui;
    ch.ui = ch.ui | t1.ui;
    sh.ui = sh.ui | t2.ui;

    t1.f = ch.f * ch.f;
    t2.f = sh.f * sh.f;
    t2.f = t1.f + t2.f;
    t1.f = Rsqrt(t2.f);

    t4.f = t1.f * 0.5f;
    t3.f = t1.f * t4.f;
    t3.f = t1.f * t3.f;
    t3.f = t2.f * t3.f;
    t1.f = t1.f + t4.f;
    t1.f = t1.f - t3.f;

    ch.f = ch.f * t1.f;
    sh.f = sh.f * t1.f;

    cs.f = ch.f * ch.f;
    sn.f = sh.f * sh.f;
    cs.f = cs.f - sn.f;
    sn.f = sh.f * ch.f;
    sn.f = sn.f + sn.f;

    Rotate(r1x, r1y, cs, sn, t1, t2);
    Rotate(r2x, r2y, cs, sn, t1, t2);
    Rotate(r3x, r3y, cs, sn, t1, t2);
    Rotate(u1x, u1y, cs, sn, t1, t2);
    Rotate(u2x, u2y, cs, sn, t1, t2);
    Rotate(u3x, u3y, cs, sn, t1, t2);
}

// svd3x3, SVD3x3.h:86-1129: A = U diag(S) V^T (row-major 3x3 arrays).
template <typename T>
__host__ __device__ __forceinline__ void Svd3x3(const T* A_3x3, T* U_3x3, T* S_3x1, T* V_3x3) {
    T gsmall_number = 1.e-12;

    Reg<T> a11, a21, a31, a12, a22, a32, a13, a23, a33;
    Reg<T> u11, u21, u31, u12, u22, u32, u13, u23, u33;
    Reg<T> v11, v21, v31, v12, v22, v32, v13, v23, v33;
    Reg<T> cs, sn, ch, sh;
    Reg<T> t1, t2, t3, t4, t5;
    Reg<T> s11, s21, s31, s22, s32, s33;
    Reg<T> qs, qx, qy, qz;
    // The reference leaves these uninitialised; only ch matters (Float64:
    // the high half survives the first masked write). Zero, as a fresh stack.
    ch.f = 0;
    sh.f = 0;
    cs.f = 0;
    sn.f = 0;
    t1.f = 0;
    t2.f = 0;
    t3.f = 0;
    t4.f = 0;
    t5.f = 0;

    a11.f = A_3x3[0];
    a12.f = A_3x3[1];
    a13.f = A_3x3[2];
    a21.f = A_3x3[3];
    a22.f = A_3x3[4];
    a23.f = A_3x3[5];
    a31.f = A_3x3[6];
    a32.f = A_3x3[7];
    a33.f = A_3x3[8];

    // normal equations matrix S = A^T A (lower triangle)
    s11.f = a11.f * a11.f;
    t1.f = a21.f * a21.f;
    s11.f = t1.f + s11.f;
    t1.f = a31.f * a31.f;
    s11.f = t1.f + s11.f;

    s21.f = a12.f * a11.f;
    t1.f = a22.f * a21.f;
    s21.f = t1.f + s21.f;
    t1.f = a32.f * a31.f;
    s21.f = t1.f + s21.f;

    s31.f = a13.f * a11.f;
    t1.f = a23.f * a21.f;
    s31.f = t1.f + s31.f;
    t1.f = a33.f * a31.f;
    s31.f = t1.f + s31.f;

    s22.f = a12.f * a12.f;
    t1.f = a22.f * a22.f;
    s22.f = t1.f + s22.f;
    t1.f = a32.f * a32.f;
    s22.f = t1.f + s22.f;

    s32.f = a13.f * a12.f;
    t1.f = a23.f * a22.f;
    s32.f = t1.f + s32.f;
    t1.f = a33.f * a32.f;
    s32.f = t1.f + s32.f;

    s33.f = a13.f * a13.f;
    t1.f = a23.f * a23.f;
    s33.f = t1.f + s33.f;
    t1.f = a33.f * a33.f;
    s33.f = t1.f + s33.f;

    qs.f = 1.f;
    qx.f = 0.f;
    qy.f = 0.f;
    qz.f = 0.f;

    // symmetric eigenproblem: 4 Jacobi sweeps, pairs (1,2), (2,3), (3,1)
    for (int i = 0; i < 4; i++) {
        JacobiConjugation<T, 1>(s11, s21, s22, s33, s31, s32, qs,
                                qx, qy, qz, cs, sn, ch, sh, t1,
                                t2, t3, t4, t5);
        JacobiConjugation<T, 2>(s22, s32, s33, s11, s21, s31, qs,
                                qx, qy, qz, cs, sn, ch, sh, t1,
                                t2, t3, t4, t5);
        JacobiConjugation<T, 3>(s33, s31, s11, s22, s32, s21, qs,
                                qx, qy, qz, cs, sn, ch, sh, t1,
                                t2, t3, t4, t5);
    }

    // normalise the quaternion (rsqrt + one Newton step), V = R(q)
    t2.f = qs.f * qs.f;
    t1.f = qx.f * qx.f;
    t2.f = t1.f + t2.f;
    t1.f = qy.f * qy.f;
    t2.f = t1.f + t2.f;
    t1.f = qz.f * qz.f;
    t2.f = t1.f + t2.f;

    t1.f = Rsqrt(t2.f);
    t4.f = t1.f * 0.5f;
    t3.f = t1.f * t4.f;
    t3.f = t1.f * t3.f;
    t3.f = t2.f * t3.f;
    t1.f = t1.f + t4.f;
    t1.f = t1.f - t3.f;

    qs.f = qs.f * t1.f;
    qx.f = qx.f * t1.f;
    qy.f = qy.f * t1.f;
    qz.f = qz.f * t1.f;

    t1.f = qx.f * qx.f;
    t2.f = qy.f * qy.f;
    t3.f = qz.f * qz.f;
    v11.f = qs.f * qs.f;
    v22.f = v11.f - t1.f;
    v33.f = v22.f - t2.f;
    v33.f = v33.f + t3.f;
    v22.f = v22.f + t2.f;
    v22.f = v22.f - t3.f;
    v11.f = v11.f + t1.f;
    v11.f = v11.f - t2.f;
    v11.f = v11.f - t3.f;
    t1.f = qx.f + qx.f;
    t2.f = qy.f + qy.f;
    t3.f = qz.f + qz.f;
    v32.f = qs.f * t1.f;
    v13.f = qs.f * t2.f;
    v21.f = qs.f * t3.f;
    t1.f = qy.f * t1.f;
    t2.f = qz.f * t2.f;
    t3.f = qx.f * t3.f;
    v12.f = t1.f - v21.f;
    v23.f = t2.f - v32.f;
    v31.f = t3.f - v13.f;
    v21.f = t1.f + v21.f;
    v32.f = t2.f + v32.f;
    v13.f = t3.f + v13.f;

    // B = A V
    RowTimesV(a11, a12, a13, v11, v21, v31, v12, v22, v32, v13, v23,
              v33, t1, t2, t3);
    RowTimesV(a21, a22, a23, v11, v21, v31, v12, v22, v32, v13, v23,
              v33, t1, t2, t3);
    RowTimesV(a31, a32, a33, v11, v21, v31, v12, v22, v32, v13, v23,
              v33, t1, t2, t3);

    // squared column norms, then the three conditional swaps
    t1.f = a11.f * a11.f;
    t4.f = a21.f * a21.f;
    t1.f = t1.f + t4.f;
    t4.f = a31.f * a31.f;
    t1.f = t1.f + t4.f;

    t2.f = a12.f * a12.f;
    t4.f = a22.f * a22.f;
    t2.f = t2.f + t4.f;
    t4.f = a32.f * a32.f;
    t2.f = t2.f + t4.f;

    t3.f = a13.f * a13.f;
    t4.f = a23.f * a23.f;
    t3.f = t3.f + t4.f;
    t4.f = a33.f * a33.f;
    t3.f = t3.f + t4.f;

    // (1,2): flips column 2; (1,3): flips column 1; (2,3): flips column 3
    SortColumns(a11, a12, a21, a22, a31, a32, v11, v12, v21, v22,
                v31, v32, t1, t2, a12, a22, a32, v12, v22, v32,
                t4, t5);
    SortColumns(a11, a13, a21, a23, a31, a33, v11, v13, v21, v23,
                v31, v33, t1, t3, a11, a21, a31, v11, v21, v31,
                t4, t5);
    SortColumns(a12, a13, a22, a23, a32, a33, v12, v13, v22, v23,
                v32, v33, t2, t3, a13, a23, a33, v13, v23, v33,
                t4, t5);

    // QR: B = U Sigma
    u11.f = 1.f;
    u12.f = 0.f;
    u13.f = 0.f;
    u21.f = 0.f;
    u22.f = 1.f;
    u23.f = 0.f;
    u31.f = 0.f;
    u32.f = 0.f;
    u33.f = 1.f;

    GivensQR(a11, a21, a11, a21, a12, a22, a13, a23, u11, u12, u21,
             u22, u31, u32, gsmall_number, cs, sn, ch, sh, t1, t2,
             t3, t4, t5);
    GivensQR(a11, a31, a11, a31, a12, a32, a13, a33, u11, u13, u21,
             u23, u31, u33, gsmall_number, cs, sn, ch, sh, t1, t2,
             t3, t4, t5);
    GivensQR(a22, a32, a21, a31, a22, a32, a23, a33, u12, u13, u22,
             u23, u32, u33, gsmall_number, cs, sn, ch, sh, t1, t2,
             t3, t4, t5);

    V_3x3[0] = v11.f;
    V_3x3[1] = v12.f;
    V_3x3[2] = v13.f;
    V_3x3[3] = v21.f;
    V_3x3[4] = v22.f;
    V_3x3[5] = v23.f;
    V_3x3[6] = v31.f;
    V_3x3[7] = v32.f;
    V_3x3[8] = v33.f;

    U_3x3[0] = u11.f;
    U_3x3[1] = u12.f;
    U_3x3[2] = u13.f;
    U_3x3[3] = u21.f;
    U_3x3[4] = u22.f;
    U_3x3[5] = u23.f;
    U_3x3[6] = u31.f;
    U_3x3[7] = u32.f;
    U_3x3[8] = u33.f;

    S_3x1[0] = a11.f;
    S_3x1[1] = a22.f;
    S_3x1[2] = a33.f;
}

// solve_svd3x3, SVD3x3.h:2170-2215: x = V Sigma^+ U^T b with singular values
// below 1e-10 (narrowed to the scalar) dropped.
template <typename T>
__host__ __device__ __forceinline__ void SolveSvd3x3(const T* A_3x3, const T* B_3x1, T* X_3x1) {
    T U[9], V[9], S[3];
    Svd3x3<T>(A_3x3, U, S, V);
    const T epsilon = 1e-10;
    S[0] = fabs(S[0]) < epsilon ? 0 : 1.0 / S[0];
    S[1] = fabs(S[1]) < epsilon ? 0 : 1.0 / S[1];
    S[2] = fabs(S[2]) < epsilon ? 0 : 1.0 / S[2];
    T S_UT[9];
    S_UT[0] = U[0] * S[0];
    S_UT[1] = U[3] * S[0];
    S_UT[2] = U[6] * S[0];
    S_UT[3] = U[1] * S[1];
    S_UT[4] = U[4] * S[1];
    S_UT[5] = U[7] * S[1];
    S_UT[6] = U[2] * S[2];
    S_UT[7] = U[5] * S[2];
    S_UT[8] = U[8] * S[2];
    T Ainv[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Ainv[i * 3 + j] = V[i * 3 + 0] * S_UT[0 * 3 + j] +
                              V[i * 3 + 1] * S_UT[1 * 3 + j] +
                              V[i * 3 + 2] * S_UT[2 * 3 + j];
    X_3x1[0] = Ainv[0] * B_3x1[0] + Ainv[1] * B_3x1[1] + Ainv[2] * B_3x1[2];
    X_3x1[1] = Ainv[3] * B_3x1[0] + Ainv[4] * B_3x1[1] + Ainv[5] * B_3x1[2];
    X_3x1[2] = Ainv[6] * B_3x1[0] + Ainv[7] * B_3x1[1] + Ainv[8] * B_3x1[2];
}

}  // namespace svd3
}  // namespace o3dmi
