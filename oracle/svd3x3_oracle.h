// TEST INFRASTRUCTURE ONLY.
//
// Restatement of the reference's approximate 3x3 SVD and the pseudo-inverse
// solve built on it (cpp/open3d/core/linalg/kernel/SVD3x3.h: svd3x3 :86-1129 /
// :1131-2168 for double / float -- the two specialisations are the same text --
// and solve_svd3x3 :2170-2215), the algorithm of McAdams, Selle, Tamstorf,
// Teran, Sifakis, "Computing the Singular Value Decomposition of 3x3 matrices
// with minimal branching and elementary floating point operations" (2011):
//
//   S = A^T A;  4 sweeps of Jacobi conjugation on S in the fixed pair order
//   (1,2), (2,3), (3,1), the rotation accumulated as a quaternion, each angle
//   from the unnormalised (ch, sh) = (s_pp - s_qq, s_pq / 2) with a clamp to
//   pi/8 when ch^2 <= (3 + 2 sqrt 2) sh^2;  V = R(q / |q|);  B = A V;  columns
//   of B and V sorted by decreasing norm with a sign flip on every swap;
//   B = U Sigma by three Givens rotations (QR), U accumulated from I.
//
// What has to be kept to reproduce the reference bit for bit is not the
// algorithm but how it is written:
//   * every intermediate lives in a union {scalar f; unsigned int ui;} and the
//     conditionals are 32-bit masks applied to .ui. For float that is a plain
//     select; for double .ui aliases only the LOW half of the mantissa, so a
//     "select" keeps the high half of whatever the register held before, an
//     XOR-swap exchanges low halves only, and (-2.0).ui & mask is always -2.0
//     (its low half is zero). The Float64 path of the reference is therefore
//     not an SVD of its input in general -- it is reproduced as it is;
//   * the registers are reused in a fixed order; since stale contents leak
//     through the masks (previous point), each statement below writes the
//     same register the reference writes, in the same order;
//   * 1 / sqrt(x) is (scalar)(1.0 / sqrt(x)) (float sqrt, float64 division)
//     followed, for the quaternion and the Givens angles, by one Newton step;
//   * literals: 1e-20 and 5.828... are double (the products / comparisons are
//     formed in float64 and narrowed), 1e-12 is narrowed to the scalar first.
// The three Jacobi conjugations and the three Givens steps are one function
// each here, called with the registers in the roles the unrolled reference
// gives them.

#pragma once

#include <cmath>
#include <cstring>

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
inline T Rsqrt(T x) {
    return (T)(1.0 / std::sqrt(x));
}
template <typename T>
inline T Max(T a, T b) {  // std::max
    return (a < b) ? b : a;
}

// One Jacobi conjugation. (a, b, c) = (s_pp, s_qp, s_qq) of the rotated pair,
// d = the third diagonal entry, (e, g) = the two off-diagonal entries that mix.
// kAxis: which quaternion component takes +sh (1: z, 2: x, 3: y).
template <typename T, int kAxis>
inline void JacobiConjugation(Reg<T>& a, Reg<T>& b, Reg<T>& c, Reg<T>& d,
                              Reg<T>& e, Reg<T>& g, Reg<T>& Sqvs,
                              Reg<T>& Sqvvx, Reg<T>& Sqvvy, Reg<T>& Sqvvz,
                              Reg<T>& Sc, Reg<T>& Ss, Reg<T>& Sch, Reg<T>& Ssh,
                              Reg<T>& Stmp1, Reg<T>& Stmp2, Reg<T>& Stmp3,
                              Reg<T>& Stmp4, Reg<T>& Stmp5) {
    Ssh.f = b.f * 0.5f;
    Stmp5.f = a.f - c.f;

    Stmp2.f = Ssh.f * Ssh.f;
    Stmp1.ui = (Stmp2.f >= kTiny) ? 0xffffffff : 0;
    Ssh.ui = Stmp1.ui & Ssh.ui;
    Sch.ui = Stmp1.ui & Stmp5.ui;
    Stmp2.ui = ~Stmp1.ui & kOne;
    Sch.ui = Sch.ui | Stmp2.ui;

    Stmp1.f = Ssh.f * Ssh.f;
    Stmp2.f = Sch.f * Sch.f;
    Stmp3.f = Stmp1.f + Stmp2.f;
    Stmp4.f = Rsqrt(Stmp3.f);

    Ssh.f = Stmp4.f * Ssh.f;
    Sch.f = Stmp4.f * Sch.f;
    Stmp1.f = kFourGammaSquared * Stmp1.f;
    Stmp1.ui = (Stmp2.f <= Stmp1.f) ? 0xffffffff : 0;

    Stmp2.ui = kSinePiOver8 & Stmp1.ui;
    Ssh.ui = ~Stmp1.ui & Ssh.ui;
    Ssh.ui = Ssh.ui | Stmp2.ui;
    Stmp2.ui = kCosinePiOver8 & Stmp1.ui;
    Sch.ui = ~Stmp1.ui & Sch.ui;
    Sch.ui = Sch.ui | Stmp2.ui;

    Stmp1.f = Ssh.f * Ssh.f;
    Stmp2.f = Sch.f * Sch.f;
    Sc.f = Stmp2.f - Stmp1.f;
    Ss.f = Sch.f * Ssh.f;
    Ss.f = Ss.f + Ss.f;

    // the conjugation itself
    Stmp3.f = Stmp1.f + Stmp2.f;
    d.f = d.f * Stmp3.f;
    e.f = e.f * Stmp3.f;
    g.f = g.f * Stmp3.f;
    d.f = d.f * Stmp3.f;

    Stmp1.f = Ss.f * e.f;
    Stmp2.f = Ss.f * g.f;
    e.f = Sc.f * e.f;
    g.f = Sc.f * g.f;
    e.f = Stmp2.f + e.f;
    g.f = g.f - Stmp1.f;

    Stmp2.f = Ss.f * Ss.f;
    Stmp1.f = c.f * Stmp2.f;
    Stmp3.f = a.f * Stmp2.f;
    Stmp4.f = Sc.f * Sc.f;
    a.f = a.f * Stmp4.f;
    c.f = c.f * Stmp4.f;
    a.f = a.f + Stmp1.f;
    c.f = c.f + Stmp3.f;
    Stmp4.f = Stmp4.f - Stmp2.f;
    Stmp2.f = b.f + b.f;
    b.f = b.f * Stmp4.f;
    Stmp4.f = Sc.f * Ss.f;
    Stmp2.f = Stmp2.f * Stmp4.f;
    Stmp5.f = Stmp5.f * Stmp4.f;
    a.f = a.f + Stmp2.f;
    b.f = b.f - Stmp5.f;
    c.f = c.f - Stmp2.f;

    // cumulative rotation, as a quaternion
    Stmp1.f = Ssh.f * Sqvvx.f;
    Stmp2.f = Ssh.f * Sqvvy.f;
    Stmp3.f = Ssh.f * Sqvvz.f;
    Ssh.f = Ssh.f * Sqvs.f;

    Sqvs.f = Sch.f * Sqvs.f;
    Sqvvx.f = Sch.f * Sqvvx.f;
    Sqvvy.f = Sch.f * Sqvvy.f;
    Sqvvz.f = Sch.f * Sqvvz.f;

    if (kAxis == 1) {
        Sqvvz.f = Sqvvz.f + Ssh.f;
        Sqvs.f = Sqvs.f - Stmp3.f;
        Sqvvx.f = Sqvvx.f + Stmp2.f;
        Sqvvy.f = Sqvvy.f - Stmp1.f;
    } else if (kAxis == 2) {
        Sqvvx.f = Sqvvx.f + Ssh.f;
        Sqvs.f = Sqvs.f - Stmp1.f;
        Sqvvy.f = Sqvvy.f + Stmp3.f;
        Sqvvz.f = Sqvvz.f - Stmp2.f;
    } else {
        Sqvvy.f = Sqvvy.f + Ssh.f;
        Sqvs.f = Sqvs.f - Stmp2.f;
        Sqvvz.f = Sqvvz.f + Stmp1.f;
        Sqvvx.f = Sqvvx.f - Stmp3.f;
    }
}

// Row r of A times V: (x, y, z) <- (x, y, z) V, through Stmp1..3.
template <typename T>
inline void RowTimesV(Reg<T>& x, Reg<T>& y, Reg<T>& z, const Reg<T>& Sv11,
                      const Reg<T>& Sv21, const Reg<T>& Sv31,
                      const Reg<T>& Sv12, const Reg<T>& Sv22,
                      const Reg<T>& Sv32, const Reg<T>& Sv13,
                      const Reg<T>& Sv23, const Reg<T>& Sv33, Reg<T>& Stmp1,
                      Reg<T>& Stmp2, Reg<T>& Stmp3) {
    Stmp2.f = y.f;
    Stmp3.f = z.f;
    y.f = Sv12.f * x.f;
    z.f = Sv13.f * x.f;
    x.f = Sv11.f * x.f;
    Stmp1.f = Sv21.f * Stmp2.f;
    x.f = x.f + Stmp1.f;
    Stmp1.f = Sv31.f * Stmp3.f;
    x.f = x.f + Stmp1.f;
    Stmp1.f = Sv22.f * Stmp2.f;
    y.f = y.f + Stmp1.f;
    Stmp1.f = Sv32.f * Stmp3.f;
    y.f = y.f + Stmp1.f;
    Stmp1.f = Sv23.f * Stmp2.f;
    z.f = z.f + Stmp1.f;
    Stmp1.f = Sv33.f * Stmp3.f;
    z.f = z.f + Stmp1.f;
}

template <typename T>
inline void MaskedSwap(Reg<T>& x, Reg<T>& y, Reg<T>& Stmp5,
                       const Reg<T>& Stmp4) {
    Stmp5.ui = x.ui ^ y.ui;
    Stmp5.ui = Stmp5.ui & Stmp4.ui;
    x.ui = x.ui ^ Stmp5.ui;
    y.ui = y.ui ^ Stmp5.ui;
}

// Conditional swap of columns i and j of B and V (and of their squared norms
// ni, nj) when ni < nj, then the sign flip of column `neg` (n1..n3 = the
// column the reference multiplies).
template <typename T>
inline void SortColumns(Reg<T>& ai1, Reg<T>& aj1, Reg<T>& ai2, Reg<T>& aj2,
                        Reg<T>& ai3, Reg<T>& aj3, Reg<T>& vi1, Reg<T>& vj1,
                        Reg<T>& vi2, Reg<T>& vj2, Reg<T>& vi3, Reg<T>& vj3,
                        Reg<T>& ni, Reg<T>& nj, Reg<T>& n1, Reg<T>& n2,
                        Reg<T>& n3, Reg<T>& w1, Reg<T>& w2, Reg<T>& w3,
                        Reg<T>& Stmp4, Reg<T>& Stmp5) {
    Stmp4.ui = (ni.f < nj.f) ? 0xffffffff : 0;
    MaskedSwap(ai1, aj1, Stmp5, Stmp4);
    MaskedSwap(ai2, aj2, Stmp5, Stmp4);
    MaskedSwap(ai3, aj3, Stmp5, Stmp4);
    MaskedSwap(vi1, vj1, Stmp5, Stmp4);
    MaskedSwap(vi2, vj2, Stmp5, Stmp4);
    MaskedSwap(vi3, vj3, Stmp5, Stmp4);
    MaskedSwap(ni, nj, Stmp5, Stmp4);
    Stmp5.f = -2.f;
    Stmp5.ui = Stmp5.ui & Stmp4.ui;
    Stmp4.f = 1.f;
    Stmp4.f = Stmp4.f + Stmp5.f;
    n1.f = n1.f * Stmp4.f;
    n2.f = n2.f * Stmp4.f;
    n3.f = n3.f * Stmp4.f;
    w1.f = w1.f * Stmp4.f;
    w2.f = w2.f * Stmp4.f;
    w3.f = w3.f * Stmp4.f;
}

template <typename T>
inline void Rotate(Reg<T>& x, Reg<T>& y, const Reg<T>& Sc, const Reg<T>& Ss,
                   Reg<T>& Stmp1, Reg<T>& Stmp2) {
    Stmp1.f = Ss.f * x.f;
    Stmp2.f = Ss.f * y.f;
    x.f = Sc.f * x.f;
    y.f = Sc.f * y.f;
    x.f = x.f + Stmp2.f;
    y.f = y.f - Stmp1.f;
}

// One Givens step of the QR factorisation: zero `below` against `pivot`,
// rotating the three row pairs of B and the three column pairs of U.
template <typename T>
inline void GivensQR(const Reg<T>& pivot, const Reg<T>& below, Reg<T>& r1x,
                     Reg<T>& r1y, Reg<T>& r2x, Reg<T>& r2y, Reg<T>& r3x,
                     Reg<T>& r3y, Reg<T>& u1x, Reg<T>& u1y, Reg<T>& u2x,
                     Reg<T>& u2y, Reg<T>& u3x, Reg<T>& u3y, T gsmall_number,
                     Reg<T>& Sc, Reg<T>& Ss, Reg<T>& Sch, Reg<T>& Ssh,
                     Reg<T>& Stmp1, Reg<T>& Stmp2, Reg<T>& Stmp3,
                     Reg<T>& Stmp4, Reg<T>& Stmp5) {
    Ssh.f = below.f * below.f;
    Ssh.ui = (Ssh.f >= gsmall_number) ? 0xffffffff : 0;
    Ssh.ui = Ssh.ui & below.ui;

    Stmp5.f = 0.f;
    Sch.f = Stmp5.f - pivot.f;
    Sch.f = Max(Sch.f, pivot.f);
    Sch.f = Max(Sch.f, gsmall_number);
    Stmp5.ui = (pivot.f >= Stmp5.f) ? 0xffffffff : 0;

    Stmp1.f = Sch.f * Sch.f;
    Stmp2.f = Ssh.f * Ssh.f;
    Stmp2.f = Stmp1.f + Stmp2.f;
    Stmp1.f = Rsqrt(Stmp2.f);

    Stmp4.f = Stmp1.f * 0.5f;
    Stmp3.f = Stmp1.f * Stmp4.f;
    Stmp3.f = Stmp1.f * Stmp3.f;
    Stmp3.f = Stmp2.f * Stmp3.f;
    Stmp1.f = Stmp1.f + Stmp4.f;
    Stmp1.f = Stmp1.f - Stmp3.f;
    Stmp1.f = Stmp1.f * Stmp2.f;

    Sch.f = Sch.f + Stmp1.f;

    Stmp1.ui = ~Stmp5.ui & Ssh.ui;
    Stmp2.ui = ~Stmp5.ui & Sch.ui;
    Sch.ui = Stmp5.ui & Sch.ui;
    Ssh.ui = Stmp5.ui & Ssh.ui;
    Sch.ui = Sch.ui | Stmp1.ui;
    Ssh.ui = Ssh.ui | Stmp2.ui;

    Stmp1.f = Sch.f * Sch.f;
    Stmp2.f = Ssh.f * Ssh.f;
    Stmp2.f = Stmp1.f + Stmp2.f;
    Stmp1.f = Rsqrt(Stmp2.f);

    Stmp4.f = Stmp1.f * 0.5f;
    Stmp3.f = Stmp1.f * Stmp4.f;
    Stmp3.f = Stmp1.f * Stmp3.f;
    Stmp3.f = Stmp2.f * Stmp3.f;
    Stmp1.f = Stmp1.f + Stmp4.f;
    Stmp1.f = Stmp1.f - Stmp3.f;

    Sch.f = Sch.f * Stmp1.f;
    Ssh.f = Ssh.f * Stmp1.f;

    Sc.f = Sch.f * Sch.f;
    Ss.f = Ssh.f * Ssh.f;
    Sc.f = Sc.f - Ss.f;
    Ss.f = Ssh.f * Sch.f;
    Ss.f = Ss.f + Ss.f;

    Rotate(r1x, r1y, Sc, Ss, Stmp1, Stmp2);
    Rotate(r2x, r2y, Sc, Ss, Stmp1, Stmp2);
    Rotate(r3x, r3y, Sc, Ss, Stmp1, Stmp2);
    Rotate(u1x, u1y, Sc, Ss, Stmp1, Stmp2);
    Rotate(u2x, u2y, Sc, Ss, Stmp1, Stmp2);
    Rotate(u3x, u3y, Sc, Ss, Stmp1, Stmp2);
}

// svd3x3, SVD3x3.h:86-1129: A = U diag(S) V^T (row-major 3x3 arrays).
template <typename T>
inline void Svd3x3(const T* A_3x3, T* U_3x3, T* S_3x1, T* V_3x3) {
    T gsmall_number = 1.e-12;

    Reg<T> Sa11, Sa21, Sa31, Sa12, Sa22, Sa32, Sa13, Sa23, Sa33;
    Reg<T> Su11, Su21, Su31, Su12, Su22, Su32, Su13, Su23, Su33;
    Reg<T> Sv11, Sv21, Sv31, Sv12, Sv22, Sv32, Sv13, Sv23, Sv33;
    Reg<T> Sc, Ss, Sch, Ssh;
    Reg<T> Stmp1, Stmp2, Stmp3, Stmp4, Stmp5;
    Reg<T> Ss11, Ss21, Ss31, Ss22, Ss32, Ss33;
    Reg<T> Sqvs, Sqvvx, Sqvvy, Sqvvz;
    // The reference leaves these uninitialised; their first use through .ui
    // happens after a full .f write except for Sch, whose high half (Float64)
    // is taken from an indeterminate value in the very first conjugation when
    // the guard fires. Zero is what a fresh stack page gives and what this
    // restatement fixes it to.
    std::memset(&Sch, 0, sizeof(Sch));
    std::memset(&Ssh, 0, sizeof(Ssh));
    std::memset(&Sc, 0, sizeof(Sc));
    std::memset(&Ss, 0, sizeof(Ss));
    std::memset(&Stmp1, 0, sizeof(Stmp1));
    std::memset(&Stmp2, 0, sizeof(Stmp2));
    std::memset(&Stmp3, 0, sizeof(Stmp3));
    std::memset(&Stmp4, 0, sizeof(Stmp4));
    std::memset(&Stmp5, 0, sizeof(Stmp5));

    Sa11.f = A_3x3[0];
    Sa12.f = A_3x3[1];
    Sa13.f = A_3x3[2];
    Sa21.f = A_3x3[3];
    Sa22.f = A_3x3[4];
    Sa23.f = A_3x3[5];
    Sa31.f = A_3x3[6];
    Sa32.f = A_3x3[7];
    Sa33.f = A_3x3[8];

    // normal equations matrix S = A^T A (lower triangle)
    Ss11.f = Sa11.f * Sa11.f;
    Stmp1.f = Sa21.f * Sa21.f;
    Ss11.f = Stmp1.f + Ss11.f;
    Stmp1.f = Sa31.f * Sa31.f;
    Ss11.f = Stmp1.f + Ss11.f;

    Ss21.f = Sa12.f * Sa11.f;
    Stmp1.f = Sa22.f * Sa21.f;
    Ss21.f = Stmp1.f + Ss21.f;
    Stmp1.f = Sa32.f * Sa31.f;
    Ss21.f = Stmp1.f + Ss21.f;

    Ss31.f = Sa13.f * Sa11.f;
    Stmp1.f = Sa23.f * Sa21.f;
    Ss31.f = Stmp1.f + Ss31.f;
    Stmp1.f = Sa33.f * Sa31.f;
    Ss31.f = Stmp1.f + Ss31.f;

    Ss22.f = Sa12.f * Sa12.f;
    Stmp1.f = Sa22.f * Sa22.f;
    Ss22.f = Stmp1.f + Ss22.f;
    Stmp1.f = Sa32.f * Sa32.f;
    Ss22.f = Stmp1.f + Ss22.f;

    Ss32.f = Sa13.f * Sa12.f;
    Stmp1.f = Sa23.f * Sa22.f;
    Ss32.f = Stmp1.f + Ss32.f;
    Stmp1.f = Sa33.f * Sa32.f;
    Ss32.f = Stmp1.f + Ss32.f;

    Ss33.f = Sa13.f * Sa13.f;
    Stmp1.f = Sa23.f * Sa23.f;
    Ss33.f = Stmp1.f + Ss33.f;
    Stmp1.f = Sa33.f * Sa33.f;
    Ss33.f = Stmp1.f + Ss33.f;

    Sqvs.f = 1.f;
    Sqvvx.f = 0.f;
    Sqvvy.f = 0.f;
    Sqvvz.f = 0.f;

    // symmetric eigenproblem: 4 Jacobi sweeps, pairs (1,2), (2,3), (3,1)
    for (int i = 0; i < 4; i++) {
        JacobiConjugation<T, 1>(Ss11, Ss21, Ss22, Ss33, Ss31, Ss32, Sqvs,
                                Sqvvx, Sqvvy, Sqvvz, Sc, Ss, Sch, Ssh, Stmp1,
                                Stmp2, Stmp3, Stmp4, Stmp5);
        JacobiConjugation<T, 2>(Ss22, Ss32, Ss33, Ss11, Ss21, Ss31, Sqvs,
                                Sqvvx, Sqvvy, Sqvvz, Sc, Ss, Sch, Ssh, Stmp1,
                                Stmp2, Stmp3, Stmp4, Stmp5);
        JacobiConjugation<T, 3>(Ss33, Ss31, Ss11, Ss22, Ss32, Ss21, Sqvs,
                                Sqvvx, Sqvvy, Sqvvz, Sc, Ss, Sch, Ssh, Stmp1,
                                Stmp2, Stmp3, Stmp4, Stmp5);
    }

    // normalise the quaternion (rsqrt + one Newton step), V = R(q)
    Stmp2.f = Sqvs.f * Sqvs.f;
    Stmp1.f = Sqvvx.f * Sqvvx.f;
    Stmp2.f = Stmp1.f + Stmp2.f;
    Stmp1.f = Sqvvy.f * Sqvvy.f;
    Stmp2.f = Stmp1.f + Stmp2.f;
    Stmp1.f = Sqvvz.f * Sqvvz.f;
    Stmp2.f = Stmp1.f + Stmp2.f;

    Stmp1.f = Rsqrt(Stmp2.f);
    Stmp4.f = Stmp1.f * 0.5f;
    Stmp3.f = Stmp1.f * Stmp4.f;
    Stmp3.f = Stmp1.f * Stmp3.f;
    Stmp3.f = Stmp2.f * Stmp3.f;
    Stmp1.f = Stmp1.f + Stmp4.f;
    Stmp1.f = Stmp1.f - Stmp3.f;

    Sqvs.f = Sqvs.f * Stmp1.f;
    Sqvvx.f = Sqvvx.f * Stmp1.f;
    Sqvvy.f = Sqvvy.f * Stmp1.f;
    Sqvvz.f = Sqvvz.f * Stmp1.f;

    Stmp1.f = Sqvvx.f * Sqvvx.f;
    Stmp2.f = Sqvvy.f * Sqvvy.f;
    Stmp3.f = Sqvvz.f * Sqvvz.f;
    Sv11.f = Sqvs.f * Sqvs.f;
    Sv22.f = Sv11.f - Stmp1.f;
    Sv33.f = Sv22.f - Stmp2.f;
    Sv33.f = Sv33.f + Stmp3.f;
    Sv22.f = Sv22.f + Stmp2.f;
    Sv22.f = Sv22.f - Stmp3.f;
    Sv11.f = Sv11.f + Stmp1.f;
    Sv11.f = Sv11.f - Stmp2.f;
    Sv11.f = Sv11.f - Stmp3.f;
    Stmp1.f = Sqvvx.f + Sqvvx.f;
    Stmp2.f = Sqvvy.f + Sqvvy.f;
    Stmp3.f = Sqvvz.f + Sqvvz.f;
    Sv32.f = Sqvs.f * Stmp1.f;
    Sv13.f = Sqvs.f * Stmp2.f;
    Sv21.f = Sqvs.f * Stmp3.f;
    Stmp1.f = Sqvvy.f * Stmp1.f;
    Stmp2.f = Sqvvz.f * Stmp2.f;
    Stmp3.f = Sqvvx.f * Stmp3.f;
    Sv12.f = Stmp1.f - Sv21.f;
    Sv23.f = Stmp2.f - Sv32.f;
    Sv31.f = Stmp3.f - Sv13.f;
    Sv21.f = Stmp1.f + Sv21.f;
    Sv32.f = Stmp2.f + Sv32.f;
    Sv13.f = Stmp3.f + Sv13.f;

    // B = A V
    RowTimesV(Sa11, Sa12, Sa13, Sv11, Sv21, Sv31, Sv12, Sv22, Sv32, Sv13, Sv23,
              Sv33, Stmp1, Stmp2, Stmp3);
    RowTimesV(Sa21, Sa22, Sa23, Sv11, Sv21, Sv31, Sv12, Sv22, Sv32, Sv13, Sv23,
              Sv33, Stmp1, Stmp2, Stmp3);
    RowTimesV(Sa31, Sa32, Sa33, Sv11, Sv21, Sv31, Sv12, Sv22, Sv32, Sv13, Sv23,
              Sv33, Stmp1, Stmp2, Stmp3);

    // squared column norms, then the three conditional swaps
    Stmp1.f = Sa11.f * Sa11.f;
    Stmp4.f = Sa21.f * Sa21.f;
    Stmp1.f = Stmp1.f + Stmp4.f;
    Stmp4.f = Sa31.f * Sa31.f;
    Stmp1.f = Stmp1.f + Stmp4.f;

    Stmp2.f = Sa12.f * Sa12.f;
    Stmp4.f = Sa22.f * Sa22.f;
    Stmp2.f = Stmp2.f + Stmp4.f;
    Stmp4.f = Sa32.f * Sa32.f;
    Stmp2.f = Stmp2.f + Stmp4.f;

    Stmp3.f = Sa13.f * Sa13.f;
    Stmp4.f = Sa23.f * Sa23.f;
    Stmp3.f = Stmp3.f + Stmp4.f;
    Stmp4.f = Sa33.f * Sa33.f;
    Stmp3.f = Stmp3.f + Stmp4.f;

    // (1,2): flips column 2; (1,3): flips column 1; (2,3): flips column 3
    SortColumns(Sa11, Sa12, Sa21, Sa22, Sa31, Sa32, Sv11, Sv12, Sv21, Sv22,
                Sv31, Sv32, Stmp1, Stmp2, Sa12, Sa22, Sa32, Sv12, Sv22, Sv32,
                Stmp4, Stmp5);
    SortColumns(Sa11, Sa13, Sa21, Sa23, Sa31, Sa33, Sv11, Sv13, Sv21, Sv23,
                Sv31, Sv33, Stmp1, Stmp3, Sa11, Sa21, Sa31, Sv11, Sv21, Sv31,
                Stmp4, Stmp5);
    SortColumns(Sa12, Sa13, Sa22, Sa23, Sa32, Sa33, Sv12, Sv13, Sv22, Sv23,
                Sv32, Sv33, Stmp2, Stmp3, Sa13, Sa23, Sa33, Sv13, Sv23, Sv33,
                Stmp4, Stmp5);

    // QR: B = U Sigma
    Su11.f = 1.f;
    Su12.f = 0.f;
    Su13.f = 0.f;
    Su21.f = 0.f;
    Su22.f = 1.f;
    Su23.f = 0.f;
    Su31.f = 0.f;
    Su32.f = 0.f;
    Su33.f = 1.f;

    GivensQR(Sa11, Sa21, Sa11, Sa21, Sa12, Sa22, Sa13, Sa23, Su11, Su12, Su21,
             Su22, Su31, Su32, gsmall_number, Sc, Ss, Sch, Ssh, Stmp1, Stmp2,
             Stmp3, Stmp4, Stmp5);
    GivensQR(Sa11, Sa31, Sa11, Sa31, Sa12, Sa32, Sa13, Sa33, Su11, Su13, Su21,
             Su23, Su31, Su33, gsmall_number, Sc, Ss, Sch, Ssh, Stmp1, Stmp2,
             Stmp3, Stmp4, Stmp5);
    GivensQR(Sa22, Sa32, Sa21, Sa31, Sa22, Sa32, Sa23, Sa33, Su12, Su13, Su22,
             Su23, Su32, Su33, gsmall_number, Sc, Ss, Sch, Ssh, Stmp1, Stmp2,
             Stmp3, Stmp4, Stmp5);

    V_3x3[0] = Sv11.f;
    V_3x3[1] = Sv12.f;
    V_3x3[2] = Sv13.f;
    V_3x3[3] = Sv21.f;
    V_3x3[4] = Sv22.f;
    V_3x3[5] = Sv23.f;
    V_3x3[6] = Sv31.f;
    V_3x3[7] = Sv32.f;
    V_3x3[8] = Sv33.f;

    U_3x3[0] = Su11.f;
    U_3x3[1] = Su12.f;
    U_3x3[2] = Su13.f;
    U_3x3[3] = Su21.f;
    U_3x3[4] = Su22.f;
    U_3x3[5] = Su23.f;
    U_3x3[6] = Su31.f;
    U_3x3[7] = Su32.f;
    U_3x3[8] = Su33.f;

    S_3x1[0] = Sa11.f;
    S_3x1[1] = Sa22.f;
    S_3x1[2] = Sa33.f;
}

// solve_svd3x3, SVD3x3.h:2170-2215: x = V Sigma^+ U^T b with singular values
// below 1e-10 (narrowed to the scalar) dropped.
template <typename T>
inline void SolveSvd3x3(const T* A_3x3, const T* B_3x1, T* X_3x1) {
    T U[9], V[9], S[3];
    Svd3x3<T>(A_3x3, U, S, V);
    const T epsilon = 1e-10;
    S[0] = std::abs(S[0]) < epsilon ? 0 : 1.0 / S[0];
    S[1] = std::abs(S[1]) < epsilon ? 0 : 1.0 / S[1];
    S[2] = std::abs(S[2]) < epsilon ? 0 : 1.0 / S[2];
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
