// The one eigenpair the reference uses of the Taubin pencil (quadric.cpp:143-153, 330-363), one sample per lane.
// Included by taubin.hip (k_taubin_eigen) and by scripts/micro/eigen_phases.hip (phase clocks of one wave).
#pragma once
#include <hip/hip_runtime.h>

#include "agh_internal.h"

#ifndef AGH_EIG_STAMP0  // phase boundaries; scripts/micro/eigen_phases.hip turns them into truncation points
#define AGH_EIG_TEMPLATE template <int LPS>
#define AGH_EIG_STAMP0
#define AGH_EIG_STAMP1
#define AGH_EIG_STAMP2
#define AGH_EIG_STAMP3
#define AGH_EIG_STAMP4
#define AGH_EIG_STAMP5
#define AGH_EIG_STAMP6
#endif

namespace agh
{

constexpr double kDeflateTol = 0x1p-40;  // relative Cholesky pivot at or below which a coordinate is deflated
constexpr double kPivMin = 0x1p-500;     // floor of |pivot| in the twisted factorisation
constexpr int kBisectSteps = 56;

__device__ __forceinline__ double pivot_floor(double x) { return (fabs(x) >= kPivMin) ? x : kPivMin; }

// sv = the 37 sums of k_taubin_moments, n = the neighbour count; v = the 10 quadric parameters (quadric.cpp:152, before
// the halving of 3..5), returns the eigenvalue.
// LPS = lanes per sample: 1, or 8 consecutive lanes that hold the SAME sample (every lane computes everything; only the
// bisection shares work).  The bisection is 56 dependent decisions, 60 % of the solve.  With eight lanes per sample the
// lanes 1..7 of a group evaluate the seven nodes of the next THREE levels of the decision tree at once -- lane q walks to
// node q (heap numbering) from the common bracket with the same fused multiply-adds the serial loop would execute on that
// path, so every midpoint is bit-identical to the serial one -- a ballot collects the seven verdicts, and every lane
// replays the three decisions: 19 passes instead of 56 steps.  Worth it while the samples x 8 still fit one wave per SIMD.
AGH_EIG_TEMPLATE __device__ __forceinline__ double taubin_smallest_eigenpair(const double (&sv)[kNumSums], double n, double (&v)[10])
{
  // S = M9 - b b^T / n, lower triangle (M's upper triangle by quadric.cpp:40-100; the 10th unknown eliminated)
  double M[10][10];
#pragma unroll
  for (int j = 0; j < 10; j++)
    M[0][j] = sv[j];
#pragma unroll
  for (int j = 1; j < 10; j++)
    M[1][j] = sv[10 + j - 1];
#pragma unroll
  for (int j = 2; j < 10; j++)
    M[2][j] = sv[19 + j - 2];
  M[3][8] = sv[27];
  M[3][9] = sv[28];
  M[4][9] = sv[29];
  M[5][9] = sv[30];
  M[6][9] = sv[31];
  M[7][9] = sv[32];
  M[8][9] = sv[33];
  M[3][3] = M[0][1];
  M[5][5] = M[0][2];
  M[3][5] = M[0][4];
  M[3][6] = M[0][7];
  M[5][6] = M[0][8];
  M[6][6] = M[0][9];
  M[4][4] = M[1][2];
  M[3][4] = M[1][5];
  M[3][7] = M[1][6];
  M[4][7] = M[1][8];
  M[7][7] = M[1][9];
  M[4][5] = M[2][3];
  M[5][8] = M[2][6];
  M[4][8] = M[2][7];
  M[8][8] = M[2][9];
  M[4][6] = M[3][8];
  M[5][7] = M[3][8];
  M[6][7] = M[3][9];
  M[7][8] = M[4][9];
  M[6][8] = M[5][9];
  // N (quadric.cpp:103-131), upper triangle: every entry except (3,3),(4,4),(5,5) is an exact power-of-two multiple of an
  // M sum; the rest is structurally zero
  double N[9][9];
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j < 9; j++)
      N[i][j] = 0.0;
  N[0][0] = 4.0 * sv[9];
  N[0][3] = 2.0 * sv[28];
  N[0][5] = 2.0 * sv[30];
  N[0][6] = 2.0 * sv[31];
  N[1][1] = 4.0 * sv[18];
  N[1][3] = 2.0 * sv[28];
  N[1][4] = 2.0 * sv[29];
  N[1][7] = 2.0 * sv[32];
  N[2][2] = 4.0 * sv[26];
  N[2][4] = 2.0 * sv[29];
  N[2][5] = 2.0 * sv[30];
  N[2][8] = 2.0 * sv[33];
  N[3][3] = sv[34];
  N[3][4] = sv[30];
  N[3][5] = sv[29];
  N[3][6] = sv[32];
  N[3][7] = sv[31];
  N[4][4] = sv[35];
  N[4][5] = sv[28];
  N[4][7] = sv[33];
  N[4][8] = sv[32];
  N[5][5] = sv[36];
  N[5][6] = sv[33];
  N[5][8] = sv[31];
  N[6][6] = n;
  N[7][7] = n;
  N[8][8] = n;
  double b[9], C[9][9], L[9][9], rinv[9];
  bool defl[9];
#pragma unroll
  for (int i = 0; i < 9; i++)
    b[i] = M[i][9];
#pragma unroll
  for (int i = 0; i < 9; i++)
  {
    const double ti = b[i] / n;
#pragma unroll
    for (int j = 0; j <= i; j++)
      C[i][j] = fma(-ti, b[j], M[j][i]);
  }
  AGH_EIG_STAMP0;
  // 1. Cholesky N9 = L L^T with deflation
#pragma unroll
  for (int j = 0; j < 9; j++)
  {
    double sp = N[j][j];
#pragma unroll
    for (int k = 0; k < j; k++)
      sp = fma(-L[j][k], L[j][k], sp);
    const bool ok = sp > kDeflateTol * N[j][j];
    const double ljj = sqrt(ok ? sp : 1.0);
    rinv[j] = ok ? 1.0 / ljj : 0.0;
    defl[j] = !ok;
#pragma unroll
    for (int k = 0; k < j; k++)
      L[j][k] = ok ? L[j][k] : 0.0;
#pragma unroll
    for (int i = j + 1; i < 9; i++)
    {
      double s2 = N[j][i];
#pragma unroll
      for (int k = 0; k < j; k++)
        s2 = fma(-L[i][k], L[j][k], s2);
      L[i][j] = s2 * rinv[j];
    }
  }
  AGH_EIG_STAMP1;
  // 2. C = L^-1 S L^-T in place on the lower triangle (the unblocked dsygs2 scheme)
#pragma unroll
  for (int k = 0; k < 9; k++)
  {
    const double akk = (C[k][k] * rinv[k]) * rinv[k];
    C[k][k] = akk;
    const double ct = -0.5 * akk;
#pragma unroll
    for (int i = k + 1; i < 9; i++)
      C[i][k] = fma(ct, L[i][k], C[i][k] * rinv[k]);
#pragma unroll
    for (int i = k + 1; i < 9; i++)
#pragma unroll
      for (int j = k + 1; j <= i; j++)
        C[i][j] = fma(-L[i][k], C[j][k], fma(-C[i][k], L[j][k], C[i][j]));
#pragma unroll
    for (int i = k + 1; i < 9; i++)
      C[i][k] = fma(ct, L[i][k], C[i][k]);
#pragma unroll
    for (int i = k + 1; i < 9; i++)
    {
      double s2 = C[i][k];
#pragma unroll
      for (int m = k + 1; m < i; m++)
        s2 = fma(-L[i][m], C[m][k], s2);
      C[i][k] = s2 * rinv[i];
    }
  }
  AGH_EIG_STAMP2;
  double tr = 0.0;
#pragma unroll
  for (int i = 0; i < 9; i++)
    tr += fabs(C[i][i]);
  const double big = 2.0 * tr + 1.0;
#pragma unroll
  for (int j = 0; j < 9; j++)
    C[j][j] = defl[j] ? big : C[j][j];
  // 3. Householder tridiagonalisation on the lower triangle; reflector k stays in column k below the diagonal
  double RH[7], e[8];
#pragma unroll
  for (int k = 0; k < 7; k++)
  {
    double sg = 0.0;
#pragma unroll
    for (int i = k + 1; i < 9; i++)
      sg = fma(C[i][k], C[i][k], sg);
    const double x0 = C[k + 1][k];
    const double rt = sqrt(sg);
    const double g = (x0 >= 0.0) ? -rt : rt;
    const double h = fma(-x0, g, sg);
    const bool live = h > 0.0;
    const double rh = live ? 1.0 / h : 0.0;
    double u[9], p[9], q[9];
#pragma unroll
    for (int i = k + 1; i < 9; i++)
      u[i] = live ? C[i][k] : 0.0;
    u[k + 1] = live ? x0 - g : 0.0;
    RH[k] = rh;
    e[k] = live ? g : x0;
#pragma unroll
    for (int i = k + 1; i < 9; i++)
    {
      double s2 = 0.0;
#pragma unroll
      for (int j = k + 1; j < 9; j++)
        s2 = fma(j <= i ? C[i][j] : C[j][i], u[j], s2);
      p[i] = s2 * rh;
    }
    double kk = 0.0;
#pragma unroll
    for (int i = k + 1; i < 9; i++)
      kk = fma(u[i], p[i], kk);
    kk = (kk * rh) * 0.5;
#pragma unroll
    for (int i = k + 1; i < 9; i++)
      q[i] = fma(-kk, u[i], p[i]);
#pragma unroll
    for (int i = k + 1; i < 9; i++)
#pragma unroll
      for (int j = k + 1; j <= i; j++)
        C[i][j] = fma(-q[i], u[j], fma(-u[i], q[j], C[i][j]));
#pragma unroll
    for (int i = k + 1; i < 9; i++)
      C[i][k] = u[i];
  }
  e[7] = C[8][7];
  double d[9], e2[8];
#pragma unroll
  for (int i = 0; i < 9; i++)
    d[i] = C[i][i];
#pragma unroll
  for (int i = 0; i < 8; i++)
    e2[i] = e[i] * e[i];
  AGH_EIG_STAMP3;
  // 4. bisection of the Gershgorin bracket of the smallest eigenvalue (deflated coordinates excluded)
  double lo = 0.0, hi = 0.0;
  bool first = true;
#pragma unroll
  for (int i = 0; i < 9; i++)
  {
    const double r = ((i > 0) ? fabs(e[i > 0 ? i - 1 : 0]) : 0.0) + ((i < 8) ? fabs(e[i < 8 ? i : 7]) : 0.0);
    const double g0 = d[i] - r;
    const bool take = !defl[i];
    lo = (take && (first || g0 < lo)) ? g0 : lo;
    hi = (take && (first || d[i] < hi)) ? d[i] : hi;
    first = first && !take;
  }
  // an eigenvalue of T lies below x iff the Sturm sequence changes sign, i.e. iff one of its members is negative
  // (p_0 = 1): the OR of their sign bits, gathered off the dependent chain of fused multiply-adds
  auto below_at = [&](double x) -> bool {
    double pm2 = 1.0, pm1 = d[0] - x;
    int sgn = __double2hiint(pm1);
#pragma unroll
    for (int i = 1; i < 9; i++)
    {
      const double pi = fma(d[i] - x, pm1, -(e2[i - 1] * pm2));
      sgn |= __double2hiint(pi);
      pm2 = pm1;
      pm1 = pi;
    }
    return sgn < 0;
  };
  if (LPS == 1)
  {
#pragma unroll 1
    for (int it = 0; it < kBisectSteps; it++)
    {
      const double mid = fma(hi - lo, 0.5, lo);
      const bool below = below_at(mid);
      hi = below ? mid : hi;
      lo = below ? lo : mid;
    }
  }
  else
  {
    static_assert(LPS == 1 || LPS == 8, "one lane per sample, or eight");
    const int q = (int) (threadIdx.x & 7);  // my node of the three-level tree (1 = root, 2 q = below, 2 q + 1 = not below); lane 0 idles
    const int gshift = (int) (threadIdx.x & 63 & ~7);
    int left = kBisectSteps;
#pragma unroll 1
    while (left > 0)
    {
      const int levels = left >= 3 ? 3 : left;
      // walk from the common bracket to my node: the path is the bits of q below its leading one, most significant first
      double nlo = lo, nhi = hi;
      const int depth = q >= 4 ? 2 : (q >= 2 ? 1 : 0);
#pragma unroll
      for (int l = 0; l < 2; l++)
        if (l < depth)
        {
          const double m = fma(nhi - nlo, 0.5, nlo);
          const bool right = ((q >> (depth - 1 - l)) & 1) != 0;  // child 2 p + 1: the parent's verdict was "not below": lo = mid
          nlo = right ? m : nlo;
          nhi = right ? nhi : m;
        }
      const double mid = fma(nhi - nlo, 0.5, nlo);
      const bool below = below_at(mid);
      const unsigned verdicts = (unsigned) (__ballot(below) >> gshift) & 0xffu;  // bit p = verdict of node p of my sample
      // replay: every lane of the group takes the same `levels` decisions
      int node = 1;
#pragma unroll
      for (int l = 0; l < 3; l++)
        if (l < levels)
        {
          const double m = fma(hi - lo, 0.5, lo);
          const bool b = ((verdicts >> node) & 1u) != 0;
          hi = b ? m : hi;
          lo = b ? lo : m;
          node = 2 * node + (b ? 0 : 1);
        }
      left -= levels;
    }
  }
  const double sigma = lo;
  // 5. twisted factorisation of T - sigma I
  double Dp[9], Dm[9], lf[8], ub[8];
  Dp[0] = pivot_floor(d[0] - sigma);
#pragma unroll
  for (int i = 0; i < 8; i++)
  {
    lf[i] = e[i] / Dp[i];
    Dp[i + 1] = pivot_floor(fma(-lf[i], e[i], d[i + 1] - sigma));
  }
  Dm[8] = pivot_floor(d[8] - sigma);
#pragma unroll
  for (int i = 7; i >= 0; i--)
  {
    ub[i] = e[i] / Dm[i + 1];
    Dm[i] = pivot_floor(fma(-ub[i], e[i], d[i] - sigma));
  }
  int ks = -1;
  double gmin = 0.0;
#pragma unroll
  for (int k = 0; k < 9; k++)
  {
    const double gk = fabs((Dp[k] + Dm[k]) - (d[k] - sigma));
    const bool take = !defl[k] && (ks < 0 || gk < gmin);
    ks = take ? k : ks;
    gmin = take ? gk : gmin;
  }
  // z_ks = 1, z_i = -l_i z_i+1 below it, z_i+1 = -u_i z_i above it (selects instead of a data-dependent loop start)
  double z[9];
  {
    double zl[10];
    zl[9] = 0.0;
#pragma unroll
    for (int i = 8; i >= 0; i--)
      zl[i] = (i == ks) ? 1.0 : ((i < ks) ? -(lf[i < 8 ? i : 7] * zl[i + 1]) : 0.0);
    double zu = 0.0;
#pragma unroll
    for (int i = 0; i < 9; i++)
    {
      zu = (i == ks) ? 1.0 : ((i > ks) ? -(ub[i > 0 ? i - 1 : 0] * zu) : 0.0);
      z[i] = (i <= ks) ? zl[i] : zu;
    }
  }
  AGH_EIG_STAMP5;
  // 6. y = H_0 ... H_6 z, v9 = L^-T y, v10 = -(b . v9) / n
#pragma unroll
  for (int k = 6; k >= 0; k--)
  {
    double s2 = 0.0;
#pragma unroll
    for (int i = k + 1; i < 9; i++)
      s2 = fma(C[i][k], z[i], s2);
    s2 = s2 * RH[k];
#pragma unroll
    for (int i = k + 1; i < 9; i++)
      z[i] = fma(-s2, C[i][k], z[i]);
  }
#pragma unroll
  for (int i = 8; i >= 0; i--)
  {
    double s2 = z[i];
#pragma unroll
    for (int k = i + 1; k < 9; k++)
      s2 = fma(-L[k][i], v[k], s2);
    v[i] = s2 * rinv[i];
  }
  AGH_EIG_STAMP6;
  double bv = 0.0;
#pragma unroll
  for (int k = 0; k < 9; k++)
    bv = fma(b[k], v[k], bv);
  v[9] = -(bv / n);
  return sigma;
}


// The unit eigenvector of the smallest eigenvalue of the symmetric 3 x 3 matrix M3 = sum n n^T (quadric.cpp:266-280), one
// lane: one Householder reflection, bisection on the Sturm sequence, twisted factorisation (the oracle's
// smallest_eigvec3, operation for operation).  m = { M00, M01, M02, M11, M12, M22 }.
__device__ __forceinline__ void smallest_eigvec3(const double (&m)[6], double (&axis)[3])
{
  const double x0 = m[1], x1 = m[2];
  const double sg = fma(x1, x1, x0 * x0);
  const double rt = sqrt(sg);
  const double g = (x0 >= 0.0) ? -rt : rt;
  const double h = fma(-x0, g, sg);
  const bool live = h > 0.0;
  const double rh = live ? 1.0 / h : 0.0;
  const double u1 = live ? x0 - g : 0.0, u2 = live ? x1 : 0.0;
  const double p1 = fma(m[4], u2, m[3] * u1) * rh;
  const double p2 = fma(m[5], u2, m[4] * u1) * rh;
  const double kk = (fma(u2, p2, u1 * p1) * rh) * 0.5;
  const double q1 = fma(-kk, u1, p1), q2 = fma(-kk, u2, p2);
  const double d0 = m[0], d1 = fma(-q1, u1, fma(-u1, q1, m[3])), d2 = fma(-q2, u2, fma(-u2, q2, m[5]));
  const double e0 = live ? g : x0, e1 = fma(-q2, u1, fma(-u2, q1, m[4]));
  const double e20 = e0 * e0, e21 = e1 * e1;
  const double r0 = fabs(e0), r1 = fabs(e0) + fabs(e1), r2 = fabs(e1);
  double lo = d0 - r0, hi = d0;
  lo = (d1 - r1 < lo) ? d1 - r1 : lo;
  lo = (d2 - r2 < lo) ? d2 - r2 : lo;
  hi = (d1 < hi) ? d1 : hi;
  hi = (d2 < hi) ? d2 : hi;
#pragma unroll 1
  for (int it = 0; it < kBisectSteps; it++)
  {
    const double mid = fma(hi - lo, 0.5, lo);
    const double s1 = d0 - mid;
    const double s2 = fma(d1 - mid, s1, -e20);
    const double s3 = fma(d2 - mid, s2, -(e21 * s1));
    const bool below = ((__double2hiint(s1) | __double2hiint(s2)) | __double2hiint(s3)) < 0;
    hi = below ? mid : hi;
    lo = below ? lo : mid;
  }
  const double sigma = lo;
  const double Dp0 = pivot_floor(d0 - sigma);
  const double lf0 = e0 / Dp0;
  const double Dp1 = pivot_floor(fma(-lf0, e0, d1 - sigma));
  const double lf1 = e1 / Dp1;
  const double Dp2 = pivot_floor(fma(-lf1, e1, d2 - sigma));
  const double Dm2 = pivot_floor(d2 - sigma);
  const double ub1 = e1 / Dm2;
  const double Dm1 = pivot_floor(fma(-ub1, e1, d1 - sigma));
  const double ub0 = e0 / Dm1;
  const double Dm0 = pivot_floor(fma(-ub0, e0, d0 - sigma));
  const double g0 = fabs((Dp0 + Dm0) - (d0 - sigma)), g1 = fabs((Dp1 + Dm1) - (d1 - sigma)), g2 = fabs((Dp2 + Dm2) - (d2 - sigma));
  int ks = 0;
  double gmin = g0;
  ks = (g1 < gmin) ? 1 : ks;
  gmin = (g1 < gmin) ? g1 : gmin;
  ks = (g2 < gmin) ? 2 : ks;
  // z_ks = 1; below it z_i = -lf_i z_i+1, above it z_i+1 = -ub_i z_i
  double z0, z1, z2;
  if (ks == 0)
  {
    z0 = 1.0;
    z1 = -(ub0 * z0);
    z2 = -(ub1 * z1);
  }
  else if (ks == 1)
  {
    z1 = 1.0;
    z0 = -(lf0 * z1);
    z2 = -(ub1 * z1);
  }
  else
  {
    z2 = 1.0;
    z1 = -(lf1 * z2);
    z0 = -(lf0 * z1);
  }
  const double sdot = fma(u2, z2, u1 * z1) * rh;
  const double y0 = z0, y1 = fma(-sdot, u1, z1), y2 = fma(-sdot, u2, z2);
  const double nn = sqrt(fma(y2, y2, fma(y1, y1, y0 * y0)));
  axis[0] = y0 / nn;
  axis[1] = y1 / nn;
  axis[2] = y2 / nn;
}

}  // namespace agh
