/*
 * agile_oracle.cpp -- CPU ORACLE for the agile_grasp hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see agile_oracle.h).  Dependency-free C++17 restatement of the reference's
 * OpenMP CPU path, operation for operation, in the reference's precision (float32 points, float64
 * arithmetic, float32 HOG/SVM).  Every function cites the reference file:line it follows
 * (paths relative to /root/reference; "x.cpp" = src/agile_grasp/x.cpp, "x.h" = include/agile_grasp/x.h).
 *
 * PARITY UNPINNED at the third-party seams (the libraries are not in the reference tree and not in this
 * image): FLANN radius search, LAPACK dggev, Eigen::EigenSolver, OpenCV 2.4 HOGDescriptor / CvSVM (predict, and --
 * training side -- train / save).  The interpretation used at each seam is stated where it is implemented;
 * tests/golden pins the eigen seam against scipy's LAPACK dggev, and the model writer against the reference's
 * shipped SVM file (byte for byte).
 *
 * Floating-point contract: compiled with -ffp-contract=off, no -ffast-math; every sum is written in the
 * order it is evaluated (left to right unless stated), so the HIP kernels can reproduce it bit for bit.
 */
#include "agile_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <unordered_map>
#include <set>
#include <array>
#include <vector>

namespace
{

struct V3
{
  double x, y, z;
};

inline double dot3(const double* a, const double* b)
{
  return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}

inline void cross3(const double* a, const double* b, double* o)
{
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

/* ---------------------------------------------------------------------------------------------------
 * a2  pcl::KdTreeFLANN<PointXYZRGBA>::radiusSearch (call sites hand_search.cpp:85,147) -- THIRD PARTY.
 * Interpretation (FLANN 1.8 L2_Simple<float>, RadiusResultSet, sorted=true):
 *   d2 = ((dx*dx) + dy*dy) + dz*dz in float32 with dx = q.x - p.x; keep iff d2 < (float)(r*r) (strict);
 *   results ascending by (d2, index) -- FLANN's DistanceIndex::operator<.  The query point is included.
 * An exact uniform grid stands in for the kd-tree (same result set, no approximation).
 * ------------------------------------------------------------------------------------------------- */
struct Neighbor
{
  float d2;
  int32_t idx;
};

inline bool nb_less(const Neighbor& a, const Neighbor& b)
{
  return (a.d2 < b.d2) || ((a.d2 == b.d2) && a.idx < b.idx);
}

struct Cloud
{
  const float* xyz;
  int64_t stride;
  const int32_t* cam;
  int64_t n;
  inline const float* pt(int64_t i) const { return xyz + i * stride; }
};

static inline bool finite_pt(const float* p) { return std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]); }

struct GridIndex
{
  double cell = 0;
  double mn[3] = { 0, 0, 0 };
  int64_t dim[3] = { 1, 1, 1 };
  std::vector<int64_t> start; /* CSR over cells */
  std::vector<int32_t> items;

  inline void coords(const float* p, int64_t c[3]) const
  {
    for (int a = 0; a < 3; a++)
    {
      int64_t v = (int64_t) std::floor(((double) p[a] - mn[a]) / cell);
      c[a] = std::min(std::max(v, (int64_t) 0), dim[a] - 1);
    }
  }

  void build(const Cloud& cl, double cell_size)
  {
    cell = cell_size;
    double mx[3] = { 0, 0, 0 };
    for (int a = 0; a < 3; a++)
    {
      mn[a] = 1e300;
      mx[a] = -1e300;
    }
    /* A point with a non-finite coordinate is not part of the search structure: pcl::KdTreeFLANN::setInputCloud
     * (hand_search.cpp:10-11; PCL 1.7 kdtree_flann.hpp, convertCloudToArray) skips every point its point representation calls
     * invalid, so radiusSearch can never return it.  It keeps its index (the other points' indices do not move). */
    int64_t n_finite = 0;
    for (int64_t i = 0; i < cl.n; i++)
    {
      if (!finite_pt(cl.pt(i)))
        continue;
      n_finite++;
      for (int a = 0; a < 3; a++)
      {
        mn[a] = std::min(mn[a], (double) cl.pt(i)[a]);
        mx[a] = std::max(mx[a], (double) cl.pt(i)[a]);
      }
    }
    if (n_finite == 0)
      for (int a = 0; a < 3; a++)
        mn[a] = mx[a] = 0;
    for (int a = 0; a < 3; a++)
      dim[a] = (int64_t) std::floor((mx[a] - mn[a]) / cell) + 1;
    /* keep the table bounded for sparse far-flung clouds */
    while ((double) dim[0] * (double) dim[1] * (double) dim[2] > 6.4e7)
    {
      cell *= 2.0;
      for (int a = 0; a < 3; a++)
        dim[a] = (int64_t) std::floor((mx[a] - mn[a]) / cell) + 1;
    }
    int64_t ncell = dim[0] * dim[1] * dim[2];
    start.assign(ncell + 1, 0);
    std::vector<int64_t> key(cl.n);
    for (int64_t i = 0; i < cl.n; i++)
    {
      if (!finite_pt(cl.pt(i)))
      {
        key[i] = -1;
        continue;
      }
      int64_t c[3];
      coords(cl.pt(i), c);
      key[i] = (c[2] * dim[1] + c[1]) * dim[0] + c[0];
      start[key[i] + 1]++;
    }
    for (int64_t k = 0; k < ncell; k++)
      start[k + 1] += start[k];
    items.resize(n_finite);
    std::vector<int64_t> fill(start.begin(), start.end() - 1);
    for (int64_t i = 0; i < cl.n; i++)
      if (key[i] >= 0)
        items[fill[key[i]]++] = (int32_t) i;
  }

  void query(const Cloud& cl, const float* q, double radius, std::vector<Neighbor>& out) const
  {
    out.clear();
    /* a non-finite query (PCL asserts on one; with assertions compiled out FLANN finds nothing): no neighbours */
    if (!finite_pt(q))
      return;
    const float r2 = static_cast<float>(radius * radius);
    int64_t lo[3], hi[3];
    for (int a = 0; a < 3; a++)
    {
      int64_t l = (int64_t) std::floor(((double) q[a] - radius - mn[a]) / cell) - 1;
      int64_t h = (int64_t) std::floor(((double) q[a] + radius - mn[a]) / cell) + 1;
      lo[a] = std::min(std::max(l, (int64_t) 0), dim[a] - 1);
      hi[a] = std::min(std::max(h, (int64_t) 0), dim[a] - 1);
    }
    for (int64_t cz = lo[2]; cz <= hi[2]; cz++)
      for (int64_t cy = lo[1]; cy <= hi[1]; cy++)
      {
        int64_t base = (cz * dim[1] + cy) * dim[0];
        for (int64_t k = start[base + lo[0]]; k < start[base + hi[0] + 1]; k++)
        {
          const int32_t i = items[k];
          const float* p = cl.pt(i);
          const float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
          float d2 = 0.0f;
          d2 += dx * dx;
          d2 += dy * dy;
          d2 += dz * dz;
          if (d2 < r2)
            out.push_back(Neighbor{ d2, i });
        }
      }
    std::sort(out.begin(), out.end(), nb_less);
  }
};

/* ---------------------------------------------------------------------------------------------------
 * glibc rand() (TYPE_3 additive feedback generator), the generator behind quadric.cpp:184-187.
 * ------------------------------------------------------------------------------------------------- */
struct GlibcRand
{
  std::vector<uint32_t> r;
  size_t pos = 0;
  explicit GlibcRand(uint32_t seed)
  {
    r.resize(344);
    int32_t s = (int32_t) (seed == 0 ? 1u : seed);
    r[0] = (uint32_t) s;
    for (int i = 1; i < 31; i++)
    {
      int64_t hi = (int32_t) r[i - 1] / 127773;
      int64_t lo = (int32_t) r[i - 1] % 127773;
      int64_t word = 16807 * lo - 2836 * hi;
      if (word < 0)
        word += 2147483647;
      r[i] = (uint32_t) word;
    }
    for (int i = 31; i < 34; i++)
      r[i] = r[i - 31];
    for (int i = 34; i < 344; i++)
      r[i] = r[i - 31] + r[i - 3];
    pos = 344;
  }
  int32_t next()
  {
    r.push_back(r[pos - 31] + r[pos - 3]);
    uint32_t o = r[pos] >> 1;
    pos++;
    return (int32_t) o;
  }
};

/* ---------------------------------------------------------------------------------------------------
 * The summation order of the two sums whose order the reference leaves to Eigen (see fit_frame): 64 interleaved
 * partial sums, term i into partial i mod 64 in index order, then a balanced binary tree
 * (P[l] += P[l + o] for l < o, o = 32, 16, 8, 4, 2, 1).  On the GPU this is one wavefront: lane l owns partial l and
 * the tree is a butterfly of cross-lane adds.
 * ------------------------------------------------------------------------------------------------- */
struct LaneSum64
{
  double p[64];
  LaneSum64()
  {
    for (int l = 0; l < 64; l++)
      p[l] = 0.0;
  }
  void add(int i, double v) { p[i & 63] += v; }
  double total()
  {
    for (int o = 32; o > 0; o >>= 1)
      for (int l = 0; l < o; l++)
        p[l] = p[l] + p[l + o];
    return p[0];
  }
};

/* ---------------------------------------------------------------------------------------------------
 * Cyclic Jacobi for a small symmetric matrix (row-major n x n).  Stands in for LAPACK dggev's QZ
 * (quadric.cpp:330-363, after the reduction below) and for Eigen::EigenSolver on the symmetric 3x3
 * (quadric.cpp:268-270).  d[j] = eigenvalue, column j of V = eigenvector.
 * ------------------------------------------------------------------------------------------------- */
template <int NN>
void jacobi_sym(double A[NN][NN], double V[NN][NN], double d[NN])
{
  for (int i = 0; i < NN; i++)
    for (int j = 0; j < NN; j++)
      V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; sweep++)
  {
    double off = 0.0;
    for (int p = 0; p < NN - 1; p++)
      for (int q = p + 1; q < NN; q++)
        off += A[p][q] * A[p][q];
    if (off == 0.0)
      break;
    for (int p = 0; p < NN - 1; p++)
      for (int q = p + 1; q < NN; q++)
      {
        const double apq = A[p][q];
        if (apq == 0.0)
          continue;
        const double app = A[p][p], aqq = A[q][q];
        const double aabs = std::fabs(apq);
        if (sweep > 3 && (std::fabs(app) + aabs == std::fabs(app)) && (std::fabs(aqq) + aabs == std::fabs(aqq)))
        {
          A[p][q] = 0.0;
          A[q][p] = 0.0;
          continue;
        }
        const double theta = (aqq - app) / (2.0 * apq);
        double t = 1.0 / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        if (theta < 0.0)
          t = -t;
        const double c = 1.0 / std::sqrt(t * t + 1.0);
        const double s = t * c;
        A[p][p] = app - t * apq;
        A[q][q] = aqq + t * apq;
        A[p][q] = 0.0;
        A[q][p] = 0.0;
        for (int k = 0; k < NN; k++)
        {
          if (k == p || k == q)
            continue;
          const double akp = A[k][p], akq = A[k][q];
          const double np_ = c * akp - s * akq;
          const double nq_ = s * akp + c * akq;
          A[k][p] = np_;
          A[p][k] = np_;
          A[k][q] = nq_;
          A[q][k] = nq_;
        }
        for (int k = 0; k < NN; k++)
        {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < NN; i++)
    d[i] = A[i][i];
}

/* ---------------------------------------------------------------------------------------------------
 * Quadric::solveGeneralizedEigenProblem + the eigenvalue selection of fitQuadric
 * (quadric.cpp:143-153, 330-363) -- dggev is THIRD PARTY.
 *
 * What the reference asks of LAPACK: the eigenvector of M v = lambda N v that belongs to the smallest of "the first
 * nine" eigenvalues.  N's 10th row/column is identically zero, so the pencil has one infinite eigenvalue, which
 * LAPACK returns last (tests/golden); eliminating the 10th unknown (v10 = -b.v9/n) leaves the symmetric 9x9 problem
 * S v = lambda N9 v with S = A - b b^T/n positive semi-definite and N9 = sum of gradient outer products.
 *
 * Only ONE eigenpair is used (quadric.cpp:150-153), so only one is computed:
 *   1. Cholesky N9 = L L^T WITH DEFLATION (below);
 *   2. C = L^-1 S L^-T;
 *   3. Householder tridiagonalisation C = Q T Q^T;
 *   4. the smallest eigenvalue of T by bisection on the Sturm sequence of T - x I (polynomial form, a fixed
 *      number of halvings of the Gershgorin bracket; sigma = the lower end, so that T - sigma I stays positive);
 *   5. its eigenvector from the twisted factorisation of T - sigma I (Fernando; Parlett & Dhillon): forward and
 *      backward pivots, gamma_k = D+_k + D-_k - (d_k - sigma), z_k = 1 at the k with the smallest |gamma_k|;
 *   6. v9 = L^-T Q z, v10 = -(b . v9)/n.
 * Pinned against scipy's LAPACK dggev by tests/golden (taubin_dggev.npz, e2e_lapack.npz): direction within
 * 4e-6 rad on regular pencils, which is dggev's own distance from itself when one input bit flips.
 *
 * RANK-DEFICIENT PENCILS.  A neighbourhood that lies exactly in one plane n.p = c -- the normal case for a table top
 * or a box face voxelised on an axis-aligned 3 mm lattice (localization.cpp:247-355, launch/baxter_grasps.launch:4) --
 * makes BOTH S and N9 annihilate w = (n.p - c)^2: the pencil is singular.  dggev does not notice; it returns three
 * eigenvalues of size 1e-14 (the quadrics (n.p - c) * {1, x, y}-ish, which vanish on the plane) and an eigenvector
 * from their span plus a multiple of w, the reference takes it (quadric.cpp:146-153, no validity test), and its
 * gradients are +-n: a correct surface normal (measured: tests/golden/make_e2e_goldens.py).  Here: a Cholesky pivot
 * that is not above 2^-40 of its diagonal entry (measured pivots: <= 4e-15 in magnitude on exactly planar patches,
 * >= 7e-6 on everything else) DEFLATES its coordinate -- the row and column are dropped from both matrices, i.e.
 * v_j = 0; since w_j != 0 at the failing pivot, every vector is (one with v_j = 0) + alpha w, and w contributes
 * neither value nor gradient on the points, so the reduced problem has the reference's eigenvectors modulo w.  Its
 * smallest eigenvalue is the (triple) zero, the vector some (n.p - c) * (linear form), the normals +-n.
 * The in-plane direction of the curvature axis that follows (quadric.cpp:268-280: sum n n^T then has a double zero
 * eigenvalue) is decided by rounding noise in the reference and is decided by rounding noise here: it is defined by
 * this file's arithmetic and solver-dependent in the reference.
 * Implementation of the deflation: row j of L := 0 and rinv_j = 0 instead of 1/L_jj -- every later use of index j
 * is a product with one of them, so the coordinate drops out without a branch -- and C_jj = BIG keeps it out of the
 * spectrum's low end.
 * Returns false only for an empty neighbourhood (n = 0).
 * ------------------------------------------------------------------------------------------------- */
static const double kDeflateTol = 0x1p-40; /* relative Cholesky pivot at or below which a coordinate is deflated */
static const double kPivMin = 0x1p-500;    /* floor of |pivot| in the twisted factorisation */
static const int kBisectSteps = 56;

static inline double pivot_floor(double x)
{
  return (std::fabs(x) >= kPivMin) ? x : kPivMin;
}

/* Every multiply-add of the solver is a FUSED multiply-add (std::fma: one rounding; exact in glibc with or without the
 * hardware instruction, v_fma_f64 on the GPU): this is this file's own algorithm, not a restatement of reference
 * source, so the choice is free, and it halves the dependent chains the GPU kernel is made of. */
bool solve_taubin(const double M[10][10], const double N[10][10], double v[10], double* lambda)
{
  const double n = M[9][9];
  if (!(n > 0.0))
    return false;
  double b[9], C[9][9], L[9][9], rinv[9];
  bool defl[9];
  for (int i = 0; i < 9; i++)
    b[i] = M[i][9];
  for (int i = 0; i < 9; i++)
  {
    const double ti = b[i] / n;
    for (int j = 0; j <= i; j++)
      C[i][j] = std::fma(-ti, b[j], M[i][j]);
  }
  /* 1. Cholesky with deflation (lower triangle; the diagonal lives in rinv) */
  for (int i = 0; i < 9; i++)
    for (int j = 0; j < 9; j++)
      L[i][j] = 0.0;
  for (int j = 0; j < 9; j++)
  {
    double sp = N[j][j];
    for (int k = 0; k < j; k++)
      sp = std::fma(-L[j][k], L[j][k], sp);
    const bool ok = sp > kDeflateTol * N[j][j];
    rinv[j] = ok ? 1.0 / std::sqrt(sp) : 0.0;
    defl[j] = !ok;
    if (!ok)
      for (int k = 0; k < j; k++)
        L[j][k] = 0.0;
    for (int i = j + 1; i < 9; i++)
    {
      double s2 = N[i][j];
      for (int k = 0; k < j; k++)
        s2 = std::fma(-L[i][k], L[j][k], s2);
      L[i][j] = s2 * rinv[j];
    }
  }
  /* 2. C = L^-1 S L^-T in place on the lower triangle (the unblocked LAPACK dsygs2 scheme, itype 1, lower) */
  for (int k = 0; k < 9; k++)
  {
    const double akk = (C[k][k] * rinv[k]) * rinv[k];
    C[k][k] = akk;
    const double ct = -0.5 * akk;
    for (int i = k + 1; i < 9; i++)
      C[i][k] = std::fma(ct, L[i][k], C[i][k] * rinv[k]);
    for (int i = k + 1; i < 9; i++)
      for (int j = k + 1; j <= i; j++)
        C[i][j] = std::fma(-L[i][k], C[j][k], std::fma(-C[i][k], L[j][k], C[i][j]));
    for (int i = k + 1; i < 9; i++)
      C[i][k] = std::fma(ct, L[i][k], C[i][k]);
    for (int i = k + 1; i < 9; i++)
    {
      double s2 = C[i][k];
      for (int m = k + 1; m < i; m++)
        s2 = std::fma(-L[i][m], C[m][k], s2);
      C[i][k] = s2 * rinv[i];
    }
  }
  double tr = 0.0;
  for (int i = 0; i < 9; i++)
    tr += std::fabs(C[i][i]);
  const double big = 2.0 * tr + 1.0;
  for (int j = 0; j < 9; j++)
    if (defl[j])
      C[j][j] = big;
  /* 3. Householder tridiagonalisation on the lower triangle (reflector k zeroes column k below the sub-diagonal,
   *    H_k = I - u u^T / h, and stays in that column) */
  double RH[7], e[8], d[9], e2[8];
  for (int k = 0; k < 7; k++)
  {
    double sg = 0.0;
    for (int i = k + 1; i < 9; i++)
      sg = std::fma(C[i][k], C[i][k], sg);
    const double x0 = C[k + 1][k];
    const double rt = std::sqrt(sg);
    const double g = (x0 >= 0.0) ? -rt : rt;
    const double h = std::fma(-x0, g, sg);
    const bool live = h > 0.0;
    const double rh = live ? 1.0 / h : 0.0;
    double u[9], p[9], q[9];
    for (int i = k + 1; i < 9; i++)
      u[i] = live ? C[i][k] : 0.0;
    u[k + 1] = live ? x0 - g : 0.0;
    RH[k] = rh;
    e[k] = live ? g : x0;
    for (int i = k + 1; i < 9; i++)
    {
      double s2 = 0.0;
      for (int j = k + 1; j < 9; j++)
        s2 = std::fma((j <= i) ? C[i][j] : C[j][i], u[j], s2);
      p[i] = s2 * rh;
    }
    double kk = 0.0;
    for (int i = k + 1; i < 9; i++)
      kk = std::fma(u[i], p[i], kk);
    kk = (kk * rh) * 0.5;
    for (int i = k + 1; i < 9; i++)
      q[i] = std::fma(-kk, u[i], p[i]);
    for (int i = k + 1; i < 9; i++)
      for (int j = k + 1; j <= i; j++)
        C[i][j] = std::fma(-q[i], u[j], std::fma(-u[i], q[j], C[i][j]));
    for (int i = k + 1; i < 9; i++)
      C[i][k] = u[i];
  }
  e[7] = C[8][7];
  for (int i = 0; i < 9; i++)
    d[i] = C[i][i];
  for (int i = 0; i < 8; i++)
    e2[i] = e[i] * e[i];
  /* 4. bisection: lambda_min lies in [min_i (d_i - (|e_i-1| + |e_i|)), min_i d_i] (deflated coordinates excluded).
   *    An eigenvalue lies below x iff the Sturm sequence p_0 = 1, p_1 = d_0 - x, p_i+1 = (d_i - x) p_i - e_i-1^2 p_i-1
   *    changes sign, i.e. iff some p_i is negative (sign bit set: -0 counts, +0 does not). */
  double lo = 0.0, hi = 0.0;
  bool first = true;
  for (int i = 0; i < 9; i++)
  {
    if (defl[i])
      continue;
    const double r = ((i > 0) ? std::fabs(e[i - 1]) : 0.0) + ((i < 8) ? std::fabs(e[i]) : 0.0);
    const double g0 = d[i] - r;
    if (first || g0 < lo)
      lo = g0;
    if (first || d[i] < hi)
      hi = d[i];
    first = false;
  }
  for (int it = 0; it < kBisectSteps; it++)
  {
    const double mid = std::fma(hi - lo, 0.5, lo);
    double pm2 = 1.0, pm1 = d[0] - mid;
    bool below = std::signbit(pm1);
    for (int i = 1; i < 9; i++)
    {
      const double pi = std::fma(d[i] - mid, pm1, -(e2[i - 1] * pm2));
      below = below || std::signbit(pi);
      pm2 = pm1;
      pm1 = pi;
    }
    if (below)
      hi = mid;
    else
      lo = mid;
  }
  const double sigma = lo;
  /* 5. twisted factorisation of T - sigma I */
  double Dp[9], Dm[9], lf[8], ub[8];
  Dp[0] = pivot_floor(d[0] - sigma);
  for (int i = 0; i < 8; i++)
  {
    lf[i] = e[i] / Dp[i];
    Dp[i + 1] = pivot_floor(std::fma(-lf[i], e[i], d[i + 1] - sigma));
  }
  Dm[8] = pivot_floor(d[8] - sigma);
  for (int i = 7; i >= 0; i--)
  {
    ub[i] = e[i] / Dm[i + 1];
    Dm[i] = pivot_floor(std::fma(-ub[i], e[i], d[i] - sigma));
  }
  int ks = -1;
  double gmin = 0.0;
  for (int k = 0; k < 9; k++)
  {
    if (defl[k])
      continue;
    const double gk = std::fabs((Dp[k] + Dm[k]) - (d[k] - sigma));
    if (ks < 0 || gk < gmin)
    {
      ks = k;
      gmin = gk;
    }
  }
  double z[9];
  for (int i = 0; i < 9; i++)
    z[i] = 0.0;
  z[ks] = 1.0;
  for (int i = ks - 1; i >= 0; i--)
    z[i] = -(lf[i] * z[i + 1]);
  for (int i = ks; i < 8; i++)
    z[i + 1] = -(ub[i] * z[i]);
  /* 6. y = H_0 ... H_6 z, v9 = L^-T y, v10 */
  for (int k = 6; k >= 0; k--)
  {
    double s2 = 0.0;
    for (int i = k + 1; i < 9; i++)
      s2 = std::fma(C[i][k], z[i], s2);
    s2 = s2 * RH[k];
    for (int i = k + 1; i < 9; i++)
      z[i] = std::fma(-s2, C[i][k], z[i]);
  }
  for (int i = 8; i >= 0; i--)
  {
    double s2 = z[i];
    for (int k = i + 1; k < 9; k++)
      s2 = std::fma(-L[k][i], v[k], s2);
    v[i] = s2 * rinv[i];
  }
  double bv = 0.0;
  for (int k = 0; k < 9; k++)
    bv = std::fma(b[k], v[k], bv);
  v[9] = -(bv / n);
  *lambda = sigma;
  return true;
}

/* ---------------------------------------------------------------------------------------------------
 * a3  Quadric::fitQuadric (quadric.cpp:14-157).  Neighbours in radius-search order, un-centred absolute
 * coordinates cast float->double (29-31); M upper triangle (40-73) with the repeated entries copied
 * (76-100), N (103-131), n on the diagonals (134,137-139), symmetrise (136,141).
 * ------------------------------------------------------------------------------------------------- */
void accumulate_MN(const Cloud& cl, const std::vector<Neighbor>& nb, double M[10][10], double N[10][10])
{
  for (int i = 0; i < 10; i++)
    for (int j = 0; j < 10; j++)
      M[i][j] = N[i][j] = 0.0;
  for (size_t t = 0; t < nb.size(); t++)
  {
    const float* p = cl.pt(nb[t].idx);
    if (std::isnan(p[0]))
      continue;
    const double x = p[0], y = p[1], z = p[2];
    const double x2 = x * x, y2 = y * y, z2 = z * z;
    const double xy = x * y, yz = y * z, xz = x * z;
    M[0][0] += x2 * x2;
    M[0][1] += x2 * y2;
    M[0][2] += x2 * z2;
    M[0][3] += x2 * xy;
    M[0][4] += x2 * yz;
    M[0][5] += x2 * xz;
    M[0][6] += x2 * x;
    M[0][7] += x2 * y;
    M[0][8] += x2 * z;
    M[0][9] += x2;
    M[1][1] += y2 * y2;
    M[1][2] += y2 * z2;
    M[1][3] += y2 * xy;
    M[1][4] += y2 * yz;
    M[1][5] += y2 * xz;
    M[1][6] += y2 * x;
    M[1][7] += y2 * y;
    M[1][8] += y2 * z;
    M[1][9] += y2;
    M[2][2] += z2 * z2;
    M[2][3] += z2 * xy;
    M[2][4] += z2 * yz;
    M[2][5] += z2 * xz;
    M[2][6] += z2 * x;
    M[2][7] += z2 * y;
    M[2][8] += z2 * z;
    M[2][9] += z2;
    M[3][8] += x * yz;
    M[3][9] += xy;
    M[4][9] += yz;
    M[5][9] += xz;
    M[6][9] += x;
    M[7][9] += y;
    M[8][9] += z;

    N[0][0] += 4.0 * x2;
    N[0][3] += 2.0 * xy;
    N[0][5] += 2.0 * xz;
    N[0][6] += 2.0 * x;
    N[1][1] += 4.0 * y2;
    N[1][3] += 2.0 * xy;
    N[1][4] += 2.0 * yz;
    N[1][7] += 2.0 * y;
    N[2][2] += 4.0 * z2;
    N[2][4] += 2.0 * yz;
    N[2][5] += 2.0 * xz;
    N[2][8] += 2.0 * z;
    N[3][3] += x2 + y2;
    N[3][4] += xz;
    N[3][5] += yz;
    N[3][6] += y;
    N[3][7] += x;
    N[4][4] += y2 + z2;
    N[4][5] += xy;
    N[4][7] += z;
    N[4][8] += y;
    N[5][5] += x2 + z2;
    N[5][6] += z;
    N[5][8] += x;
  }
  /* repeating elements in M (quadric.cpp:76-100; assigning after the loop gives the same values) */
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
  const double n = (double) (int) nb.size();
  M[9][9] = n;
  N[6][6] = n;
  N[7][7] = n;
  N[8][8] = n;
  for (int i = 0; i < 10; i++)
    for (int j = i + 1; j < 10; j++)
    {
      M[j][i] = M[i][j];
      N[j][i] = N[i][j];
    }
}

inline double pow6(double v, int libm)
{
  if (libm)
    return std::pow(v, 6.0); /* Eigen 3.2 ArrayBase::pow(6) -> std::pow(double,double) */
  const double v2 = v * v;
  return (v2 * v2) * v2;
}

/* ---------------------------------------------------------------------------------------------------
 * Eigen::EigenSolver on M3 = sum n n^T (quadric.cpp:268-280) -- THIRD PARTY -- as far as the reference uses it: the unit
 * eigenvector of the smallest eigenvalue (its sign cancels at quadric.cpp:291-304).  Same scheme as solve_taubin, for a
 * symmetric 3 x 3: one Householder reflection to tridiagonal form, the smallest eigenvalue by bisection on the Sturm
 * sequence inside its Gershgorin bracket, the eigenvector from the twisted factorisation, back-transformed and
 * normalised; every multiply-add fused.  (Round 2 ran a cyclic Jacobi here: ~20 us of dependent divisions and square roots
 * on one GPU lane per sample.)  Pinned against numpy.linalg.eig (LAPACK dgeev) by tests/golden/e2e_lapack.npz.
 * A double smallest eigenvalue (all normals parallel: exactly planar patches) leaves the direction inside the
 * eigenspace to this arithmetic, as it is left to the solver's in the reference; an exactly diagonal M3 = diag(0, 0, n)
 * yields the x axis, as EigenSolver's identity eigenvectors would.
 * ------------------------------------------------------------------------------------------------- */
void smallest_eigvec3(const double M3[3][3], double axis[3])
{
  /* Householder: zero the (2,0) entry */
  const double x0 = M3[1][0], x1 = M3[2][0];
  const double sg = std::fma(x1, x1, x0 * x0);
  const double rt = std::sqrt(sg);
  const double g = (x0 >= 0.0) ? -rt : rt;
  const double h = std::fma(-x0, g, sg);
  const bool live = h > 0.0;
  const double rh = live ? 1.0 / h : 0.0;
  const double u1 = live ? x0 - g : 0.0, u2 = live ? x1 : 0.0;
  /* p = A22 u / h, kk = u.p / (2h), q = p - kk u, A22 -= u q^T + q u^T (lower triangle) */
  const double p1 = std::fma(M3[2][1], u2, M3[1][1] * u1) * rh;
  const double p2 = std::fma(M3[2][2], u2, M3[2][1] * u1) * rh;
  const double kk = (std::fma(u2, p2, u1 * p1) * rh) * 0.5;
  const double q1 = std::fma(-kk, u1, p1), q2 = std::fma(-kk, u2, p2);
  const double d[3] = { M3[0][0], std::fma(-q1, u1, std::fma(-u1, q1, M3[1][1])), std::fma(-q2, u2, std::fma(-u2, q2, M3[2][2])) };
  const double e[2] = { live ? g : x0, std::fma(-q2, u1, std::fma(-u2, q1, M3[2][1])) };
  const double e2[2] = { e[0] * e[0], e[1] * e[1] };
  /* bisection in [min_i (d_i - |e_i-1| - |e_i|), min_i d_i] */
  const double r0 = std::fabs(e[0]), r1 = std::fabs(e[0]) + std::fabs(e[1]), r2 = std::fabs(e[1]);
  double lo = d[0] - r0, hi = d[0];
  if (d[1] - r1 < lo)
    lo = d[1] - r1;
  if (d[2] - r2 < lo)
    lo = d[2] - r2;
  if (d[1] < hi)
    hi = d[1];
  if (d[2] < hi)
    hi = d[2];
  for (int it = 0; it < kBisectSteps; it++)
  {
    const double mid = std::fma(hi - lo, 0.5, lo);
    const double s1 = d[0] - mid;
    const double s2 = std::fma(d[1] - mid, s1, -e2[0]);
    const double s3 = std::fma(d[2] - mid, s2, -(e2[1] * s1));
    if (std::signbit(s1) || std::signbit(s2) || std::signbit(s3))
      hi = mid;
    else
      lo = mid;
  }
  const double sigma = lo;
  /* twisted factorisation of T - sigma I */
  double Dp[3], Dm[3], lf[2], ub[2];
  Dp[0] = pivot_floor(d[0] - sigma);
  lf[0] = e[0] / Dp[0];
  Dp[1] = pivot_floor(std::fma(-lf[0], e[0], d[1] - sigma));
  lf[1] = e[1] / Dp[1];
  Dp[2] = pivot_floor(std::fma(-lf[1], e[1], d[2] - sigma));
  Dm[2] = pivot_floor(d[2] - sigma);
  ub[1] = e[1] / Dm[2];
  Dm[1] = pivot_floor(std::fma(-ub[1], e[1], d[1] - sigma));
  ub[0] = e[0] / Dm[1];
  Dm[0] = pivot_floor(std::fma(-ub[0], e[0], d[0] - sigma));
  int ks = 0;
  double gmin = std::fabs((Dp[0] + Dm[0]) - (d[0] - sigma));
  for (int k = 1; k < 3; k++)
  {
    const double gk = std::fabs((Dp[k] + Dm[k]) - (d[k] - sigma));
    if (gk < gmin)
    {
      ks = k;
      gmin = gk;
    }
  }
  double z[3] = { 0.0, 0.0, 0.0 };
  z[ks] = 1.0;
  for (int i = ks - 1; i >= 0; i--)
    z[i] = -(lf[i] * z[i + 1]);
  for (int i = ks; i < 2; i++)
    z[i + 1] = -(ub[i] * z[i]);
  /* y = H z, normalised */
  const double sdot = std::fma(u2, z[2], u1 * z[1]) * rh;
  const double y0 = z[0], y1 = std::fma(-sdot, u1, z[1]), y2 = std::fma(-sdot, u2, z[2]);
  const double nn = std::sqrt(std::fma(y2, y2, std::fma(y1, y1, y0 * y0)));
  axis[0] = y0 / nn;
  axis[1] = y1 / nn;
  axis[2] = y2 / nn;
}

/* ---------------------------------------------------------------------------------------------------
 * a3+a4+a5  Quadric::fitQuadric -> findTaubinNormalAxis -> findAverageNormalAxis
 * (quadric.cpp:14-157, 159-251, 263-305) for one sample.  draws = the glibc rand() values this sample
 * consumes in ORC_NORMALS_RAND50 mode (quadric.cpp:184), or nullptr for the deterministic mode.
 * ------------------------------------------------------------------------------------------------- */
void fit_frame(const orc_params& P, const Cloud& cl, const std::vector<Neighbor>& nb, const float* sample_f,
  const int32_t* draws, orc_frame& F)
{
  std::memset(&F, 0, sizeof(F));
  for (int a = 0; a < 3; a++)
    F.sample[a] = (double) sample_f[a]; /* hand_search.cpp:95 */
  F.n_nb = (int32_t) nb.size();
  double M[10][10], N[10][10];
  accumulate_MN(cl, nb, M, N);
  double v[10], lambda = 0.0;
  if (!solve_taubin(M, N, v, &lambda))
  {
    F.valid = 0;
    return;
  }
  F.valid = 1;
  F.eigenvalue = lambda;
  for (int k = 0; k < 10; k++)
    F.params[k] = v[k];
  for (int k = 3; k < 6; k++)
    F.params[k] *= 0.5; /* quadric.cpp:153 */

  /* findTaubinNormalAxis (quadric.cpp:159-251) */
  const double a = F.params[0], b = F.params[1], c = F.params[2];
  const double d = 2.0 * F.params[3], e = 2.0 * F.params[4], f = 2.0 * F.params[5];
  const double g = F.params[6], h = F.params[7], i9 = F.params[8];
  const int n = (int) nb.size();
  int k_s = n;
  std::vector<int32_t> pick; /* neighbour slot of each subsample */
  if (P.normals_mode == ORC_NORMALS_RAND50 && n > 50)
  {
    k_s = 50;
    pick.resize(50);
    for (int t = 0; t < 50; t++)
      pick[t] = draws[t] % n; /* quadric.cpp:184 (no NaN in the cloud, 185-188 never loops) */
  }
  else
  {
    pick.resize(n);
    for (int t = 0; t < n; t++)
      pick[t] = t;
  }
  /* majority camera (quadric.cpp:215-226): first maximum wins */
  double num_source[2] = { 0.0, 0.0 };
  for (int t = 0; t < k_s; t++)
  {
    const int cs = cl.cam[nb[pick[t]].idx];
    if (cs == 0)
      num_source[0]++;
    else if (cs == 1)
      num_source[1]++;
  }
  F.majority_cam = (num_source[1] > num_source[0]) ? 1 : 0;

  /* normals = normalised quadric gradient (quadric.cpp:238-247) */
  std::vector<double> nrm(3 * (size_t) k_s);
  for (int t = 0; t < k_s; t++)
  {
    const float* p = cl.pt(nb[pick[t]].idx);
    const double x = p[0], y = p[1], z = p[2];
    const double fx = (((2.0 * a) * x + d * y) + f * z) + g;
    const double fy = (((2.0 * b) * y + d * x) + e * z) + h;
    const double fz = (((2.0 * c) * z + e * y) + f * x) + i9;
    const double mag = std::sqrt((fx * fx + fy * fy) + fz * fz);
    nrm[3 * t + 0] = fx / mag;
    nrm[3 * t + 1] = fy / mag;
    nrm[3 * t + 2] = fz / mag;
  }

  /* findAverageNormalAxis (quadric.cpp:263-305) */
  /* M = normals * normals^T (quadric.cpp:266) is an Eigen matrix product and the column sums below are an Eigen
   * colwise().sum(): the ORDER in which their n terms are added is Eigen's (GEMM blocking, packet width and FMA use
   * depend on the Eigen version and the build flags), not something the reference's source pins.  Both are DEFINED
   * here as LaneSum64: term i is added to partial i mod 64 (each partial accumulates in index order), and the 64
   * partials are combined by a balanced binary tree. */
  double M3[3][3];
  for (int r = 0; r < 3; r++)
    for (int q = r; q < 3; q++)
    {
      LaneSum64 acc;
      for (int t = 0; t < k_s; t++)
        acc.add(t, nrm[3 * t + r] * nrm[3 * t + q]);
      M3[r][q] = acc.total();
      M3[q][r] = M3[r][q];
    }
  double axis[3];
  smallest_eigvec3(M3, axis);

  /* max_index: argmax_j sum_i (n_i . n_j)^6, first maximum wins (quadric.cpp:283-284) */
  int max_index = 0;
  double best = 0.0;
  for (int j = 0; j < k_s; j++)
  {
    LaneSum64 acc;
    for (int i = 0; i < k_s; i++)
      acc.add(i, pow6(dot3(&nrm[3 * i], &nrm[3 * j]), P.pow6_libm));
    const double s = acc.total();
    if (j == 0 || s > best)
    {
      best = s;
      max_index = j;
    }
  }
  F.max_index = max_index;
  /* normal = normalise((I - a a^T) n_max)  (quadric.cpp:285-288) */
  double np_[3];
  for (int r = 0; r < 3; r++)
  {
    double pr[3];
    for (int q = 0; q < 3; q++)
      pr[q] = ((r == q) ? 1.0 : 0.0) - axis[r] * axis[q];
    np_[r] = (pr[0] * nrm[3 * max_index + 0] + pr[1] * nrm[3 * max_index + 1]) + pr[2] * nrm[3 * max_index + 2];
  }
  const double nn = std::sqrt((np_[0] * np_[0] + np_[1] * np_[1]) + np_[2] * np_[2]);
  double normal[3] = { np_[0] / nn, np_[1] / nn, np_[2] / nn };
  double binormal[3];
  cross3(axis, normal, binormal); /* quadric.cpp:291 */
  double s2s[3];
  for (int r = 0; r < 3; r++)
    s2s[r] = F.sample[r] - P.cam_origin[F.majority_cam][r]; /* quadric.cpp:294 */
  if (dot3(normal, s2s) > 0)
    for (int r = 0; r < 3; r++)
      normal[r] *= -1.0;
  if (dot3(binormal, s2s) > 0)
    for (int r = 0; r < 3; r++)
      binormal[r] *= -1.0;
  cross3(normal, binormal, axis); /* quadric.cpp:304 */
  for (int r = 0; r < 3; r++)
  {
    F.normal[r] = normal[r];
    F.axis[r] = axis[r];
    F.binormal[r] = binormal[r];
  }
}

/* ---------------------------------------------------------------------------------------------------
 * a10-a12  FingerHand (finger_hand.cpp:3-233), literal.
 * ------------------------------------------------------------------------------------------------- */
struct FingerHandO
{
  double fw, od, depth;
  double back_of_hand = 0, grasp_width = 0;
  double fs[20];
  bool fingers[20];
  bool hand[10];
  const double* px = nullptr; /* points_ row 0 */
  const double* py = nullptr; /* points_ row 1 */
  int np = 0;
  double bottom[2], surface[2];

  FingerHandO(double finger_width, double hand_outer_diameter, double hand_depth)
    : fw(finger_width), od(hand_outer_diameter), depth(hand_depth)
  {
    /* finger_hand.cpp:8-15: fs_half = LinSpaced(10, 0, od - fw) -> low + i*step (Eigen 3.2 linspaced_op) */
    const double low = 0.0, high = od - fw;
    const double step = (high - low) / 9.0;
    for (int i = 0; i < 10; i++)
    {
      const double h = low + i * step;
      fs[i] = (h - od) + fw;
      fs[10 + i] = h;
    }
    for (int i = 0; i < 20; i++)
      fingers[i] = false;
    for (int i = 0; i < 10; i++)
      hand[i] = false;
  }

  void evaluateFingers(double bite) /* finger_hand.cpp:20-98 */
  {
    back_of_hand = -1.0 * (depth - bite);
    for (int i = 0; i < 20; i++)
      fingers[i] = false;
    std::vector<int> cropped;
    for (int i = 0; i < np; i++)
      if (py[i] < bite)
      {
        cropped.push_back(i);
        if (py[i] < back_of_hand)
          return;
      }
    const int m = 20;
    for (int i = 0; i < m; i++)
    {
      int num_in_gap = 0;
      for (size_t j = 0; j < cropped.size(); j++)
      {
        const double x = px[cropped[j]];
        if (x > fs[i] && x < fs[i] + fw)
          num_in_gap++;
      }
      if (num_in_gap == 0)
      {
        int sum = 0;
        if (i <= m / 2)
        {
          for (size_t j = 0; j < cropped.size(); j++)
            if (px[cropped[j]] > fs[i] + fw)
              sum++;
        }
        else
        {
          for (size_t j = 0; j < cropped.size(); j++)
            if (px[cropped[j]] < fs[i])
              sum++;
        }
        if (sum > 0)
          fingers[i] = true;
      }
    }
  }

  void evaluateHand() /* finger_hand.cpp:100-115 */
  {
    for (int i = 0; i < 10; i++)
      hand[i] = fingers[i] && fingers[10 + i];
  }

  int handSum() const
  {
    int s = 0;
    for (int i = 0; i < 10; i++)
      s += hand[i] ? 1 : 0;
    return s;
  }

  /* finger_hand.cpp:173-233; returns the number of successful deepen steps, *e_out = eroded index */
  int deepenHand(double init_deepness, double max_deepness, int* e_out)
  {
    std::vector<int> hand_idx;
    for (int i = 0; i < 10; i++)
      if (hand[i])
        hand_idx.push_back(i);
    if (hand_idx.empty())
      return 0;
    const int e = hand_idx[(int) std::ceil(hand_idx.size() / 2.0) - 1];
    *e_out = e;
    FingerHandO new_hand = *this;
    FingerHandO last_new_hand = new_hand;
    int steps = 0;
    const double deepen_step_size = 0.005;
    for (double d = init_deepness + deepen_step_size; d <= max_deepness; d += deepen_step_size)
    {
      new_hand.evaluateFingers(d);
      new_hand.evaluateHand();
      if (!new_hand.hand[e])
        break;
      last_new_hand = new_hand;
      steps++;
    }
    *this = last_new_hand;
    for (int i = 0; i < 10; i++)
      hand[i] = (i == e);
    return steps;
  }

  void evaluateGraspParameters(double bite) /* finger_hand.cpp:117-171 */
  {
    double fs_sum = 0.0;
    int hsum = 0;
    for (int i = 0; i < 10; i++)
    {
      fs_sum += fs[i] * (hand[i] ? 1.0 : 0.0);
      hsum += hand[i] ? 1 : 0;
    }
    const double hor_pos = (od / 2.0) + (fs_sum / hsum);
    double ymax = py[0], ymin = py[0];
    for (int i = 1; i < np; i++)
    {
      if (py[i] > ymax)
        ymax = py[i];
      if (py[i] < ymin)
        ymin = py[i];
    }
    bottom[0] = hor_pos;
    bottom[1] = ymax;
    surface[0] = hor_pos;
    surface[1] = ymin;
    std::vector<int> hand_idx;
    for (int i = 0; i < 10; i++)
      if (hand[i])
        hand_idx.push_back(i);
    const int e = hand_idx[hand_idx.size() / 2];
    const double left = fs[e], right = fs[10 + e];
    double mx = -100000.0, mn = 100000.0;
    for (int i = 0; i < np; i++)
      if (py[i] < bite && px[i] > left && px[i] < right)
      {
        if (px[i] < mn)
          mn = px[i];
        if (px[i] > mx)
          mx = px[i];
      }
    grasp_width = mx - mn;
  }
};

inline void mat3mul(const double A[3][3], const double B[3][3], double C[3][3])
{
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      C[i][j] = (A[i][0] * B[0][j] + A[i][1] * B[1][j]) + A[i][2] * B[2][j];
}

inline void mat3vec(const double A[3][3], const double* v, double* o)
{
  for (int i = 0; i < 3; i++)
    o[i] = (A[i][0] * v[0] + A[i][1] * v[1]) + A[i][2] * v[2];
}

/* ---------------------------------------------------------------------------------------------------
 * a17  Learning::convertToImage (learning.cpp:320-365) with createInstance (375-400): 80 rows x 100 cols.
 * ------------------------------------------------------------------------------------------------- */
void make_image(const std::vector<double>& bx, const std::vector<double>& by, bool positive_x, uint8_t* image)
{
  std::memset(image, 0, 8000);
  const double HL0 = -0.05, HL1 = 0.05, VL0 = 0.0;
  const double cell = (HL1 - HL0) / (double) 100;
  for (size_t i = 0; i < bx.size(); i++)
  {
    double hx = positive_x ? (bx[i] - HL0) / cell : (-bx[i] - HL0) / cell;
    double vy = (by[i] - VL0) / cell;
    int hc = (int) std::floor(hx);
    int vc = (int) std::floor(vy);
    hc = std::min(99, std::max(0, hc));
    vc = std::min(79, std::max(0, vc));
    image[(80 - 1 - vc) * 100 + hc] = 255;
  }
}

/* ---------------------------------------------------------------------------------------------------
 * a7-a9,a13  HandSearch::findHands (private, hand_search.cpp:116-206) body for one sample:
 * gather (154-160), RotatingHand ctor (rotating_hand.cpp:4-16), transformPoints (19-75),
 * evaluateHand (78-177), Antipodal::evaluateGrasp (antipodal.cpp:12-86).
 * normals: per-cloud-point normals (cloud_normals_, 3 doubles each) or nullptr (all zero).
 * ------------------------------------------------------------------------------------------------- */
struct HandOut
{
  orc_hypothesis h;
  std::vector<uint8_t> image;
  std::vector<uint8_t> image_cam[2]; /* createInstance(h, cam_pos, cam = 0 / 1): only that camera's points */
  std::vector<double> pts;           /* points_for_learning, 3 x n_b column-major (rotating_hand.cpp:132-139) */
  std::vector<int32_t> pts_cam;      /* camera id per column (indices_cam1 / indices_cam2, 143-151) */
};

void hands_for_sample(const orc_params& P, const Cloud& cl, const std::vector<Neighbor>& nb, const orc_frame& F,
  const double* normals, int sample_pos, int sample_cam, int want_images, std::vector<HandOut>& out)
{
  out.clear();
  if (!F.valid)
    return;
  const int n = (int) nb.size();
  float sample_f[3] = { (float) F.sample[0], (float) F.sample[1], (float) F.sample[2] }; /* hand_search.cpp:141-144 */
  /* frame_ << normal, normal x axis, axis (rotating_hand.cpp:24-25) */
  double fr[3][3];
  double nxa[3];
  cross3(F.normal, F.axis, nxa);
  for (int r = 0; r < 3; r++)
  {
    fr[r][0] = F.normal[r];
    fr[r][1] = nxa[r];
    fr[r][2] = F.axis[r];
  }
  /* points_ = frame^T * centered (26), normals_ likewise (33); crop |z| < hand_height (37-51) */
  std::vector<double> X, Y, Z, NX, NY;
  std::vector<int> CAMS;
  X.reserve(n);
  Y.reserve(n);
  for (int j = 0; j < n; j++)
  {
    const float* p = cl.pt(nb[j].idx);
    const double cx = (double) (p[0] - sample_f[0]); /* float subtraction, hand_search.cpp:157-158 */
    const double cy = (double) (p[1] - sample_f[1]);
    const double cz = (double) (p[2] - sample_f[2]);
    const double tx = (fr[0][0] * cx + fr[1][0] * cy) + fr[2][0] * cz;
    const double ty = (fr[0][1] * cx + fr[1][1] * cy) + fr[2][1] * cz;
    const double tz = (fr[0][2] * cx + fr[1][2] * cy) + fr[2][2] * cz;
    if (tz > -1.0 * P.hand_height && tz < P.hand_height)
    {
      X.push_back(tx);
      Y.push_back(ty);
      Z.push_back(tz);
      CAMS.push_back(cl.cam[nb[j].idx]);
      if (normals)
      {
        const double* nn = normals + 3 * (size_t) nb[j].idx;
        NX.push_back((fr[0][0] * nn[0] + fr[1][0] * nn[1]) + fr[2][0] * nn[2]);
        NY.push_back((fr[0][1] * nn[0] + fr[1][1] * nn[1]) + fr[2][1] * nn[2]);
      }
    }
  }
  const int nc = (int) X.size();
  double cams[2][3]; /* cams_ columns = camera origin - sample (hand_search.cpp:164-166) */
  for (int c = 0; c < 2; c++)
    for (int r = 0; r < 3; r++)
      cams[c][r] = P.cam_origin[c][r] - F.sample[r];

  FingerHandO finger_hand(P.finger_width, P.hand_outer_diameter, P.hand_depth);
  std::vector<double> XR(nc), YR(nc), NXR(nc);
  const double step = (M_PI - (-1.0 * M_PI)) / 8.0; /* LinSpaced(9, -pi, pi) (rotating_hand.cpp:14) */
  for (int o = 0; o < 8; o++)
  {
    const double ang = -1.0 * M_PI + o * step;
    const double cs = std::cos(ang), sn = std::sin(ang);
    const double rot[3][3] = { { cs, -1.0 * sn, 0.0 }, { sn, cs, 0.0 }, { 0.0, 0.0, 1.0 } };
    double rotT[3][3];
    for (int r = 0; r < 3; r++)
      for (int q = 0; q < 3; q++)
        rotT[r][q] = rot[q][r];
    double T[3][3];
    mat3mul(fr, rotT, T); /* frame_ * rot^T (rotating_hand.cpp:96,104,120-121) */
    const double ex[3] = { 1.0, 0.0, 0.0 }, ey[3] = { 0.0, 1.0, 0.0 };
    double approach[3], binormal[3];
    mat3vec(T, ey, approach);
    if (dot3(approach, cams[0]) > 0 && dot3(approach, cams[1]) > 0) /* rotating_hand.cpp:99-102 */
      continue;
    mat3vec(T, ex, binormal);
    if (nc == 0)
      continue; /* Eigen would read empty matrices; no point can make a finger true */
    for (int j = 0; j < nc; j++)
    {
      /* points_rot = rot * points_ (91); the 0.0*z terms cannot change a comparison and are dropped */
      XR[j] = rot[0][0] * X[j] + rot[0][1] * Y[j];
      YR[j] = rot[1][0] * X[j] + rot[1][1] * Y[j];
      if (normals)
        NXR[j] = rot[0][0] * NX[j] + rot[0][1] * NY[j];
    }
    finger_hand.px = XR.data();
    finger_hand.py = YR.data();
    finger_hand.np = nc;
    finger_hand.evaluateFingers(P.init_bite);
    finger_hand.evaluateHand();
    if (finger_hand.handSum() > 0)
    {
      int e = -1;
      const int steps = finger_hand.deepenHand(P.init_bite, P.hand_depth, &e);
      finger_hand.evaluateGraspParameters(P.init_bite);
      double surf_l[3] = { finger_hand.surface[0], finger_hand.surface[1], 0.0 };
      double bot_l[3] = { finger_hand.bottom[0], finger_hand.bottom[1], 0.0 };
      double surface[3], bottom[3];
      mat3vec(T, surf_l, surface);
      mat3vec(T, bot_l, bottom);
      const double box_y = finger_hand.back_of_hand + finger_hand.depth; /* rotating_hand.cpp:127 */
      std::vector<double> bx, by;
      std::vector<double> bxc[2], byc[2]; /* indices_cam1 / indices_cam2 (rotating_hand.cpp:143-151) */
      std::vector<double> pts3;
      std::vector<int32_t> pts_cam;
      int numl = 0, numr = 0, nbox = 0;
      const double cos_thresh = std::cos(20 * M_PI / 180.0); /* antipodal.cpp:16 */
      for (int j = 0; j < nc; j++)
        if (YR[j] < box_y)
        {
          nbox++;
          bx.push_back(XR[j] - surface[0]); /* rotating_hand.cpp:138 (world-frame offset, replicated as-is) */
          by.push_back(YR[j] - surface[1]);
          if (want_images == 3) /* the whole column: rot leaves z alone, and surface is subtracted as a 3-vector */
          {
            pts3.push_back(bx.back());
            pts3.push_back(by.back());
            pts3.push_back(Z[j] - surface[2]);
            pts_cam.push_back(CAMS[j]);
          }
          if (want_images == 2 && (CAMS[j] == 0 || CAMS[j] == 1))
          {
            bxc[CAMS[j]].push_back(bx.back());
            byc[CAMS[j]].push_back(by.back());
          }
          if (normals)
          {
            if (-1.0 * NXR[j] > cos_thresh)
              numl++;
            if (NXR[j] > cos_thresh)
              numr++;
          }
        }
      HandOut ho;
      std::memset(&ho.h, 0, sizeof(ho.h));
      for (int r = 0; r < 3; r++)
      {
        ho.h.axis[r] = F.axis[r];
        ho.h.approach[r] = approach[r];
        ho.h.binormal[r] = binormal[r];
        ho.h.bottom[r] = bottom[r] + F.sample[r];   /* rotating_hand.cpp:153-154 */
        ho.h.surface[r] = surface[r] + F.sample[r];
      }
      ho.h.width = finger_hand.grasp_width;
      ho.h.sample = sample_pos;
      ho.h.orientation = o;
      ho.h.cam_source = sample_cam;
      ho.h.n_in_box = nbox;
      ho.h.valid = 1;
      ho.h.finger_index = e;
      ho.h.depth_index = steps;
      /* antipodal.cpp:12-86: HALF iff numl>6 or numr>6, FULL iff both (order independent) */
      const bool half = (numl > 6) || (numr > 6);
      const bool full = (numl > 6) && (numr > 6);
      ho.h.half_antipodal = (half || full) ? 1 : 0;
      ho.h.full_antipodal = full ? 1 : 0;
      if (want_images)
      {
        /* learning.cpp:375-400 + 320-365 with cam_pos = camera origins (localization.cpp:148-150) */
        double s2c[3];
        const int cs_i = (sample_cam == 1) ? 1 : 0;
        for (int r = 0; r < 3; r++)
          s2c[r] = ho.h.surface[r] - P.cam_origin[cs_i][r];
        ho.image.resize(8000);
        make_image(bx, by, dot3(binormal, s2c) > 0, ho.image.data());
        ho.pts = std::move(pts3);
        ho.pts_cam = std::move(pts_cam);
        for (int c = 0; c < 2 && want_images == 2; c++) /* same source_to_center, the camera's subset of pts */
        {
          ho.image_cam[c].resize(8000);
          make_image(bxc[c], byc[c], dot3(binormal, s2c) > 0, ho.image_cam[c].data());
        }
      }
      out.push_back(std::move(ho));
    }
  }
}

/* ---------------------------------------------------------------------------------------------------
 * a18  cv::HOGDescriptor::compute -- THIRD PARTY (OpenCV 2.4 modules/objdetect/src/hog.cpp), restated:
 * computeGradient (gamma sqrt LUT, BORDER_REFLECT_101, fastAtan2 polynomial, 9 unsigned bins with linear
 * bin interpolation), HOGCache::init (Gaussian sigma 4, cell bilinear weights, count1/2/4 pixel groups),
 * getBlock (sequential float accumulation in pixData order), normalizeBlockHistogram (L2-Hys 0.2).
 * Call site learning.cpp:194-195,220: winSize 64x64, winStride 32x32, padding 0 -> windows at x=0,32.
 * ------------------------------------------------------------------------------------------------- */
struct HogTables
{
  struct PixData
  {
    int gradOfs, qangleOfs;
    int histOfs[4];
    float histWeights[4];
    float gradWeight;
  };
  std::vector<PixData> pix;
  int count1 = 0, count2 = 0, count4 = 0;
  HogTables()
  {
    const int bs = 16, cellsz = 8, ncell = 2, nbins = 9, raw = 256;
    const int W = 100; /* grad.cols */
    float weights[16][16];
    const float sigma = 4.0f; /* winSigma=-1 -> (blockSize.w+blockSize.h)/8 */
    const float scale = 1.f / (sigma * sigma * 2);
    for (int i = 0; i < bs; i++)
      for (int j = 0; j < bs; j++)
      {
        const float di = i - bs * 0.5f, dj = j - bs * 0.5f;
        weights[i][j] = std::exp(-(di * di + dj * dj) * scale);
      }
    std::vector<PixData> tmp(raw * 3);
    int c1 = 0, c2 = 0, c4 = 0;
    for (int j = 0; j < bs; j++)
      for (int i = 0; i < bs; i++)
      {
        PixData* data = nullptr;
        float cellX = (j + 0.5f) / cellsz - 0.5f;
        float cellY = (i + 0.5f) / cellsz - 0.5f;
        int icellX0 = (int) std::floor(cellX), icellY0 = (int) std::floor(cellY);
        int icellX1 = icellX0 + 1, icellY1 = icellY0 + 1;
        cellX -= icellX0;
        cellY -= icellY0;
        if ((unsigned) icellX0 < (unsigned) ncell && (unsigned) icellX1 < (unsigned) ncell)
        {
          if ((unsigned) icellY0 < (unsigned) ncell && (unsigned) icellY1 < (unsigned) ncell)
          {
            data = &tmp[raw * 2 + (c4++)];
            data->histOfs[0] = (icellX0 * ncell + icellY0) * nbins;
            data->histWeights[0] = (1.f - cellX) * (1.f - cellY);
            data->histOfs[1] = (icellX1 * ncell + icellY0) * nbins;
            data->histWeights[1] = cellX * (1.f - cellY);
            data->histOfs[2] = (icellX0 * ncell + icellY1) * nbins;
            data->histWeights[2] = (1.f - cellX) * cellY;
            data->histOfs[3] = (icellX1 * ncell + icellY1) * nbins;
            data->histWeights[3] = cellX * cellY;
          }
          else
          {
            data = &tmp[raw + (c2++)];
            if ((unsigned) icellY0 < (unsigned) ncell)
            {
              icellY1 = icellY0;
              cellY = 1.f - cellY;
            }
            data->histOfs[0] = (icellX0 * ncell + icellY1) * nbins;
            data->histWeights[0] = (1.f - cellX) * cellY;
            data->histOfs[1] = (icellX1 * ncell + icellY1) * nbins;
            data->histWeights[1] = cellX * cellY;
            data->histOfs[2] = data->histOfs[3] = 0;
            data->histWeights[2] = data->histWeights[3] = 0;
          }
        }
        else
        {
          if ((unsigned) icellX0 < (unsigned) ncell)
          {
            icellX1 = icellX0;
            cellX = 1.f - cellX;
          }
          if ((unsigned) icellY0 < (unsigned) ncell && (unsigned) icellY1 < (unsigned) ncell)
          {
            data = &tmp[raw + (c2++)];
            data->histOfs[0] = (icellX1 * ncell + icellY0) * nbins;
            data->histWeights[0] = cellX * (1.f - cellY);
            data->histOfs[1] = (icellX1 * ncell + icellY1) * nbins;
            data->histWeights[1] = cellX * cellY;
            data->histOfs[2] = data->histOfs[3] = 0;
            data->histWeights[2] = data->histWeights[3] = 0;
          }
          else
          {
            data = &tmp[c1++];
            if ((unsigned) icellY0 < (unsigned) ncell)
            {
              icellY1 = icellY0;
              cellY = 1.f - cellY;
            }
            data->histOfs[0] = (icellX1 * ncell + icellY1) * nbins;
            data->histWeights[0] = cellX * cellY;
            data->histOfs[1] = data->histOfs[2] = data->histOfs[3] = 0;
            data->histWeights[1] = data->histWeights[2] = data->histWeights[3] = 0;
          }
        }
        data->gradOfs = (W * i + j) * 2;
        data->qangleOfs = (W * i + j) * 2;
        data->gradWeight = weights[i][j];
      }
    pix.resize(raw);
    for (int k = 0; k < c1; k++)
      pix[k] = tmp[k];
    for (int k = 0; k < c2; k++)
      pix[k + c1] = tmp[k + raw];
    for (int k = 0; k < c4; k++)
      pix[k + c1 + c2] = tmp[k + raw * 2];
    count1 = c1;
    count2 = c1 + c2;
    count4 = c1 + c2 + c4;
  }
};

const HogTables& hog_tables()
{
  static HogTables t;
  return t;
}

inline int border101(int p, int len)
{
  if (p < 0)
    return -p;
  if (p >= len)
    return 2 * len - 2 - p;
  return p;
}

inline float fast_atan2_deg(float y, float x)
{
  /* OpenCV 2.4 modules/core/src/mathfuncs.cpp FastAtan2_32f (scalar tail) */
  const float sc = (float) (180 / M_PI);
  const float p1 = 0.9997878412794807f * sc, p3 = -0.3258083974640975f * sc;
  const float p5 = 0.1555786518463281f * sc, p7 = -0.04432655554792128f * sc;
  const float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay)
  {
    c = ay / (ax + (float) 2.2204460492503131e-16);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  else
  {
    c = ax / (ay + (float) 2.2204460492503131e-16);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0)
    a = 180.f - a;
  if (y < 0)
    a = 360.f - a;
  return a;
}

void hog_compute(const uint8_t* img, float* desc)
{
  const int W = 100, H = 80, nbins = 9;
  const HogTables& T = hog_tables();
  std::vector<float> grad((size_t) W * H * 2);
  std::vector<uint8_t> qangle((size_t) W * H * 2);
  float lut[256];
  for (int i = 0; i < 256; i++)
    lut[i] = std::sqrt((float) i); /* gammaCorrection = true */
  const float angleScale = (float) (nbins / M_PI);
  const float rad = (float) (M_PI / 180);
  for (int y = 0; y < H; y++)
  {
    const uint8_t* cur = img + W * border101(y, H);
    const uint8_t* prev = img + W * border101(y - 1, H);
    const uint8_t* next = img + W * border101(y + 1, H);
    for (int x = 0; x < W; x++)
    {
      const int x1 = border101(x, W);
      const float dx = (float) (lut[cur[border101(x + 1, W)]] - lut[cur[border101(x - 1, W)]]);
      const float dy = (float) (lut[next[x1]] - lut[prev[x1]]);
      const float mag = std::sqrt(dx * dx + dy * dy);
      float angle = (float) (fast_atan2_deg(dy, dx) * rad);
      angle = angle * angleScale - 0.5f;
      int hidx = (int) std::floor(angle);
      angle -= hidx;
      grad[(y * W + x) * 2] = mag * (1.f - angle);
      grad[(y * W + x) * 2 + 1] = mag * angle;
      if (hidx < 0)
        hidx += nbins;
      else if (hidx >= nbins)
        hidx -= nbins;
      qangle[(y * W + x) * 2] = (uint8_t) hidx;
      hidx++;
      hidx &= hidx < nbins ? -1 : 0;
      qangle[(y * W + x) * 2 + 1] = (uint8_t) hidx;
    }
  }
  const int nbx = 7, nby = 7, bhs = 36;
  for (int win = 0; win < 2; win++)
  {
    const int wx = win * 32, wy = 0;
    for (int j = 0; j < nbx; j++)
      for (int i = 0; i < nby; i++)
      {
        float* hist = desc + win * 1764 + (j * nby + i) * bhs;
        const int ptx = wx + j * 8, pty = wy + i * 8;
        const float* gradPtr = grad.data() + ((size_t) pty * W + ptx) * 2;
        const uint8_t* qPtr = qangle.data() + ((size_t) pty * W + ptx) * 2;
        for (int k = 0; k < bhs; k++)
          hist[k] = 0.f;
        int k = 0;
        for (; k < T.count4; k++)
        {
          const HogTables::PixData& pk = T.pix[k];
          const float a0 = gradPtr[pk.gradOfs], a1 = gradPtr[pk.gradOfs + 1];
          const int h0 = qPtr[pk.qangleOfs], h1 = qPtr[pk.qangleOfs + 1];
          const int ncontrib = (k < T.count1) ? 1 : (k < T.count2 ? 2 : 4);
          for (int cidx = 0; cidx < ncontrib; cidx++)
          {
            float* hh = hist + pk.histOfs[cidx];
            const float w = pk.gradWeight * pk.histWeights[cidx];
            const float t0 = hh[h0] + a0 * w;
            const float t1 = hh[h1] + a1 * w;
            hh[h0] = t0;
            hh[h1] = t1;
          }
        }
        /* normalizeBlockHistogram */
        float sum = 0;
        for (k = 0; k < bhs; k++)
          sum += hist[k] * hist[k];
        float scale = 1.f / (std::sqrt(sum) + bhs * 0.1f);
        const float thresh = 0.2f;
        sum = 0;
        for (k = 0; k < bhs; k++)
        {
          hist[k] = std::min(hist[k] * scale, thresh);
          sum += hist[k] * hist[k];
        }
        scale = 1.f / (std::sqrt(sum) + 1e-3f);
        for (k = 0; k < bhs; k++)
          hist[k] *= scale;
      }
  }
}

/* a19  CvSVM::predict, linear kernel, 1 support vector -- THIRD PARTY (OpenCV 2.4 modules/ml/src/svm.cpp):
 * calc_non_rbf_base: float products, 4 summed in float, accumulated in double; result stored as float;
 * sum = -rho + alpha*that; class_labels[sum > 0 ? 0 : 1]; the reference keeps prediction == 1
 * (learning.cpp:225-227), i.e. sum <= 0. */
int svm_keep(const float* desc, const float* w, int n_w, double rho, double* sum_out)
{
  double s = 0;
  int k = 0;
  for (; k <= n_w - 4; k += 4)
    s += w[k] * desc[k] + w[k + 1] * desc[k + 1] + w[k + 2] * desc[k + 2] + w[k + 3] * desc[k + 3];
  for (; k < n_w; k++)
    s += w[k] * desc[k];
  const float res = (float) (s * 1.0 + 0.0);
  const double sum = -rho + 1.0 * res;
  if (sum_out)
    *sum_out = sum;
  return (sum > 0) ? 0 : 1;
}

struct Searcher
{
  Cloud cl;
  GridIndex g_taubin, g_hands, g_normals;
};

void sample_point(const Cloud& cl, int32_t idx, float q[3])
{
  q[0] = cl.pt(idx)[0];
  q[1] = cl.pt(idx)[1];
  q[2] = cl.pt(idx)[2];
}

/* HandSearch::findQuadrics (hand_search.cpp:65-113): OMP loop A. */
void fit_frames_impl(const orc_params& P, const Cloud& cl, const GridIndex& grid, double radius,
  const int32_t* sample_idx, int64_t S, orc_frame* frames, GlibcRand* rng)
{
  /* RAND50: the reference consumes rand() in loop order on one thread; reproduce that order by assigning
   * each sample its 50 draws up front (needs the neighbour counts first). */
  std::vector<int32_t> draws;
  std::vector<int64_t> draw_ofs(S, -1);
  if (P.normals_mode == ORC_NORMALS_RAND50)
  {
    std::vector<int32_t> counts(S);
#pragma omp parallel for num_threads(P.num_threads) schedule(static)
    for (int64_t i = 0; i < S; i++)
    {
      std::vector<Neighbor> nb;
      float q[3];
      sample_point(cl, sample_idx[i], q);
      grid.query(cl, q, radius, nb);
      counts[i] = (int32_t) nb.size();
    }
    int64_t total = 0;
    for (int64_t i = 0; i < S; i++)
      if (counts[i] > 50)
      {
        draw_ofs[i] = total;
        total += 50;
      }
    draws.resize(total);
    for (int64_t k = 0; k < total; k++)
      draws[k] = rng->next();
  }
#pragma omp parallel for num_threads(P.num_threads) schedule(static)
  for (int64_t i = 0; i < S; i++)
  {
    std::vector<Neighbor> nb;
    float q[3];
    sample_point(cl, sample_idx[i], q);
    grid.query(cl, q, radius, nb);
    fit_frame(P, cl, nb, q, draw_ofs[i] >= 0 ? &draws[draw_ofs[i]] : nullptr, frames[i]);
  }
}

int hands_impl(const orc_params& P, const Cloud& cl, const GridIndex& grid, const int32_t* sample_idx, int64_t S,
  const orc_frame* frames, const double* normals, orc_hypothesis* out, int64_t cap, int64_t* n_out, int32_t* nh_out,
  uint8_t* images_out, uint8_t* cam_images_out = nullptr, std::vector<double>* pts_out = nullptr,
  std::vector<int32_t>* pts_cam_out = nullptr, std::vector<int64_t>* pts_ofs_out = nullptr)
{
  std::vector<std::vector<HandOut>> lists(S);
#pragma omp parallel for num_threads(P.num_threads) schedule(static)
  for (int64_t i = 0; i < S; i++)
  {
    std::vector<Neighbor> nb;
    float q[3] = { (float) frames[i].sample[0], (float) frames[i].sample[1], (float) frames[i].sample[2] };
    grid.query(cl, q, P.nn_radius_hands, nb);
    if (nh_out)
      nh_out[i] = (int32_t) nb.size();
    /* hands_cam_source(i) = pts_cam_source(indices[i]) (hand_search.cpp:40-42; defined so for explicit indices) */
    hands_for_sample(P, cl, nb, frames[i], normals, (int) i, cl.cam[sample_idx[i]],
      pts_out ? 3 : (cam_images_out ? 2 : (images_out != nullptr ? 1 : 0)), lists[i]);
  }
  int64_t k = 0;
  for (int64_t i = 0; i < S; i++) /* concatenation, hand_search.cpp:194-200 */
    for (size_t j = 0; j < lists[i].size(); j++)
    {
      if (k < cap)
      {
        out[k] = lists[i][j].h;
        if (images_out)
          std::memcpy(images_out + k * 8000, lists[i][j].image.data(), 8000);
        for (int c = 0; c < 2 && cam_images_out; c++)
          std::memcpy(cam_images_out + (k * 2 + c) * 8000, lists[i][j].image_cam[c].data(), 8000);
        if (pts_out)
        {
          pts_ofs_out->push_back((int64_t) pts_cam_out->size());
          pts_out->insert(pts_out->end(), lists[i][j].pts.begin(), lists[i][j].pts.end());
          pts_cam_out->insert(pts_cam_out->end(), lists[i][j].pts_cam.begin(), lists[i][j].pts_cam.end());
        }
      }
      k++;
    }
  *n_out = k;
  return (k <= cap) ? 0 : -2;
}

} // namespace

extern "C" {

int64_t orc_radius_search(const float* xyz, int64_t stride_floats, int64_t n, const float q[3], double radius,
  int32_t* idx_out, float* d2_out, int64_t cap)
{
  Cloud cl{ xyz, stride_floats, nullptr, n };
  GridIndex g;
  g.build(cl, radius);
  std::vector<Neighbor> nb;
  g.query(cl, q, radius, nb);
  for (int64_t i = 0; i < (int64_t) nb.size() && i < cap; i++)
  {
    idx_out[i] = nb[i].idx;
    d2_out[i] = nb[i].d2;
  }
  return (int64_t) nb.size();
}

int orc_fit_frames(const orc_params* p, const float* xyz, int64_t stride_floats, const int32_t* cam, int64_t n,
  const int32_t* sample_idx, int64_t n_samples, double radius, orc_frame* frames_out)
{
  Cloud cl{ xyz, stride_floats, cam, n };
  GridIndex g;
  g.build(cl, radius);
  GlibcRand rng(p->rand_seed);
  fit_frames_impl(*p, cl, g, radius, sample_idx, n_samples, frames_out, &rng);
  return 0;
}

static int find_hands_full(const orc_params* p, const float* xyz, int64_t stride_floats, const int32_t* cam, int64_t n,
  const int32_t* sample_idx, int64_t n_samples, int calculates_antipodal, orc_hypothesis* out, int64_t cap,
  int64_t* n_out, orc_frame* frames_out, int32_t* nh_out, uint8_t* images_out, uint8_t* cam_images_out)
{
  Cloud cl{ xyz, stride_floats, cam, n };
  GridIndex g_t, g_h;
  g_t.build(cl, p->nn_radius_taubin);
  g_h.build(cl, p->nn_radius_hands);
  GlibcRand rng(p->rand_seed);
  std::vector<double> normals;
  if (calculates_antipodal)
  {
    /* hand_search.cpp:17-26: findQuadrics over ALL points with r = 0.01; cloud_normals_.col(i) = normal */
    GridIndex g_n;
    g_n.build(cl, p->nn_radius_normals);
    std::vector<int32_t> all(n);
    for (int64_t i = 0; i < n; i++)
      all[i] = (int32_t) i;
    std::vector<orc_frame> fr(n);
    fit_frames_impl(*p, cl, g_n, p->nn_radius_normals, all.data(), n, fr.data(), &rng);
    normals.assign(3 * (size_t) n, 0.0);
    for (int64_t i = 0; i < n; i++)
      if (fr[i].valid)
        for (int r = 0; r < 3; r++)
          normals[3 * i + r] = fr[i].normal[r];
  }
  std::vector<orc_frame> frames(n_samples);
  fit_frames_impl(*p, cl, g_t, p->nn_radius_taubin, sample_idx, n_samples, frames.data(), &rng);
  if (calculates_antipodal)
    for (int64_t i = 0; i < n_samples; i++) /* hand_search.cpp:102 also runs for the sample pass */
      if (frames[i].valid)
        for (int r = 0; r < 3; r++)
          normals[3 * (size_t) sample_idx[i] + r] = frames[i].normal[r];
  if (frames_out)
    std::memcpy(frames_out, frames.data(), sizeof(orc_frame) * n_samples);
  return hands_impl(*p, cl, g_h, sample_idx, n_samples, frames.data(), calculates_antipodal ? normals.data() : nullptr,
    out, cap, n_out, nh_out, images_out, cam_images_out);
}

int orc_find_hands(const orc_params* p, const float* xyz, int64_t stride_floats, const int32_t* cam, int64_t n,
  const int32_t* sample_idx, int64_t n_samples, int calculates_antipodal, orc_hypothesis* out, int64_t cap,
  int64_t* n_out, orc_frame* frames_out, int32_t* nh_out, uint8_t* images_out)
{
  return find_hands_full(p, xyz, stride_floats, cam, n, sample_idx, n_samples, calculates_antipodal, out, cap, n_out,
    frames_out, nh_out, images_out, nullptr);
}

/* a14: the variable part of GraspHypothesis -- points_for_learning and the camera id of each column -- for every
 * hypothesis of a (non-antipodal) search.  ofs_out[k] .. ofs_out[k+1] are hypothesis k's columns in pts_out (3 doubles
 * each) / cam_out.  Returns the total number of columns (may exceed pts_cap: then nothing is copied). */
int64_t orc_find_hands_points(const orc_params* p, const float* xyz, int64_t stride_floats, const int32_t* cam, int64_t n,
  const int32_t* sample_idx, int64_t n_samples, orc_hypothesis* out, int64_t cap, int64_t* n_out, double* pts_out,
  int32_t* cam_out, int64_t pts_cap, int64_t* ofs_out)
{
  Cloud cl{ xyz, stride_floats, cam, n };
  GridIndex g_t, g_h;
  g_t.build(cl, p->nn_radius_taubin);
  g_h.build(cl, p->nn_radius_hands);
  GlibcRand rng(p->rand_seed);
  std::vector<orc_frame> frames(n_samples);
  fit_frames_impl(*p, cl, g_t, p->nn_radius_taubin, sample_idx, n_samples, frames.data(), &rng);
  std::vector<double> pts;
  std::vector<int32_t> pc;
  std::vector<int64_t> ofs;
  std::vector<uint8_t> images((size_t) cap * 8000);
  if (hands_impl(*p, cl, g_h, sample_idx, n_samples, frames.data(), nullptr, out, cap, n_out, nullptr, images.data(), nullptr,
        &pts, &pc, &ofs) != 0)
    return -2;
  ofs.push_back((int64_t) pc.size());
  std::memcpy(ofs_out, ofs.data(), sizeof(int64_t) * ofs.size());
  if ((int64_t) pc.size() <= pts_cap)
  {
    std::memcpy(pts_out, pts.data(), sizeof(double) * pts.size());
    std::memcpy(cam_out, pc.data(), sizeof(int32_t) * pc.size());
  }
  return (int64_t) pc.size();
}

int orc_find_hands_training(const orc_params* p, const float* xyz, int64_t stride_floats, const int32_t* cam, int64_t n,
  const int32_t* sample_idx, int64_t n_samples, orc_hypothesis* out, int64_t cap, int64_t* n_out, uint8_t* images_out,
  uint8_t* cam_images_out)
{
  return find_hands_full(p, xyz, stride_floats, cam, n, sample_idx, n_samples, 1, out, cap, n_out, nullptr, nullptr,
    images_out, cam_images_out);
}

int orc_hands_from_frames(const orc_params* p, const float* xyz, int64_t stride_floats, const int32_t* cam,
  int64_t n, const int32_t* sample_idx, int64_t n_samples, const orc_frame* frames, const double* normals,
  orc_hypothesis* out, int64_t cap, int64_t* n_out, int32_t* nh_out, uint8_t* images_out)
{
  Cloud cl{ xyz, stride_floats, cam, n };
  GridIndex g_h;
  g_h.build(cl, p->nn_radius_hands);
  return hands_impl(*p, cl, g_h, sample_idx, n_samples, frames, normals, out, cap, n_out, nh_out, images_out);
}

int orc_hog(const uint8_t* image, float* desc_out)
{
  hog_compute(image, desc_out);
  return 0;
}

int orc_svm_keep(const float* desc, const float* weights, int32_t n_w, double rho, double* sum_out)
{
  return svm_keep(desc, weights, n_w, rho, sum_out);
}

int orc_classify(const uint8_t* images, int64_t n_hyp, const float* weights, int32_t n_w, double rho,
  uint8_t* keep_out, double* sum_out, int num_threads)
{
  if (n_w != 3528)
    return -1;
  hog_tables();
#pragma omp parallel for num_threads(num_threads) schedule(static) /* learning.cpp:198-201, OMP loop C */
  for (int64_t i = 0; i < n_hyp; i++)
  {
    float desc[3528];
    hog_compute(images + i * 8000, desc);
    double s = 0;
    keep_out[i] = (uint8_t) svm_keep(desc, weights, n_w, rho, &s);
    if (sum_out)
      sum_out[i] = s;
  }
  return 0;
}

int orc_load_svm(const char* path, float* weights_out, int32_t cap, double* rho_out)
{
  FILE* f = std::fopen(path, "rb");
  if (!f)
    return -1;
  std::string txt;
  char buf[4096];
  size_t r;
  while ((r = std::fread(buf, 1, sizeof(buf), f)) > 0)
    txt.append(buf, r);
  std::fclose(f);
  size_t sv = txt.find("support_vectors:");
  size_t df = txt.find("decision_functions:");
  if (sv == std::string::npos || df == std::string::npos)
    return -2;
  size_t lb = txt.find('[', sv);
  size_t rb = txt.find(']', lb);
  if (lb == std::string::npos || rb == std::string::npos || rb > df)
    return -2;
  int32_t n = 0;
  const char* s = txt.c_str() + lb + 1;
  const char* end = txt.c_str() + rb;
  while (s < end)
  {
    char* e2 = nullptr;
    double v = std::strtod(s, &e2);
    if (e2 == s)
    {
      s++;
      continue;
    }
    if (n < cap)
      weights_out[n] = (float) v;
    n++;
    s = e2;
  }
  size_t rp = txt.find("rho:", df);
  if (rp == std::string::npos)
    return -2;
  *rho_out = std::strtod(txt.c_str() + rp + 4, nullptr);
  return n;
}

void orc_glibc_rand(uint32_t seed, int32_t* out, int64_t count)
{
  GlibcRand g(seed);
  for (int64_t i = 0; i < count; i++)
    out[i] = g.next();
}

void orc_smallest_eigvec3(const double* M3, double* axis_out)
{
  double Mm[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      Mm[i][j] = M3[i * 3 + j];
  smallest_eigvec3(Mm, axis_out);
}

int orc_solve_taubin(const double* M, const double* N, double* v_out, double* lambda_out)
{
  double Mm[10][10], Nm[10][10];
  for (int i = 0; i < 10; i++)
    for (int j = 0; j < 10; j++)
    {
      Mm[i][j] = M[i * 10 + j];
      Nm[i][j] = N[i * 10 + j];
    }
  return solve_taubin(Mm, Nm, v_out, lambda_out) ? 0 : -1;
}

// ---------------------------------------------------------------------------------------------------------------
// f1: the preprocessing in front of HandSearch::findHands (/root/reference/src/agile_grasp/localization.cpp)
//   17-24   camera id of point i = (i >= size_left), assigned BEFORE the NaN removal and never re-indexed
//   25-27   pcl::removeNaNFromPointCloud: order-preserving removal of points with a non-finite coordinate; a cloud
//           flagged is_dense is copied as it is (PCL filter.hpp) -- `dense` selects that branch
//   216-245 filterWorkspace: keep min <= p <= max per axis (float promoted to double), camera id taken at the
//           point's position in the NaN-free cloud
//   247-355 voxelizeCloud: per-camera minimum, floor((p - min) / cell) in double, std::set in lexicographic
//           (x, y, z) order (localization.h:273-293), coordinates back as v * cell + 1 * min, stored as float;
//           camera 0 block, then camera 1 block.
int64_t orc_preprocess(const float* xyz, int64_t stride_floats, int64_t n, int64_t size_left, int dense,
  const double workspace[6], double cell_size, float* xyz_out, int32_t* cam_out, int64_t cap)
{
  std::vector<std::array<float, 3>> pts;
  pts.reserve((size_t) n);
  for (int64_t i = 0; i < n; i++)
  {
    const float* p = xyz + i * stride_floats;
    if (dense || (std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2])))
      pts.push_back({ p[0], p[1], p[2] });
  }
  std::vector<std::array<float, 3>> ws;
  std::vector<int> cam;
  for (size_t i = 0; i < pts.size(); i++)
  {
    const std::array<float, 3>& p = pts[i];
    if (p[0] >= workspace[0] && p[0] <= workspace[1] && p[1] >= workspace[2] && p[1] <= workspace[3] &&
        p[2] >= workspace[4] && p[2] <= workspace[5])
    {
      ws.push_back(p);
      cam.push_back((int64_t) i >= size_left ? 1 : 0);
    }
  }
  double mn[2][3] = { { 10000, 10000, 10000 }, { 10000, 10000, 10000 } };
  for (size_t i = 0; i < ws.size(); i++)
    for (int a = 0; a < 3; a++)
      if (ws[i][a] < mn[cam[i]][a])
        mn[cam[i]][a] = ws[i][a];
  std::set<std::array<int, 3>> bins[2];  // std::array compares lexicographically, like UniqueVectorComparator
  for (size_t i = 0; i < ws.size(); i++)
  {
    std::array<int, 3> v;
    for (int a = 0; a < 3; a++)
      v[a] = (int) std::floor(((double) ws[i][a] - mn[cam[i]][a]) / cell_size);
    bins[cam[i]].insert(v);
  }
  int64_t k = 0;
  for (int c = 0; c < 2; c++)
    for (const std::array<int, 3>& v : bins[c])
    {
      if (k < cap)
      {
        for (int a = 0; a < 3; a++)
          xyz_out[3 * k + a] = (float) ((double) v[a] * cell_size + 1.0 * mn[c][a]);
        cam_out[k] = c;
      }
      k++;
    }
  return k;
}

// ---------------------------------------------------------------------------------------------------------------
// f2: HandleSearch::findHandles (/root/reference/src/agile_grasp/handle_search.cpp:4-128) and Handle
// (/root/reference/src/agile_grasp/handle.cpp:3-74), literal, with three places given a defined meaning:
//  * shortenHandle reads inliers[i](2) of a Vector2d (handle_search.cpp:103).  In the reference's build
//    (-DNDEBUG -O3, CMakeLists.txt:18) that is the first component of the NEXT vector element, an inlier index >= 0,
//    so the `< 0` branch is never taken: the list is cut to the elements BEFORE position i (element i itself is
//    dropped, :111) and the loop in findHandles runs once.
//  * std::sort with LastElementComparator leaves the order of equal distances open; here ties are ordered by index.
//  * Eigen::EigenSolver returns the axis with an arbitrary sign; here it agrees with the first inlier's axis.
//    axis_mat * axis_mat^T is an Eigen product: LaneSum64 order, like M3 in fit_frame.
static double safe_acos(double x) /* handle_search.cpp:120-127 */
{
  if (x < -1.0)
    x = -1.0;
  else if (x > 1.0)
    x = 1.0;
  return std::acos(x);
}

int64_t orc_find_handles(const orc_hypothesis* hands, int64_t n_hands, int32_t min_inliers, double min_length,
  orc_handle* handles_out, int64_t handle_cap, int32_t* inlier_idx, int64_t idx_cap)
{
  const int64_t H = n_hands;
  std::vector<double> width((size_t) H);
  for (int64_t i = 0; i < H; i++)
    width[(size_t) i] = hands[i].width;
  int64_t n_handles = 0, n_idx = 0;
  for (int64_t i = 0; i < H; i++)
  {
    if (width[(size_t) i] == -1)
      continue;
    const double* ia = hands[i].axis;
    const double* ip = hands[i].bottom;
    const double* in_ = hands[i].approach;
    std::vector<std::pair<double, int32_t>> inl; /* (dist_along_line, j) */
    for (int64_t j = 0; j < H; j++)
    {
      if (width[(size_t) j] == -1)
        continue;
      const double* ja = hands[j].axis;
      const double* jp = hands[j].bottom;
      const double* jn = hands[j].approach;
      const double d[3] = { jp[0] - ip[0], jp[1] - ip[1], jp[2] - ip[2] };
      double v[3];
      for (int r = 0; r < 3; r++) /* (I - a a^T) d, row by row, left to right */
      {
        const double p0 = ((r == 0) ? 1.0 : 0.0) - ia[r] * ia[0];
        const double p1 = ((r == 1) ? 1.0 : 0.0) - ia[r] * ia[1];
        const double p2 = ((r == 2) ? 1.0 : 0.0) - ia[r] * ia[2];
        v[r] = (p0 * d[0] + p1 * d[1]) + p2 * d[2];
      }
      const double dist_from_line = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
      const double dist_along_line = (ia[0] * d[0] + ia[1] * d[1]) + ia[2] * d[2];
      const double aa = (ia[0] * ja[0] + ia[1] * ja[1]) + ia[2] * ja[2];
      const double ang0 = safe_acos(aa), ang1 = M_PI - safe_acos(aa);
      const double dist_angle_axis = ang1 < ang0 ? ang1 : ang0; /* Eigen minCoeff */
      const double dist_from_normal = safe_acos((in_[0] * jn[0] + in_[1] * jn[1]) + in_[2] * jn[2]);
      if (dist_from_line < 0.01 && dist_angle_axis < 0.34 && dist_from_normal < 0.34)
        inl.push_back(std::make_pair(dist_along_line, (int32_t) j));
    }
    if ((int64_t) inl.size() < min_inliers)
      continue;
    std::sort(inl.begin(), inl.end()); /* (distance, index) */
    for (size_t k = 0; k + 1 < inl.size(); k++)
      if (inl[k + 1].first - inl[k].first > 0.02) /* handle_gap_threshold */
      {
        inl.resize(k);
        break;
      }
    if ((int64_t) inl.size() < min_inliers)
      continue;
    double min_dist = 10000000, max_dist = -10000000;
    for (size_t k = 0; k < inl.size(); k++)
    {
      if (inl[k].first < min_dist)
        min_dist = inl[k].first;
      if (inl[k].first > max_dist)
        max_dist = inl[k].first;
    }
    if (!(max_dist - min_dist > min_length))
      continue;
    /* ---- Handle(hand_list, in) ---- */
    const int n = (int) inl.size();
    orc_handle hd;
    std::memset(&hd, 0, sizeof(hd));
    double M3[3][3];
    for (int r = 0; r < 3; r++)
      for (int q = r; q < 3; q++)
      {
        LaneSum64 acc;
        for (int k = 0; k < n; k++)
          acc.add(k, hands[inl[(size_t) k].second].axis[r] * hands[inl[(size_t) k].second].axis[q]);
        M3[r][q] = acc.total();
        M3[q][r] = M3[r][q];
      }
    double V3[3][3], d3[3];
    jacobi_sym<3>(M3, V3, d3);
    int mx = 0;
    for (int r = 1; r < 3; r++)
      if (d3[r] > d3[mx])
        mx = r; /* maxCoeff: first maximum */
    double axis[3] = { V3[0][mx], V3[1][mx], V3[2][mx] };
    if (dot3(axis, hands[inl[0].second].axis) < 0)
      for (int r = 0; r < 3; r++)
        axis[r] *= -1.0;
    double dmin = 0, dmax = 0;
    std::vector<double> along((size_t) n);
    for (int k = 0; k < n; k++)
    {
      along[(size_t) k] = dot3(axis, hands[inl[(size_t) k].second].bottom);
      if (k == 0 || along[(size_t) k] < dmin)
        dmin = along[(size_t) k];
      if (k == 0 || along[(size_t) k] > dmax)
        dmax = along[(size_t) k];
    }
    const double center_dist = (dmax + dmin) / 2.0;
    double best = 10000000;
    int min_idx = -1;
    for (int k = 0; k < n; k++)
    {
      const double dist = std::fabs(along[(size_t) k] - center_dist);
      if (dist < best)
      {
        best = dist;
        min_idx = k;
      }
    }
    if (min_idx < 0)
      min_idx = 0; /* (the reference would index with -1 if every distance were >= 1e7 or NaN) */
    const orc_hypothesis& c = hands[inl[(size_t) min_idx].second];
    double wsum = 0.0;
    for (int k = 0; k < n; k++)
      wsum += hands[inl[(size_t) k].second].width;
    for (int r = 0; r < 3; r++)
    {
      hd.axis[r] = axis[r];
      hd.center[r] = c.bottom[r];
      hd.approach[r] = c.approach[r];
      hd.hands_center[r] = c.surface[r];
    }
    cross3(c.approach, axis, hd.binormal);
    hd.width = wsum / (double) n;
    hd.n_inliers = n;
    hd.first_inlier = (int32_t) n_idx;
    if (n_handles < handle_cap)
      handles_out[n_handles] = hd;
    for (int k = 0; k < n; k++)
    {
      if (n_idx < idx_cap)
        inlier_idx[n_idx] = inl[(size_t) k].second;
      n_idx++;
      width[(size_t) inl[(size_t) k].second] = -1; /* eliminate from future search (handle_search.cpp:75-78) */
    }
    n_handles++;
  }
  return n_handles;
}


// ---------------------------------------------------------------------------------------------------------------
// f4: the training side, Learning::convertData (/root/reference/src/agile_grasp/learning.cpp:249-318):
//   CvSVMParams{C_SVC, LINEAR} (defaults C = 1, term_crit = {1000 iterations, FLT_EPSILON}); CvSVM::train; CvSVM::save.
// THIRD PARTY -- OpenCV 2.4 modules/ml/src/svm.cpp is not in /root/reference and OpenCV is not in this image, so the
// solver below restates its published algorithm and its parity with OpenCV itself is UNPINNED (no golden vector of a
// training run exists in the reference; the shipped model's training set is not available).  What is pinned: the file
// writer reproduces the reference's shipped model file byte for byte from its (weights, rho), and the solver's result
// satisfies the KKT conditions of the C-SVC dual (tests/test_training.py).
//   CvSVM::do_train            samples sorted by class with the original order kept inside a class
//                              (cvSortSamplesByClasses: qsort on (class, index)); class 0 = label -1 gets y = +1,
//                              class 1 = label +1 gets y = -1; alpha *= y after the solve; support vectors = samples
//                              with |alpha| > 0 in that order
//   CvSVMSolver::solve_c_svc   alpha = 0, b = -1, C_i = C
//   CvSVMSolver::solve_generic gradient init, loop { select_working_set; two kernel rows; clipped pair update;
//                              G[k] += Q_i[k]*d_i + Q_j[k]*d_j }, at most max_iter updates
//   select_working_set         maximal violating pair, strict '>' so the lowest index wins ties; stop when
//                              Gmax1 + Gmax2 < eps
//   CvSVMKernel::calc_linear   float products summed four at a time in float, accumulated in double, stored as float
//                              (Qfloat), clamped to FLT_MAX*1e-3; get_row_svc multiplies by y_i*y_j
//   calc_rho                   mean of y*G over the free alphas, else the midpoint of the bounds
//   CvSVM::optimize_linear_svm v[k] += sv[k]*alpha in double over the support vectors, stored as float; rho unchanged
typedef struct orc_train_info_
{
  int32_t iterations, n_sv, n_class0, n_class1;
  double objective;
} orc_train_info_;

int orc_train_svm(const float* features, const int8_t* labels, int64_t n, int32_t var_count, int32_t kernel, double C,
  int32_t max_iter, double eps, float* weights_out, double* rho_out,
  int32_t* info_out /* iterations, n_sv, n_class0, n_class1 */,
  double* alpha_out /* optional, n entries in the caller's sample order, signed */,
  int32_t* sv_order_out /* optional, n_sv entries: the support vectors' sample indices in model order */, int num_threads)
{
  if (kernel != 0 && kernel != 1)
    return -1;
  if (n <= 0 || var_count <= 0)
    return -1;
  std::vector<int64_t> order;
  order.reserve((size_t) n);
  for (int64_t i = 0; i < n; i++)
    if (labels[i] <= 0)
      order.push_back(i);
  const int64_t n0 = (int64_t) order.size();
  for (int64_t i = 0; i < n; i++)
    if (labels[i] > 0)
      order.push_back(i);
  if (n0 == 0 || n0 == n)
    return -3; /* a two-class problem needs both classes */
  std::vector<signed char> y((size_t) n);
  for (int64_t k = 0; k < n; k++)
    y[(size_t) k] = k < n0 ? 1 : -1;
  std::vector<double> alpha((size_t) n, 0.0), G((size_t) n, -1.0);
  std::vector<signed char> status((size_t) n, -1);
  auto upd = [&](int64_t i) { status[(size_t) i] = alpha[(size_t) i] >= C ? 1 : (alpha[(size_t) i] <= 0 ? -1 : 0); };
  const float max_val = (float) (FLT_MAX * 1e-3);
  // kernel rows, cached (a cache changes nothing in the values)
  std::vector<std::vector<float>> cache((size_t) n);
  size_t cached_rows = 0;
  const size_t max_rows = std::max<size_t>(4, (size_t) (((size_t) 1 << 30) / ((size_t) n * 4)));
  auto get_row = [&](int64_t i) -> const float* {
    std::vector<float>& row = cache[(size_t) i];
    if (!row.empty())
      return row.data();
    if (cached_rows >= max_rows)
    {
      for (auto& r : cache)
        std::vector<float>().swap(r);
      cached_rows = 0;
    }
    row.resize((size_t) n);
    const float* another = features + order[(size_t) i] * var_count;
#pragma omp parallel for num_threads(num_threads) schedule(static)
    for (int64_t j = 0; j < n; j++)
    {
      const float* sample = features + order[(size_t) j] * var_count;
      double s = 0;
      int k = 0;
      for (; k <= var_count - 4; k += 4)
        s += sample[k] * another[k] + sample[k + 1] * another[k + 1] + sample[k + 2] * another[k + 2] +
             sample[k + 3] * another[k + 3];
      for (; k < var_count; k++)
        s += sample[k] * another[k];
      float q = (float) (s * 1.0 + 0.0); /* gamma = 1, coef0 = 0 (CvSVMParams defaults) */
      if (kernel == 1)
        q = q * q; /* calc_poly: cvPow(R, R, degree = 2) = multiply(src, src) in float */
      if (q > max_val)
        q = max_val;
      row[(size_t) j] = y[(size_t) i] > 0 ? y[(size_t) j] * q : -y[(size_t) j] * q;
    }
    cached_rows++;
    return row.data();
  };
  int iter = 0;
  for (;;)
  {
    double Gmax1 = -DBL_MAX, Gmax2 = -DBL_MAX;
    int64_t i1 = -1, i2 = -1;
    for (int64_t i = 0; i < n; i++)
    {
      double t;
      const bool ub = status[(size_t) i] > 0, lb = status[(size_t) i] < 0;
      if (y[(size_t) i] > 0)
      {
        if (!ub && (t = -G[(size_t) i]) > Gmax1)
        {
          Gmax1 = t;
          i1 = i;
        }
        if (!lb && (t = G[(size_t) i]) > Gmax2)
        {
          Gmax2 = t;
          i2 = i;
        }
      }
      else
      {
        if (!ub && (t = -G[(size_t) i]) > Gmax2)
        {
          Gmax2 = t;
          i2 = i;
        }
        if (!lb && (t = G[(size_t) i]) > Gmax1)
        {
          Gmax1 = t;
          i1 = i;
        }
      }
    }
    if (Gmax1 + Gmax2 < eps || iter++ >= max_iter)
      break;
    const int64_t i = i1, j = i2;
    // two rows may evict each other from a tiny cache: copy the first
    std::vector<float> Qi_copy(get_row(i), get_row(i) + n);
    const float* Q_i = Qi_copy.data();
    const float* Q_j = get_row(j);
    const double C_i = C, C_j = C;
    double alpha_i = alpha[(size_t) i], alpha_j = alpha[(size_t) j];
    const double old_i = alpha_i, old_j = alpha_j;
    if (y[(size_t) i] != y[(size_t) j])
    {
      const double denom = Q_i[i] + Q_j[j] + 2 * Q_i[j];
      const double delta = (-G[(size_t) i] - G[(size_t) j]) / std::max(std::fabs(denom), (double) FLT_EPSILON);
      const double diff = alpha_i - alpha_j;
      alpha_i += delta;
      alpha_j += delta;
      if (diff > 0 && alpha_j < 0)
      {
        alpha_j = 0;
        alpha_i = diff;
      }
      else if (diff <= 0 && alpha_i < 0)
      {
        alpha_i = 0;
        alpha_j = -diff;
      }
      if (diff > C_i - C_j && alpha_i > C_i)
      {
        alpha_i = C_i;
        alpha_j = C_i - diff;
      }
      else if (diff <= C_i - C_j && alpha_j > C_j)
      {
        alpha_j = C_j;
        alpha_i = C_j + diff;
      }
    }
    else
    {
      const double denom = Q_i[i] + Q_j[j] - 2 * Q_i[j];
      const double delta = (G[(size_t) i] - G[(size_t) j]) / std::max(std::fabs(denom), (double) FLT_EPSILON);
      const double sum = alpha_i + alpha_j;
      alpha_i -= delta;
      alpha_j += delta;
      if (sum > C_i && alpha_i > C_i)
      {
        alpha_i = C_i;
        alpha_j = sum - C_i;
      }
      else if (sum <= C_i && alpha_j < 0)
      {
        alpha_j = 0;
        alpha_i = sum;
      }
      if (sum > C_j && alpha_j > C_j)
      {
        alpha_j = C_j;
        alpha_i = sum - C_j;
      }
      else if (sum <= C_j && alpha_i < 0)
      {
        alpha_i = 0;
        alpha_j = sum;
      }
    }
    alpha[(size_t) i] = alpha_i;
    alpha[(size_t) j] = alpha_j;
    upd(i);
    upd(j);
    const double d_i = alpha_i - old_i, d_j = alpha_j - old_j;
    for (int64_t k = 0; k < n; k++)
      G[(size_t) k] += Q_i[k] * d_i + Q_j[k] * d_j;
  }
  // calc_rho
  int nr_free = 0;
  double ub = DBL_MAX, lb = -DBL_MAX, sum_free = 0;
  for (int64_t i = 0; i < n; i++)
  {
    const double yG = y[(size_t) i] * G[(size_t) i];
    if (status[(size_t) i] < 0)
    {
      if (y[(size_t) i] > 0)
        ub = std::min(ub, yG);
      else
        lb = std::max(lb, yG);
    }
    else if (status[(size_t) i] > 0)
    {
      if (y[(size_t) i] < 0)
        ub = std::min(ub, yG);
      else
        lb = std::max(lb, yG);
    }
    else
    {
      ++nr_free;
      sum_free += yG;
    }
  }
  const double rho = nr_free > 0 ? sum_free / nr_free : (ub + lb) * 0.5;
  std::vector<double> v((size_t) var_count, 0.0);
  int n_sv = 0;
  for (int64_t k = 0; k < n; k++)
  {
    const double a = alpha[(size_t) k] * y[(size_t) k];
    if (alpha_out)
      alpha_out[order[(size_t) k]] = a;
    if (std::fabs(a) > 0)
    {
      if (sv_order_out)
        sv_order_out[n_sv] = (int32_t) order[(size_t) k];
      n_sv++;
      const float* src = features + order[(size_t) k] * var_count;
      for (int q = 0; q < var_count; q++)
        v[(size_t) q] += src[q] * a;
    }
  }
  for (int q = 0; q < var_count && weights_out; q++) /* optimize_linear_svm (LINEAR models only) */
    weights_out[q] = (float) v[(size_t) q];
  *rho_out = rho;
  if (info_out)
  {
    info_out[0] = iter > max_iter ? max_iter : iter;
    info_out[1] = n_sv;
    info_out[2] = (int32_t) n0;
    info_out[3] = (int32_t) (n - n0);
  }
  return 0;
}

/* CvSVM::save for the compacted linear model, as cv::FileStorage's YAML emitter lays it out (floats "%.8e" / "%d.",
 * doubles "%.16e", flow sequences wrapped when the next item would pass column 71).  Pinned: regenerates the
 * reference's shipped svm_032015_linear_20_20_same byte for byte. */
int orc_save_svm(const char* path, const float* weights, int32_t n_w, double rho)
{
  FILE* f = std::fopen(path, "wb");
  if (!f)
    return -1;
  auto real = [](double v, bool dbl, char* buf) {
    const long iv = std::lrint(v);
    if ((double) iv == v)
      std::snprintf(buf, 64, "%ld.", iv);
    else
      std::snprintf(buf, 64, dbl ? "%.16e" : "%.8e", v);
  };
  std::fprintf(f, "%%YAML:1.0\nmy_svm: !!opencv-ml-svm\n   svm_type: C_SVC\n   kernel: { type:LINEAR }\n   C: 1.\n"
                  "   term_criteria: { epsilon:1.1920928955078125e-07, iterations:1000 }\n   var_all: %d\n"
                  "   var_count: %d\n   class_count: 2\n   class_labels: !!opencv-matrix\n      rows: 1\n      cols: 2\n"
                  "      dt: i\n      data: [ -1, 1 ]\n   sv_total: 1\n   support_vectors:\n",
    n_w, n_w);
  std::string line = "      - [";
  char buf[64];
  for (int k = 0; k < n_w; k++)
  {
    real((double) weights[k], false, buf);
    if (k)
      line += ",";
    const size_t off = line.size() + std::strlen(buf);
    if (off > 71 && off - 10 > 10)
    {
      std::fprintf(f, "%s\n", line.c_str());
      line = std::string(10, ' ') + buf;
    }
    else
      line += std::string(" ") + buf;
  }
  std::fprintf(f, "%s ]\n", line.c_str());
  real(rho, true, buf);
  std::fprintf(f, "   decision_functions:\n      -\n         sv_count: 1\n         rho: %s\n         alpha: [ 1. ]\n"
                  "         index: [ 0 ]\n", buf);
  std::fclose(f);
  return 0;
}


/* CvSVM::save for any of the two model shapes Learning::convertData produces (learning.cpp:296-312): LINEAR (compacted
 * to one support vector by optimize_linear_svm) and POLY degree 2 (uses_linear_kernel = false, the header's default):
 * write_params' kernel map, sv_total support vectors, one decision function with alpha ("%.16e") and index. */
int orc_save_svm_model(const char* path, int32_t kernel, const float* sv, int32_t n_sv, int32_t n_w, const double* alpha,
  double rho)
{
  FILE* f = std::fopen(path, "wb");
  if (!f)
    return -1;
  auto real = [](double v, bool dbl, char* buf) {
    const long iv = std::lrint(v);
    if ((double) iv == v)
      std::snprintf(buf, 64, "%ld.", iv);
    else
      std::snprintf(buf, 64, dbl ? "%.16e" : "%.8e", v);
  };
  std::string out;
  auto flow_seq = [&](const std::string& head, size_t indent, int count, const std::function<void(int, char*)>& item) {
    std::string line = head + "[";
    char buf[64];
    for (int k = 0; k < count; k++)
    {
      item(k, buf);
      if (k)
        line += ",";
      const size_t off = line.size() + std::strlen(buf);
      if (off > 71 && off - indent > 10)
      {
        out += line + "\n";
        line = std::string(indent, ' ') + buf;
      }
      else
        line += std::string(" ") + buf;
    }
    out += line + " ]\n";
  };
  out += "%YAML:1.0\nmy_svm: !!opencv-ml-svm\n   svm_type: C_SVC\n";
  out += kernel == 0 ? "   kernel: { type:LINEAR }\n" : "   kernel: { type:POLY, degree:2., gamma:1., coef0:0. }\n";
  out += "   C: 1.\n   term_criteria: { epsilon:1.1920928955078125e-07, iterations:1000 }\n";
  out += "   var_all: " + std::to_string(n_w) + "\n   var_count: " + std::to_string(n_w) + "\n";
  out += "   class_count: 2\n   class_labels: !!opencv-matrix\n      rows: 1\n      cols: 2\n      dt: i\n"
         "      data: [ -1, 1 ]\n";
  out += "   sv_total: " + std::to_string(n_sv) + "\n   support_vectors:\n";
  for (int v = 0; v < n_sv; v++)
    flow_seq("      - ", 10, n_w, [&](int k, char* buf) { real((double) sv[(size_t) v * n_w + k], false, buf); });
  char rb[64];
  real(rho, true, rb);
  out += "   decision_functions:\n      -\n         sv_count: " + std::to_string(n_sv) + "\n         rho: " + rb + "\n";
  flow_seq("         alpha: ", 13, n_sv, [&](int k, char* buf) { real(alpha[k], true, buf); });
  flow_seq("         index: ", 13, n_sv, [&](int k, char* buf) { std::snprintf(buf, 64, "%d", k); });
  const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
  return (std::fclose(f) == 0 && ok) ? 0 : -1;
}

/* CvSVM::load for those files: returns n_sv (the arrays hold min(n_sv, sv_cap) vectors), <0 on a parse error or an
 * unsupported model (anything but C_SVC with LINEAR, or POLY degree 2 / gamma 1 / coef0 0). */
int orc_load_svm_model(const char* path, int32_t* kernel_out, float* sv_out, int32_t sv_cap, int32_t n_w,
  double* alpha_out, double* rho_out)
{
  FILE* f = std::fopen(path, "rb");
  if (!f)
    return -1;
  std::string txt;
  char buf[1 << 16];
  size_t r;
  while ((r = std::fread(buf, 1, sizeof(buf), f)) > 0)
    txt.append(buf, r);
  std::fclose(f);
  const size_t kp = txt.find("kernel:"), svp = txt.find("support_vectors:"), dfp = txt.find("decision_functions:");
  if (kp == std::string::npos || svp == std::string::npos || dfp == std::string::npos || txt.find("C_SVC") == std::string::npos)
    return -2;
  const std::string kline = txt.substr(kp, txt.find('\n', kp) - kp);
  int kernel;
  if (kline.find("LINEAR") != std::string::npos)
    kernel = 0;
  else if (kline.find("POLY") != std::string::npos)
  {
    auto field = [&](const char* name) {
      const size_t p = kline.find(name);
      return p == std::string::npos ? NAN : std::strtod(kline.c_str() + p + std::strlen(name), nullptr);
    };
    if (field("degree:") != 2.0 || field("gamma:") != 1.0 || field("coef0:") != 0.0)
      return -3;
    kernel = 1;
  }
  else
    return -3;
  *kernel_out = kernel;
  int32_t n_sv = 0;
  size_t pos = svp;
  for (;;)
  {
    const size_t lb = txt.find('[', pos);
    if (lb == std::string::npos || lb > dfp)
      break;
    const size_t rb = txt.find(']', lb);
    if (rb == std::string::npos || rb > dfp)
      return -2;
    const char* s = txt.c_str() + lb + 1;
    const char* end = txt.c_str() + rb;
    int32_t k = 0;
    while (s < end)
    {
      char* e2 = nullptr;
      const double v = std::strtod(s, &e2);
      if (e2 == s)
      {
        s++;
        continue;
      }
      if (n_sv < sv_cap && k < n_w)
        sv_out[(size_t) n_sv * n_w + k] = (float) v;
      k++;
      s = e2;
    }
    if (k != n_w)
      return -2;
    n_sv++;
    pos = rb + 1;
  }
  const size_t rp = txt.find("rho:", dfp), ap = txt.find("alpha:", dfp);
  if (rp == std::string::npos || ap == std::string::npos || n_sv == 0)
    return -2;
  *rho_out = std::strtod(txt.c_str() + rp + 4, nullptr);
  const size_t lb = txt.find('[', ap), rb = txt.find(']', ap);
  if (lb == std::string::npos || rb == std::string::npos)
    return -2;
  const char* s = txt.c_str() + lb + 1;
  const char* end = txt.c_str() + rb;
  int32_t k = 0;
  while (s < end)
  {
    char* e2 = nullptr;
    const double v = std::strtod(s, &e2);
    if (e2 == s)
    {
      s++;
      continue;
    }
    if (k < sv_cap)
      alpha_out[k] = v;
    k++;
    s = e2;
  }
  return k == n_sv ? n_sv : -2;
}

/* CvSVM::predict (C_SVC, two classes) for those models: buffer[k] = K(sv_k, x) as a float (calc_linear / calc_poly with
 * the clamp of CvSVMKernel::calc), sum = -rho + sum_k alpha[k] * buffer[index[k]] in double, class 0 (label -1) iff
 * sum > 0; Learning::classify keeps prediction == 1 (learning.cpp:225-227). */
int orc_classify_model(const uint8_t* images, int64_t n_hyp, int32_t kernel, const float* sv, int32_t n_sv, int32_t n_w,
  const double* alpha, double rho, uint8_t* keep_out, double* sum_out, int num_threads)
{
  if (n_w != 3528)
    return -1;
  hog_tables();
  const float max_val = (float) (FLT_MAX * 1e-3);
#pragma omp parallel for num_threads(num_threads) schedule(static)
  for (int64_t i = 0; i < n_hyp; i++)
  {
    float desc[3528];
    hog_compute(images + i * 8000, desc);
    double sum = -rho;
    for (int32_t v = 0; v < n_sv; v++)
    {
      const float* sample = sv + (size_t) v * n_w;
      double s = 0;
      int k = 0;
      for (; k <= n_w - 4; k += 4)
        s += sample[k] * desc[k] + sample[k + 1] * desc[k + 1] + sample[k + 2] * desc[k + 2] + sample[k + 3] * desc[k + 3];
      for (; k < n_w; k++)
        s += sample[k] * desc[k];
      float q = (float) (s * 1.0 + 0.0);
      if (kernel == 1)
        q = q * q;
      if (q > max_val)
        q = max_val;
      sum += alpha[v] * q;
    }
    keep_out[i] = (uint8_t) ((sum > 0) ? 0 : 1);
    if (sum_out)
      sum_out[i] = sum;
  }
  return 0;
}

} // extern "C"
