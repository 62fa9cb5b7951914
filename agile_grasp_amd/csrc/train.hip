// train.hip -- f4, the training side: Learning::convertData (src/agile_grasp/learning.cpp:249-318).
//
//   image list -> cv::HOGDescriptor::compute (learning.cpp:253-281)    k_hog_svm without a model (hog_svm.hip)
//   CvSVM::train(features, labels, C_SVC, LINEAR)   (296-311)          the kernels below
//   CvSVM::save                                     (312)              agh_save_svm_file
//
// CvSVM::train is OpenCV 2.4 modules/ml/src/svm.cpp (THIRD PARTY, not in the reference tree, not in this image).  Its
// solver is a sequential SMO -- one maximal-violating pair per step, at most term_crit.max_iter = 1000 steps -- whose
// per-step work is data parallel over the n training instances:
//   * select_working_set: two arg-max reductions over the gradient (strict '>': lowest index wins a tie);
//   * two kernel rows Q_i, Q_j: n dot products of 3528 floats each against x_i and x_j (calc_non_rbf_base: float
//     products summed four at a time in float, accumulated in double in index order, stored as float);
//   * G[k] += Q_i[k] * d_alpha_i + Q_j[k] * d_alpha_j.
// k_svm_select (one work-group) does the reductions, the K(i,j) dot product and the clipped two-variable update;
// k_svm_update (one thread per instance) computes its entry of both rows from the transposed feature matrix (so a
// wavefront reads 64 consecutive instances of one feature: coalesced) and updates its gradient entry.  Every float
// and double operation keeps the order of the CPU code, so the model is bit-identical to the test suite's CPU
// restatement of the same algorithm (parity with OpenCV itself is unpinned -- see DESIGN.md).
//
// HBM traffic per step: the transposed features once (n x 3528 x 4 B) + O(n) vectors; the kernel is bound by the
// dependent double-precision accumulation chains (882 links per row and instance), not by bandwidth.
#include "agh_internal.h"

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace agh
{

constexpr int kDesc = 3528;  // 2 windows x 49 blocks x 36
constexpr int kGroups = kDesc / 4;

struct SvmState
{
  int i, j;
  int stop;
  int iter;
  double d_i, d_j;
  double gap;  // Gmax1 + Gmax2 of the last selection
  // kernel-row cache (what CvSVMSolver's row cache does: the values are the same floats, cached or recomputed)
  int slot_i, slot_j;  // where rows Q_i, Q_j live
  int need_i, need_j;  // 1: the step's update kernel computes the row and stores it; 0: it reads it
  int next_slot;       // round-robin victim
  int rows_computed, rows_reused;
};

struct SvmCache
{
  float* rows;       // n_slots x pitch
  int* slot_of_row;  // n, -1 = not cached
  int* row_of_slot;  // n_slots, -1 = free
  int n_slots;
  int64_t pitch;
};

// X (n x kDesc, row-major) -> XT, tiles of 64 instances: XT[(t / 64) * kDesc * 64 + k * 64 + t % 64].  A wavefront's
// 64 instances of feature k are one 256-byte line and consecutive features follow each other: every wave reads one
// sequential 903 KB stream per solver step.
__device__ __forceinline__ int64_t xt_index(int t, int k) { return ((int64_t) (t >> 6) * kDesc + k) * 64 + (t & 63); }

__global__ __launch_bounds__(256) void k_svm_transpose(const float* __restrict__ X, float* __restrict__ XT, int n)
{
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int k0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
  for (int r = ty; r < 32; r += 8)
  {
    const int t = t0 + r, k = k0 + tx;
    tile[r][tx] = (t < n && k < kDesc) ? X[(int64_t) t * kDesc + k] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
  {
    const int k = k0 + r, t = t0 + tx;
    if (k < kDesc && t < n)
      XT[xt_index(t, k)] = tile[tx][r];
  }
}

// CvSVMKernel::calc_non_rbf_base for one pair of vectors: `a` strided (transposed matrix), `b` contiguous.
// poly: CvSVMKernel::calc_poly with convertData's degree = 2 and CvSVMParams' gamma = 1, coef0 = 0: cvPow(R, R, 2) is
// multiply(src, src) in float.
__device__ __forceinline__ float qfloat(double s, int poly)
{
  float q = (float) (s * 1.0 + 0.0);
  if (poly)
    q = q * q;
  const float max_val = (float) (FLT_MAX * 1e-3);
  return q > max_val ? max_val : q;  // CvSVMKernel::calc
}

__global__ __launch_bounds__(256) void k_svm_init(const float* __restrict__ XT, int n, float* __restrict__ Kdiag,
  double* __restrict__ alpha, double* __restrict__ G, int8_t* __restrict__ status, SvmState* __restrict__ st, int poly)
{
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t == 0)
  {
    st->i = st->j = -1;
    st->stop = 0;
    st->iter = 0;
    st->d_i = st->d_j = 0.0;
    st->gap = 0.0;
    st->slot_i = st->slot_j = -1;
    st->need_i = st->need_j = 0;
    st->next_slot = 0;
    st->rows_computed = st->rows_reused = 0;
  }
  if (t >= n)
    return;
  double s = 0;
  for (int g = 0; g < kGroups; g++)
  {
    const float a0 = XT[xt_index(t, 4 * g + 0)], a1 = XT[xt_index(t, 4 * g + 1)];
    const float a2 = XT[xt_index(t, 4 * g + 2)], a3 = XT[xt_index(t, 4 * g + 3)];
    s += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
  }
  Kdiag[t] = qfloat(s, poly);
  alpha[t] = 0.0;  // solve_c_svc: alpha = 0, b = -1  =>  G = b, every alpha at its lower bound
  G[t] = -1.0;
  status[t] = -1;
}

struct ArgMax
{
  double v;
  int idx;
};
__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b)  // larger value; on a tie the lower index (sequential '>')
{
  const bool take_b = b.v > a.v || (b.v == a.v && b.idx >= 0 && (a.idx < 0 || b.idx < a.idx));
  return take_b ? b : a;
}
__device__ __forceinline__ ArgMax wave_argmax(ArgMax a)
{
  for (int off = 32; off >= 1; off >>= 1)
  {
    ArgMax b;
    b.v = __shfl_xor(a.v, off);
    b.idx = __shfl_xor(a.idx, off);
    a = better(a, b);
  }
  return a;
}

// CvSVMSolver::select_working_set + the two-variable update of solve_generic.  y[k] = +1 for class 0 (label -1).
__global__ __launch_bounds__(1024) void k_svm_select(const float* __restrict__ X, int n, const int8_t* __restrict__ y,
  double* __restrict__ alpha, int8_t* __restrict__ status, const double* __restrict__ G, const float* __restrict__ Kdiag,
  SvmState* __restrict__ st, double C, double eps, int max_iter, int poly, SvmCache cache)
{
  __shared__ ArgMax red1[16], red2[16];
  __shared__ int sel[2];
  __shared__ int had[2];  // the rows' cache slots before this step (-1: not cached)
  __shared__ float grp[kGroups];
  if (st->stop)
    return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  ArgMax m1{ -DBL_MAX, -1 }, m2{ -DBL_MAX, -1 };
  for (int t = tid; t < n; t += 1024)
  {
    const double g = G[t];
    const int s = status[t];
    const bool ub = s > 0, lb = s < 0;
    if (y[t] > 0)
    {
      if (!ub && -g > m1.v)
        m1 = ArgMax{ -g, t };
      if (!lb && g > m2.v)
        m2 = ArgMax{ g, t };
    }
    else
    {
      if (!ub && -g > m2.v)
        m2 = ArgMax{ -g, t };
      if (!lb && g > m1.v)
        m1 = ArgMax{ g, t };
    }
  }
  m1 = wave_argmax(m1);
  m2 = wave_argmax(m2);
  if (lane == 0)
  {
    red1[wave] = m1;
    red2[wave] = m2;
  }
  __syncthreads();
  if (tid == 0)
  {
    for (int w = 1; w < 16; w++)
    {
      m1 = better(m1, red1[w]);
      m2 = better(m2, red2[w]);
    }
    int stop = 0;
    if (m1.v + m2.v < eps)
      stop = 1;
    else if (st->iter >= max_iter)  // `select() != 0 || iter++ >= max_iter`
      stop = 1;
    else
      st->iter = st->iter + 1;
    st->gap = m1.v + m2.v;
    if (stop)
      st->stop = 1;
    sel[0] = stop ? -1 : m1.idx;
    sel[1] = stop ? -1 : m2.idx;
    if (!stop && m1.idx >= 0 && m2.idx >= 0)
    {
      // cache bookkeeping: a missing row gets the round-robin victim slot (never the partner's slot)
      const int ri = m1.idx, rj = m2.idx;
      int si = cache.slot_of_row[ri], sj = cache.slot_of_row[rj];
      had[0] = si;
      had[1] = sj;
      auto alloc = [&](int row, int avoid) {
        int v = st->next_slot;
        if (v == avoid)
          v = (v + 1) % cache.n_slots;
        st->next_slot = (v + 1) % cache.n_slots;
        const int old = cache.row_of_slot[v];
        if (old >= 0)
          cache.slot_of_row[old] = -1;
        cache.row_of_slot[v] = row;
        cache.slot_of_row[row] = v;
        return v;
      };
      st->need_i = si < 0 ? 1 : 0;
      st->need_j = sj < 0 ? 1 : 0;
      if (si < 0)
        si = alloc(ri, sj);
      if (sj < 0)
        sj = alloc(rj, si);
      st->slot_i = si;
      st->slot_j = sj;
      st->rows_computed += st->need_i + st->need_j;
      st->rows_reused += 2 - st->need_i - st->need_j;
    }
  }
  __syncthreads();
  const int i = sel[0], j = sel[1];
  if (i < 0 || j < 0)
  {
    if (tid == 0)
      st->stop = 1;
    return;
  }
  const bool kij_cached = had[0] >= 0 || had[1] >= 0;  // Q_i[j] == Q_j[i] bit for bit (the float products commute)
  // K(i, j), calc_non_rbf_base order: 882 float group sums, then one double chain -- unless one of the rows is cached
  const float* xi = X + (int64_t) i * kDesc;
  const float* xj = X + (int64_t) j * kDesc;
  if (!kij_cached && tid < kGroups)
  {
    const float4 a = reinterpret_cast<const float4*>(xj)[tid];  // sample = vecs[j], another = x_i
    const float4 b = reinterpret_cast<const float4*>(xi)[tid];
    grp[tid] = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
  }
  __syncthreads();
  if (tid != 0)
    return;
  const int yi = y[i], yj = y[j];
  float Qij;  // row_i[j]
  if (had[0] >= 0)
    Qij = cache.rows[(int64_t) had[0] * cache.pitch + j];
  else if (had[1] >= 0)
    Qij = cache.rows[(int64_t) had[1] * cache.pitch + i];
  else
  {
    double s = 0;
    for (int g0 = 0; g0 < kGroups; g0 += 14)  // 882 = 63 x 14
    {
      float v[14];
#pragma unroll
      for (int u = 0; u < 14; u++)
        v[u] = grp[g0 + u];
#pragma unroll
      for (int u = 0; u < 14; u++)
        s += v[u];
    }
    const float kij = qfloat(s, poly);
    Qij = yi > 0 ? yj * kij : -yj * kij;
  }
  const float Qii = Kdiag[i], Qjj = Kdiag[j];  // get_row_svc: y_i * y_i = 1
  const double C_i = C, C_j = C;
  double alpha_i = alpha[i], alpha_j = alpha[j];
  const double old_i = alpha_i, old_j = alpha_j;
  const double Gi = G[i], Gj = G[j];
  if (yi != yj)
  {
    const double denom = Qii + Qjj + 2 * Qij;  // float arithmetic, as the Qfloat expression in the reference
    const double delta = (-Gi - Gj) / fmax(fabs(denom), (double) FLT_EPSILON);
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
    const double denom = Qii + Qjj - 2 * Qij;
    const double delta = (Gi - Gj) / fmax(fabs(denom), (double) FLT_EPSILON);
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
  alpha[i] = alpha_i;
  alpha[j] = alpha_j;
  status[i] = alpha_i >= C_i ? 1 : (alpha_i <= 0 ? -1 : 0);
  status[j] = alpha_j >= C_j ? 1 : (alpha_j <= 0 ? -1 : 0);
  st->i = i;
  st->j = j;
  st->d_i = alpha_i - old_i;
  st->d_j = alpha_j - old_j;
}

// Rows Q_i, Q_j of the step's pair (one entry per thread) and the gradient update.
// x_i and x_j sit interleaved in LDS ({x_i[k], x_j[k]} pairs: one packed multiply serves both rows); the instance's
// features come from its tile of the transposed matrix through buffer loads (scalar base, constant offsets), two
// batches of UNROLL groups in ping-pong so that loads are in flight under the two dependent double-precision chains.
// Measured (6231 instances): 184 us per solver step with plain per-lane addressing and two groups in flight, 80 us in
// this form; what remains is the issue time of one wavefront's 882-link chains (a wave per SIMD: the instance count is
// all the parallelism the exact summation order leaves).
// The two chains of one lane: dot products (calc_non_rbf_base order) of the lane's vector -- column `lane` of the tile,
// kDesc lines of 256 bytes -- with the two vectors interleaved in LDS.
template <int UNROLL>
__device__ __forceinline__ void dual_dot(const float* tile /* wave-uniform */, unsigned lane_off, const float2* xij,
  double& si, double& sj)
{
  static_assert(kGroups % (2 * UNROLL) == 0, "two batches per trip");
  typedef float v2f __attribute__((ext_vector_type(2)));
  typedef float v4f __attribute__((ext_vector_type(4)));
  float a[UNROLL][4], b[UNROLL][4];
  auto load = [&](float (&dst)[UNROLL][4], int g0) {
    const __amdgpu_buffer_rsrc_t rows =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tile + (int64_t) g0 * 256), 0, -1, 0x00020000);
#pragma unroll
    for (int u = 0; u < UNROLL; u++)
#pragma unroll
      for (int q = 0; q < 4; q++)
        dst[u][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rows, lane_off, (4 * u + q) * 256, 0));
  };
  auto chain = [&](const float (&src)[UNROLL][4], int g0) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++)
    {
      const v4f b01 = reinterpret_cast<const v4f*>(xij)[2 * (g0 + u)];      // x_i[k], x_j[k], x_i[k+1], x_j[k+1]
      const v4f b23 = reinterpret_cast<const v4f*>(xij)[2 * (g0 + u) + 1];
      v2f p = b01.xy * src[u][0];  // lane .x: row i, lane .y: row j -- packed float math, same order per lane
      p = p + b01.zw * src[u][1];
      p = p + b23.xy * src[u][2];
      p = p + b23.zw * src[u][3];
      si += p.x;
      sj += p.y;
    }
  };
  // two register batches in ping-pong: the loads of one are in flight while the other feeds the chains
  load(a, 0);
  for (int g0 = 0; g0 < kGroups; g0 += 2 * UNROLL)
  {
    load(b, g0 + UNROLL);
    chain(a, g0);
    load(a, g0 + 2 * UNROLL < kGroups ? g0 + 2 * UNROLL : 0);  // (the last trip reloads the first batch: no branch)
    chain(b, g0 + UNROLL);
  }
}

// The tile of the wave that holds vector index `first` (wave-uniform: made visible to the compiler, or every buffer load
// is wrapped in a waterfall loop); lanes past n read the tile's padding and are dropped by the caller.
__device__ __forceinline__ const float* wave_tile(const float* XT, unsigned first, int n)
{
  const int tile_id = __builtin_amdgcn_readfirstlane(min((int) (first >> 6), (n - 1) >> 6));
  return XT + (int64_t) tile_id * kDesc * 64;
}

__global__ __launch_bounds__(256) void k_svm_update(const float* __restrict__ X, const float* __restrict__ XT, int n,
  const int8_t* __restrict__ y, double* __restrict__ G, const SvmState* __restrict__ st, int poly, SvmCache cache)
{
  __shared__ __attribute__((aligned(16))) float2 xij[kDesc];
  if (st->stop)
    return;
  const int i = st->i, j = st->j;
  const double d_i = st->d_i, d_j = st->d_j;
  const int t = blockIdx.x * 256 + threadIdx.x;
  float* row_i = cache.rows + (int64_t) st->slot_i * cache.pitch;
  float* row_j = cache.rows + (int64_t) st->slot_j * cache.pitch;
  float Qi, Qj;
  if (st->need_i || st->need_j)  // (uniform) at least one row is new: both come out of one pass over the features
  {
    for (int k = threadIdx.x; k < kDesc; k += 256)
      xij[k] = make_float2(X[(int64_t) i * kDesc + k], X[(int64_t) j * kDesc + k]);
    __syncthreads();
    double si = 0, sj = 0;
    dual_dot<3>(wave_tile(XT, blockIdx.x * 256u + (threadIdx.x & ~63u), n), (threadIdx.x & 63u) * 4u, xij, si, sj);
    if (t >= n)
      return;
    const float ki = qfloat(si, poly), kj = qfloat(sj, poly);
    const int yt = y[t];
    Qi = y[i] > 0 ? yt * ki : -yt * ki;  // get_row_svc
    Qj = y[j] > 0 ? yt * kj : -yt * kj;
    if (st->need_i)
      row_i[t] = Qi;
    if (st->need_j)
      row_j[t] = Qj;
  }
  else
  {
    if (t >= n)
      return;
    Qi = row_i[t];
    Qj = row_j[t];
  }
  G[t] = G[t] + (Qi * d_i + Qj * d_j);
}

// ---- prediction with a general model (CvSVM::predict, C_SVC, two classes) --------------------------------------
// buffer[h][v] = K(sv_v, desc_h) for two hypotheses per work-group (the pair shares the packed multiplies), then
// sum_h = -rho + sum_v alpha[v] * buffer[h][v] in double, index order.
__global__ __launch_bounds__(256) void k_svm_kvals(const float* __restrict__ desc, const int64_t* __restrict__ n_hyp,
  const float* __restrict__ SVT, int n_sv, int poly, float* __restrict__ kbuf, int64_t h_base, int64_t cap)
{
  __shared__ __attribute__((aligned(16))) float2 xij[kDesc];
  const int64_t H = n_hyp ? (*n_hyp < cap ? *n_hyp : cap) : cap;  // never past the buffers, whatever the device count says
  const int64_t h0 = h_base + (int64_t) blockIdx.y * 2;
  if (h0 >= H)
    return;
  const int64_t h1 = h0 + 1 < H ? h0 + 1 : h0;
  for (int k = threadIdx.x; k < kDesc; k += 256)
    xij[k] = make_float2(desc[h0 * kDesc + k], desc[h1 * kDesc + k]);
  __syncthreads();
  const int v = blockIdx.x * 256 + threadIdx.x;
  double s0 = 0, s1 = 0;
  dual_dot<3>(wave_tile(SVT, blockIdx.x * 256u + (threadIdx.x & ~63u), n_sv), (threadIdx.x & 63u) * 4u, xij, s0, s1);
  if (v >= n_sv)
    return;
  kbuf[h0 * n_sv + v] = qfloat(s0, poly);
  if (h1 != h0)
    kbuf[h1 * n_sv + v] = qfloat(s1, poly);
}

__global__ __launch_bounds__(64) void k_svm_decide(const float* __restrict__ kbuf, const int64_t* __restrict__ n_hyp, int n_sv,
  const double* __restrict__ alpha, double rho, agh_hypothesis* __restrict__ out, uint8_t* __restrict__ keep,
  double* __restrict__ sums, int64_t cap)
{
  const int64_t h = (int64_t) blockIdx.x * 64 + threadIdx.x;
  if (h >= cap || (n_hyp && h >= *n_hyp))
    return;
  double sum = -rho;
  const float* row = kbuf + h * n_sv;
  int v = 0;
  for (; v + 16 <= n_sv; v += 16)  // loads in flight, the chain in index order
  {
    float q[16];
#pragma unroll
    for (int u = 0; u < 16; u++)
      q[u] = row[v + u];
#pragma unroll
    for (int u = 0; u < 16; u++)
      sum += alpha[v + u] * q[u];
  }
  for (; v < n_sv; v++)
    sum += alpha[v] * row[v];
  const uint8_t k = (sum > 0) ? 0 : 1;  // class_labels[sum > 0 ? 0 : 1] = {-1, +1}; the reference keeps prediction == 1
  if (keep)
    keep[h] = k;
  if (sums)
    sums[h] = sum;
  if (out)
    out[h].svm_keep = k;
}

__global__ __launch_bounds__(256) void k_svm_gather(const float* __restrict__ X, const int32_t* __restrict__ rows, int n_rows,
  float* __restrict__ out)
{
  const int64_t e = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t) n_rows * kDesc)
    return;
  out[e] = X[(int64_t) rows[e / kDesc] * kDesc + e % kDesc];
}

// CvSVM::optimize_linear_svm: v[k] = sum over the support vectors, in order, of sv[k] * alpha (double), stored as float.
__global__ __launch_bounds__(256) void k_svm_compress(const float* __restrict__ X, int n, const double* __restrict__ a_signed,
  float* __restrict__ w)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= kDesc)
    return;
  double v = 0;
  for (int t = 0; t < n; t++)
  {
    const double a = a_signed[t];
    if (fabs(a) > 0)
      v += X[(int64_t) t * kDesc + k] * a;
  }
  w[k] = (float) v;
}

// images (n_images x kImageWords, packed) + order (instance k of the solver = image order[k]; class 0 first) -> model.
// Outputs: LINEAR -- sv_out = the compacted vector, alpha_out[0] = 1, *n_sv_out = 1 (info_out[1] keeps the solver's
// support-vector count); POLY -- the support vectors' descriptors in model order, their signed alphas, their count.
int svm_train(Ctx* c, const uint32_t* h_images, int64_t n_images, const int32_t* h_order, const int8_t* h_y, int64_t n,
  int poly, double C, int max_iter, double eps, float* sv_out, int64_t sv_cap, double* alpha_out, int32_t* n_sv_out,
  double* rho_out, int32_t* info_out, hipStream_t st)
{
  int32_t* d_rows = nullptr;
  float* d_svrows = nullptr;
  SvmCache cache{ nullptr, nullptr, nullptr, 0, 0 };
  uint32_t* d_img = nullptr;
  int32_t* d_ord = nullptr;
  int8_t *d_y = nullptr, *d_status = nullptr;
  float *d_X = nullptr, *d_XT = nullptr, *d_diag = nullptr, *d_w = nullptr;
  double *d_alpha = nullptr, *d_G = nullptr, *d_as = nullptr;
  SvmState* d_st = nullptr;
  int rc = AGH_OK;
  auto fail = [&](const char* what, hipError_t e) {
    c->err = std::string("agh_train_svm: ") + what + ": " + hipGetErrorString(e);
    rc = AGH_ERR_HIP;
  };
#define TRY(expr)                     \
  do                                  \
  {                                   \
    hipError_t e__ = (expr);          \
    if (rc == AGH_OK && e__ != hipSuccess) \
      fail(#expr, e__);               \
  } while (0)
  TRY(hipMalloc((void**) &d_img, (size_t) n_images * kImageWords * 4));
  TRY(hipMalloc((void**) &d_ord, (size_t) n * 4));
  TRY(hipMalloc((void**) &d_y, (size_t) n));
  TRY(hipMalloc((void**) &d_status, (size_t) n));
  TRY(hipMalloc((void**) &d_X, (size_t) n * kDesc * 4));
  TRY(hipMalloc((void**) &d_XT, (size_t) ((n + 63) / 64) * 64 * kDesc * 4));
  TRY(hipMalloc((void**) &d_diag, (size_t) n * 4));
  TRY(hipMalloc((void**) &d_w, (size_t) kDesc * 4));
  TRY(hipMalloc((void**) &d_alpha, (size_t) n * 8));
  TRY(hipMalloc((void**) &d_G, (size_t) n * 8));
  TRY(hipMalloc((void**) &d_as, (size_t) n * 8));
  TRY(hipMalloc((void**) &d_st, sizeof(SvmState)));
  // kernel-row cache: every row if that fits 16 GiB, else as many as do (at least the step's two)
  cache.pitch = (n + 63) / 64 * 64;
  cache.n_slots = (int) std::max<int64_t>(2, std::min<int64_t>(n, ((int64_t) 16 << 30) / (cache.pitch * 4)));
  if (const char* e = std::getenv("AGH_SVM_CACHE_ROWS"))  // testing aid: force evictions with a small cache
    cache.n_slots = (int) std::max<int64_t>(2, std::min<int64_t>(cache.n_slots, std::atoll(e)));
  TRY(hipMalloc((void**) &cache.rows, (size_t) cache.n_slots * cache.pitch * 4));
  TRY(hipMalloc((void**) &cache.slot_of_row, (size_t) n * 4));
  TRY(hipMalloc((void**) &cache.row_of_slot, (size_t) cache.n_slots * 4));
  std::vector<double> alpha((size_t) n), G((size_t) n), a_signed((size_t) n);
  std::vector<int8_t> status((size_t) n);
  SvmState hs;
  std::memset(&hs, 0, sizeof(hs));
  if (rc == AGH_OK)
  {
    TRY(hipMemcpyAsync(d_img, h_images, (size_t) n_images * kImageWords * 4, hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(d_ord, h_order, (size_t) n * 4, hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(d_y, h_y, (size_t) n, hipMemcpyHostToDevice, st));
    TRY(hipMemsetAsync(cache.slot_of_row, 0xff, (size_t) n * 4, st));
    TRY(hipMemsetAsync(cache.row_of_slot, 0xff, (size_t) cache.n_slots * 4, st));
    if (rc == AGH_OK)
      rc = hog_images(c, d_img, d_ord, n, d_X, st);
    const int ni = (int) n;
    hipLaunchKernelGGL(k_svm_transpose, dim3((kDesc + 31) / 32, (unsigned) ((n + 31) / 32)), dim3(256), 0, st, d_X, d_XT, ni);
    hipLaunchKernelGGL(k_svm_init, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, d_XT, ni, d_diag, d_alpha, d_G,
      d_status, d_st, poly);
    const int batch = 50;  // steps between looks at the stop flag (a finished solve turns the rest into empty launches)
    for (int done = 0; rc == AGH_OK && done <= max_iter; done += batch)
    {
      for (int b = 0; b < batch; b++)
      {
        hipLaunchKernelGGL(k_svm_select, dim3(1), dim3(1024), 0, st, d_X, ni, d_y, d_alpha, d_status, d_G, d_diag, d_st, C,
          eps, max_iter, poly, cache);
        hipLaunchKernelGGL(k_svm_update, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, d_X, d_XT, ni, d_y, d_G, d_st, poly,
          cache);
      }
      TRY(hipMemcpyAsync(&hs, d_st, sizeof(hs), hipMemcpyDeviceToHost, st));
      TRY(hipStreamSynchronize(st));
      if (hs.stop)
        break;
    }
    if (rc == AGH_OK && !hs.stop)
    {
      c->err = "agh_train_svm: the solver did not reach its stop state";
      rc = AGH_ERR_STATE;
    }
  }
  if (rc == AGH_OK)
  {
    TRY(hipMemcpyAsync(alpha.data(), d_alpha, (size_t) n * 8, hipMemcpyDeviceToHost, st));
    TRY(hipMemcpyAsync(G.data(), d_G, (size_t) n * 8, hipMemcpyDeviceToHost, st));
    TRY(hipMemcpyAsync(status.data(), d_status, (size_t) n, hipMemcpyDeviceToHost, st));
    TRY(hipStreamSynchronize(st));
  }
  if (rc == AGH_OK)
  {
    // CvSVMSolver::calc_rho (host: one pass over n values)
    int nr_free = 0, n_sv = 0;
    double ub = DBL_MAX, lb = -DBL_MAX, sum_free = 0;
    for (int64_t k = 0; k < n; k++)
    {
      const double yG = h_y[k] * G[(size_t) k];
      if (status[(size_t) k] < 0)
      {
        if (h_y[k] > 0)
          ub = std::fmin(ub, yG);
        else
          lb = std::fmax(lb, yG);
      }
      else if (status[(size_t) k] > 0)
      {
        if (h_y[k] < 0)
          ub = std::fmin(ub, yG);
        else
          lb = std::fmax(lb, yG);
      }
      else
      {
        ++nr_free;
        sum_free += yG;
      }
      a_signed[(size_t) k] = alpha[(size_t) k] * h_y[k];  // do_train: alpha *= y
      n_sv += std::fabs(a_signed[(size_t) k]) > 0 ? 1 : 0;
    }
    *rho_out = nr_free > 0 ? sum_free / nr_free : (ub + lb) * 0.5;
    if (info_out)
    {
      info_out[0] = hs.iter;
      info_out[1] = n_sv;
      info_out[2] = hs.rows_computed;
      info_out[3] = hs.rows_reused;
    }
    if (!poly)
    {
      TRY(hipMemcpyAsync(d_as, a_signed.data(), (size_t) n * 8, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_svm_compress, dim3((kDesc + 255) / 256), dim3(256), 0, st, d_X, (int) n, d_as, d_w);
      TRY(hipMemcpyAsync(sv_out, d_w, (size_t) kDesc * 4, hipMemcpyDeviceToHost, st));
      alpha_out[0] = 1.0;
      *n_sv_out = 1;
    }
    else
    {
      // do_train: the support vectors are the samples with |alpha| > 0, in the class-sorted order; index = 0 .. n_sv-1
      *n_sv_out = n_sv;
      if (n_sv > sv_cap)
      {
        c->err = "agh_train_svm: " + std::to_string(n_sv) + " support vectors, room for " + std::to_string(sv_cap);
        rc = AGH_ERR_CAPACITY;
      }
      else if (n_sv > 0)
      {
        std::vector<int32_t> rows;
        for (int64_t k = 0; k < n; k++)
          if (std::fabs(a_signed[(size_t) k]) > 0)
          {
            alpha_out[rows.size()] = a_signed[(size_t) k];
            rows.push_back((int32_t) k);
          }
        TRY(hipMalloc((void**) &d_rows, rows.size() * 4));
        TRY(hipMalloc((void**) &d_svrows, rows.size() * kDesc * 4));
        TRY(hipMemcpyAsync(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, st));
        if (rc == AGH_OK)
          hipLaunchKernelGGL(k_svm_gather, dim3((unsigned) ((rows.size() * kDesc + 255) / 256)), dim3(256), 0, st, d_X, d_rows,
            (int) rows.size(), d_svrows);
        TRY(hipMemcpyAsync(sv_out, d_svrows, rows.size() * kDesc * 4, hipMemcpyDeviceToHost, st));
      }
    }
    TRY(hipStreamSynchronize(st));
    TRY(hipGetLastError());
  }
#undef TRY
  for (void* p : { (void*) d_img, (void*) d_ord, (void*) d_y, (void*) d_status, (void*) d_X, (void*) d_XT, (void*) d_diag,
         (void*) d_w, (void*) d_alpha, (void*) d_G, (void*) d_as, (void*) d_st, (void*) d_rows, (void*) d_svrows, (void*) cache.rows, (void*) cache.slot_of_row,
         (void*) cache.row_of_slot })
    if (p)
      (void) hipFree(p);
  return rc;
}

// CvSVM::predict with a model that is not the compacted linear vector (Learning::classify, learning.cpp:220-225).
int svm_predict_general(Ctx* c, const float* d_desc, int64_t cap, uint8_t* d_keep, hipStream_t st)
{
  return svm_predict_desc(c, d_desc, cap, c->d_nout_last, c->d_out_last, d_keep, c->d_svm_sums, st);
}

// The same for `cap` descriptors that belong to no search (agh_classify_images): exact count, no records to mark.
int svm_predict_images(Ctx* c, const float* d_desc, int64_t n, uint8_t* d_keep, double* d_sums, hipStream_t st)
{
  return svm_predict_desc(c, d_desc, n, nullptr, nullptr, d_keep, d_sums, st);
}

int svm_predict_desc(Ctx* c, const float* d_desc, int64_t cap, const int64_t* d_nhyp, agh_hypothesis* d_out, uint8_t* d_keep,
  double* d_sums, hipStream_t st)
{
  if (cap <= 0)
    return AGH_OK;
  const int n_sv = c->svm_n_sv;
  if ((int64_t) cap * n_sv > c->cls_kbuf_cap)
  {
    if (c->d_cls_kbuf)
      (void) hipFree(c->d_cls_kbuf);
    c->d_cls_kbuf = nullptr;
    c->cls_kbuf_cap = 0;
    if (hipMalloc((void**) &c->d_cls_kbuf, (size_t) cap * n_sv * 4) != hipSuccess)
    {
      c->err = "agh_classify: out of device memory for the kernel-value buffer";
      return AGH_ERR_HIP;
    }
    c->cls_kbuf_cap = (int64_t) cap * n_sv;
  }
  for (int64_t h_base = 0; h_base < cap; h_base += 2 * 65535)  // (gridDim.y <= 65535: two hypotheses per work-group)
  {
    const int64_t pairs = std::min<int64_t>(65535, (cap - h_base + 1) / 2);
    hipLaunchKernelGGL(k_svm_kvals, dim3((unsigned) ((n_sv + 255) / 256), (unsigned) pairs), dim3(256), 0, st, d_desc,
      d_nhyp, (const float*) c->d_svm_svT, n_sv, c->svm_kernel == AGH_SVM_POLY2 ? 1 : 0, c->d_cls_kbuf, h_base, cap);
  }
  hipLaunchKernelGGL(k_svm_decide, dim3((unsigned) ((cap + 63) / 64)), dim3(64), 0, st, (const float*) c->d_cls_kbuf,
    d_nhyp, n_sv, (const double*) c->d_svm_alpha, c->svm_rho, d_out, d_keep, d_sums, cap);
  return hipGetLastError() == hipSuccess ? AGH_OK : AGH_ERR_HIP;
}

int svm_load_general(Ctx* c, int kernel_type, const float* sv, int n_sv, const double* alpha, double rho)
{
  float* d_sv = nullptr;
  (void) hipDeviceSynchronize();  // classify work on a caller's stream may still read the old model
  for (void** p : { (void**) &c->d_svm_svT, (void**) &c->d_svm_alpha })
    if (*p)
    {
      (void) hipFree(*p);
      *p = nullptr;
    }
  c->has_svm = false;
  const size_t padded = (size_t) ((n_sv + 63) / 64) * 64;
  if (hipMalloc((void**) &d_sv, (size_t) n_sv * kDesc * 4) != hipSuccess ||
      hipMalloc((void**) &c->d_svm_svT, padded * kDesc * 4) != hipSuccess ||
      hipMalloc((void**) &c->d_svm_alpha, (size_t) n_sv * 8) != hipSuccess ||
      hipMemcpyAsync(d_sv, sv, (size_t) n_sv * kDesc * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
      hipMemcpyAsync(c->d_svm_alpha, alpha, (size_t) n_sv * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
      hipMemsetAsync(c->d_svm_svT, 0, padded * kDesc * 4, c->stream) != hipSuccess)
  {
    if (d_sv)
      (void) hipFree(d_sv);
    c->err = "agh_load_svm_model: device allocation or copy failed";
    return AGH_ERR_HIP;
  }
  hipLaunchKernelGGL(k_svm_transpose, dim3((kDesc + 31) / 32, (unsigned) ((n_sv + 31) / 32)), dim3(256), 0, c->stream, d_sv,
    c->d_svm_svT, n_sv);
  const bool ok = hipStreamSynchronize(c->stream) == hipSuccess && hipGetLastError() == hipSuccess;
  (void) hipFree(d_sv);
  if (!ok)
  {
    c->err = "agh_load_svm_model: transpose failed";
    return AGH_ERR_HIP;
  }
  c->svm_kernel = kernel_type;
  c->svm_n_sv = n_sv;
  c->svm_rho = rho;
  c->svm_general = true;
  c->has_svm = true;
  return AGH_OK;
}

}  // namespace agh

// ---- C ABI -----------------------------------------------------------------------------------------------------
using namespace agh;

extern "C" {

int agh_set_training_images(agh_ctx* ctx, int on)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  c->training_images = on != 0;
  if (!on)
    return AGH_OK;
  if (hipSetDevice(c->device) != hipSuccess)
    return AGH_ERR_HIP;
  if (c->s_cap > c->images_cam_cap)  // (otherwise allocated with the per-call buffers)
  {
    if (c->d_images_cam)
      (void) hipFree(c->d_images_cam);
    c->d_images_cam = nullptr;
    c->images_cam_cap = 0;
    if (hipMalloc((void**) &c->d_images_cam, (size_t) c->s_cap * 16 * kImageWords * 4) != hipSuccess)
    {
      c->err = "agh_set_training_images: out of device memory";
      return AGH_ERR_HIP;
    }
    c->images_cam_cap = c->s_cap;
  }
  return AGH_OK;
}

int agh_get_training_images(agh_ctx* ctx, uint32_t* images, int64_t cap_hyp)
{
  if (!ctx || !images)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (c->last_nout < 0 || !c->last_has_cam_images)
  {
    c->err = "agh_get_training_images: needs agh_set_training_images(1) and a completed "
             "agh_find_hands(calculates_antipodal = 1) call";
    return AGH_ERR_STATE;
  }
  const int64_t n = std::min<int64_t>(cap_hyp, c->last_nout);
  if (n == 0)
    return 0;
  if (hipSetDevice(c->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
    return AGH_ERR_HIP;
  std::vector<int32_t> slot((size_t) n);
  std::vector<uint32_t> all((size_t) c->last_s * 8 * kImageWords), cam((size_t) c->last_s * 16 * kImageWords);
  if (hipMemcpy(slot.data(), c->d_slot_index, sizeof(int32_t) * n, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(all.data(), c->d_images, all.size() * 4, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(cam.data(), c->d_images_cam, cam.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
  {
    c->err = "agh_get_training_images: copy failed";
    return AGH_ERR_HIP;
  }
  for (int64_t h = 0; h < n; h++)  // instance order of Learning::train: cam = -1, 0, 1 (learning.cpp:93-97)
  {
    uint32_t* dst = images + h * 3 * kImageWords;
    std::memcpy(dst, &all[(size_t) slot[h] * kImageWords], kImageWords * 4);
    std::memcpy(dst + kImageWords, &cam[(size_t) slot[h] * 2 * kImageWords], 2 * kImageWords * 4);
  }
  return (int) n;
}

int agh_hog_images(agh_ctx* ctx, const uint32_t* images, int64_t n, float* desc)
{
  if (!ctx || n < 0 || (n > 0 && (!images || !desc)))
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (n == 0)
    return AGH_OK;
  if (hipSetDevice(c->device) != hipSuccess)
    return AGH_ERR_HIP;
  uint32_t* d_img = nullptr;
  float* d_desc = nullptr;
  int rc = AGH_OK;
  if (hipMalloc((void**) &d_img, (size_t) n * kImageWords * 4) != hipSuccess ||
      hipMalloc((void**) &d_desc, (size_t) n * kDesc * 4) != hipSuccess)
    rc = AGH_ERR_HIP;
  if (rc == AGH_OK && hipMemcpyAsync(d_img, images, (size_t) n * kImageWords * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess)
    rc = AGH_ERR_HIP;
  if (rc == AGH_OK)
    rc = hog_images(c, d_img, nullptr, n, d_desc, c->stream);
  if (rc == AGH_OK && (hipMemcpyAsync(desc, d_desc, (size_t) n * kDesc * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                        hipStreamSynchronize(c->stream) != hipSuccess))
    rc = AGH_ERR_HIP;
  if (rc != AGH_OK)
    c->err = "agh_hog_images: device allocation, copy or launch failed";
  if (d_img)
    (void) hipFree(d_img);
  if (d_desc)
    (void) hipFree(d_desc);
  return rc;
}

int agh_train_svm(agh_ctx* ctx, const uint32_t* images, const int8_t* labels, int64_t n, int32_t kernel_type, double C,
  int32_t max_iter, double eps, float* sv_out, int64_t sv_cap, double* alpha_out, int32_t* n_sv_out, double* rho_out,
  int32_t* info_out)
{
  if (!ctx || !images || !labels || !sv_out || !alpha_out || !n_sv_out || !rho_out || n <= 0 || n > 2000000 ||
      !(C > 0) || max_iter < 0 || sv_cap < 1 || (kernel_type != AGH_SVM_LINEAR && kernel_type != AGH_SVM_POLY2))
  {
    if (ctx)
      ctx->c.err = "agh_train_svm: need images, labels, 0 < n <= 2000000, a supported kernel, C > 0, max_iter >= 0, "
                   "sv_cap >= 1 and the outputs";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  Ctx* c = &ctx->c;
  // cvSortSamplesByClasses: class 0 (label -1, y = +1) first, original order inside a class
  std::vector<int32_t> order;
  order.reserve((size_t) n);
  for (int64_t k = 0; k < n; k++)
    if (labels[k] <= 0)
      order.push_back((int32_t) k);
  const int64_t n0 = (int64_t) order.size();
  for (int64_t k = 0; k < n; k++)
    if (labels[k] > 0)
      order.push_back((int32_t) k);
  if (n0 == 0 || n0 == n)
  {
    c->err = "agh_train_svm: the training set holds a single class";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  std::vector<int8_t> y((size_t) n);
  for (int64_t k = 0; k < n; k++)
    y[(size_t) k] = k < n0 ? 1 : -1;
  if (hipSetDevice(c->device) != hipSuccess)
    return AGH_ERR_HIP;
  int32_t info[4] = { 0, 0, 0, 0 };
  const int rc = svm_train(c, images, n, order.data(), y.data(), n, kernel_type == AGH_SVM_POLY2 ? 1 : 0, C, max_iter, eps,
    sv_out, sv_cap, alpha_out, n_sv_out, rho_out, info, c->stream);
  if ((rc == AGH_OK || rc == AGH_ERR_CAPACITY) && info_out)
  {
    info_out[0] = info[0];
    info_out[1] = info[1];
    info_out[2] = (int32_t) n0;
    info_out[3] = (int32_t) (n - n0);
    info_out[4] = info[2];
    info_out[5] = info[3];
  }
  return rc;
}

int agh_load_svm_model(agh_ctx* ctx, int32_t kernel_type, const float* sv, int32_t n_sv, int32_t n_weights,
  const double* alpha, double rho)
{
  if (!ctx || !sv || !alpha || n_sv < 1 || (kernel_type != AGH_SVM_LINEAR && kernel_type != AGH_SVM_POLY2))
  {
    if (ctx)
      ctx->c.err = "agh_load_svm_model: need support vectors, alphas and a supported kernel (LINEAR, POLY degree 2)";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  Ctx* c = &ctx->c;
  if (n_weights != kDesc)
  {
    c->err = "agh_load_svm_model: the HOG descriptor has 3528 entries (2 windows x 49 blocks x 36)";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  if (kernel_type == AGH_SVM_LINEAR && n_sv == 1 && alpha[0] == 1.0)  // the compacted form: fused into the HOG kernel
    return agh_load_svm(ctx, sv, n_weights, rho);
  if (hipSetDevice(c->device) != hipSuccess)
    return AGH_ERR_HIP;
  return svm_load_general(c, kernel_type, sv, n_sv, alpha, rho);
}

// CvSVM::save (learning.cpp:312) laid out as cv::FileStorage's YAML emitter does (reals "%.8e" / "%.16e", integer-valued
// reals "%d.", flow sequences wrapped before column 72): the reference's shipped model file is reproduced byte for byte
// from its weights and rho (tests/test_training.py).  Two shapes: the compacted LINEAR vector, or the n_sv support vectors
// of a POLY degree-2 model with their alphas (convertData's uses_linear_kernel = false).
int agh_save_svm_file(const char* path, int32_t kernel_type, const float* sv, int32_t n_sv, int32_t n_weights,
  const double* alpha, double rho)
{
  // CvSVMParams' defaults, what Learning::convertData trains with (learning.cpp:297)
  return agh_save_svm_file_ex(path, kernel_type, sv, n_sv, n_weights, alpha, rho, 1.0, 1000, 1.1920928955078125e-07);
}

int agh_save_svm_file_ex(const char* path, int32_t kernel_type, const float* sv, int32_t n_sv, int32_t n_weights,
  const double* alpha, double rho, double C, int32_t max_iter, double eps)
{
  if (!path || !sv || !alpha || n_sv < 1 || n_weights <= 0 || (kernel_type != AGH_SVM_LINEAR && kernel_type != AGH_SVM_POLY2))
    return AGH_ERR_INVALID_ARGUMENT;
  FILE* f = std::fopen(path, "wb");
  if (!f)
    return AGH_ERR_IO;
  auto real = [](double v, const char* fmt, char* buf) {
    const long iv = std::lrint(v);
    if ((double) iv == v)
      std::snprintf(buf, 64, "%ld.", iv);
    else
      std::snprintf(buf, 64, fmt, v);
  };
  std::string txt;
  const size_t margin = 71;
  char buf[64];
  auto begin_seq = [](const std::string& head) { return head + "["; };
  auto add_item = [&](std::string& line, bool first, size_t indent) {  // icvYMLWrite: wrap when the item would pass the margin
    if (!first)
      line += ',';
    const size_t off = line.size() + std::strlen(buf);
    if (off > margin && off - indent > 10)
    {
      txt += line + "\n";
      line = std::string(indent, ' ') + buf;
    }
    else
    {
      line += ' ';
      line += buf;
    }
  };
  txt += "%YAML:1.0\nmy_svm: !!opencv-ml-svm\n   svm_type: C_SVC\n";
  txt += kernel_type == AGH_SVM_LINEAR ? "   kernel: { type:LINEAR }\n" : "   kernel: { type:POLY, degree:2., gamma:1., coef0:0. }\n";
  real(C, "%.16e", buf);  // the parameters the model was trained with (CvSVM::write_params)
  txt += std::string("   C: ") + buf + "\n";
  real(eps, "%.16e", buf);
  txt += std::string("   term_criteria: { epsilon:") + buf + ", iterations:" + std::to_string(max_iter) + " }\n";
  txt += "   var_all: " + std::to_string(n_weights) + "\n   var_count: " + std::to_string(n_weights) + "\n";
  txt += "   class_count: 2\n   class_labels: !!opencv-matrix\n      rows: 1\n      cols: 2\n      dt: i\n"
         "      data: [ -1, 1 ]\n   sv_total: " + std::to_string(n_sv) + "\n   support_vectors:\n";
  for (int v = 0; v < n_sv; v++)
  {
    std::string line = begin_seq("      - ");
    for (int k = 0; k < n_weights; k++)
    {
      real((double) sv[(size_t) v * n_weights + k], "%.8e", buf);
      add_item(line, k == 0, 10);
    }
    txt += line + " ]\n";
  }
  real(rho, "%.16e", buf);
  txt += "   decision_functions:\n      -\n         sv_count: " + std::to_string(n_sv) + "\n         rho: " + buf + "\n";
  std::string line = begin_seq("         alpha: ");
  for (int k = 0; k < n_sv; k++)
  {
    real(alpha[k], "%.16e", buf);
    add_item(line, k == 0, 13);
  }
  txt += line + " ]\n";
  line = begin_seq("         index: ");
  for (int k = 0; k < n_sv; k++)
  {
    std::snprintf(buf, sizeof(buf), "%d", k);
    add_item(line, k == 0, 13);
  }
  txt += line + " ]\n";
  const bool ok = std::fwrite(txt.data(), 1, txt.size(), f) == txt.size();
  return (std::fclose(f) == 0 && ok) ? AGH_OK : AGH_ERR_IO;
}

}  // extern "C"
