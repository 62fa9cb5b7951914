// grid.hip -- K0: uniform-grid build (bounding box, cell histogram, scan, cell-major scatter).
//
// Stands in for pcl::KdTreeFLANN::setInputCloud (reference src/agile_grasp/hand_search.cpp:10-11): the search
// structure behind the three radius searches of the hot path.  The cloud is re-laid out cell-major as
// float4 {x, y, z, bits((index << 1) | cam)} so that a ball query is a handful of contiguous runs read with
// coalesced 16-byte loads.  Everything runs on the device without a host round trip: the grid descriptor
// (origin, cell size, dimensions) lives in device memory and later kernels read it from there.
#include "agh_internal.h"

namespace agh
{

__global__ void k_desc_init(GridDesc* d)
{
  for (int a = 0; a < 3; a++)
  {
    d->bbox[a] = 0xffffffffu;  // min
    d->bbox[3 + a] = 0u;       // max
  }
}

__global__ __launch_bounds__(256) void k_bbox(const float* __restrict__ xyz, int64_t stride, int64_t n, GridDesc* d)
{
  float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
  for (int64_t i = blockIdx.x * (int64_t) blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x)
  {
    const float* p = xyz + i * stride;
    for (int a = 0; a < 3; a++)
    {
      mn[a] = fminf(mn[a], p[a]);
      mx[a] = fmaxf(mx[a], p[a]);
    }
  }
  __shared__ float smn[4][3], smx[4][3];
  for (int a = 0; a < 3; a++)
    for (int o = 32; o > 0; o >>= 1)
    {
      mn[a] = fminf(mn[a], __shfl_down(mn[a], o));
      mx[a] = fmaxf(mx[a], __shfl_down(mx[a], o));
    }
  if ((threadIdx.x & 63) == 0)
    for (int a = 0; a < 3; a++)
    {
      smn[threadIdx.x >> 6][a] = mn[a];
      smx[threadIdx.x >> 6][a] = mx[a];
    }
  __syncthreads();
  if (threadIdx.x < 3)  // one atomic pair per block and axis: same-address atomics cost ~12 ns each
  {
    const int a = threadIdx.x;
    const float lo = fminf(fminf(smn[0][a], smn[1][a]), fminf(smn[2][a], smn[3][a]));
    const float hi = fmaxf(fmaxf(smx[0][a], smx[1][a]), fmaxf(smx[2][a], smx[3][a]));
    atomicMin(&d->bbox[a], enc_float(lo));
    atomicMax(&d->bbox[3 + a], enc_float(hi));
  }
}

__global__ void k_desc_finish(GridDesc* d, double base_cell, int64_t n)
{
  double mn[3], mx[3];
  for (int a = 0; a < 3; a++)
  {
    mn[a] = n > 0 ? (double) dec_float(d->bbox[a]) : 0.0;
    mx[a] = n > 0 ? (double) dec_float(d->bbox[3 + a]) : 0.0;
  }
  double cell = base_cell;
  int dim[3];
  for (;;)
  {
    double prod = 1.0;
    for (int a = 0; a < 3; a++)
    {
      dim[a] = (int) floor((mx[a] - mn[a]) / cell) + 1;
      prod *= (double) dim[a];
    }
    if (prod <= (double) kCellCap)
      break;
    cell *= 2.0;
  }
  for (int a = 0; a < 3; a++)
  {
    d->mn[a] = mn[a];
    d->dim[a] = dim[a];
  }
  d->cell = cell;
  d->inv_cell = 1.0 / cell;
  d->ncell = dim[0] * dim[1] * dim[2];
}

// Cell histogram.  Clouds arrive in voxel order (localization.cpp:282-351), so consecutive points mostly share a cell:
// each run of equal cells inside a wave issues ONE atomic (with return), and every point remembers its rank inside its
// cell, which makes the scatter below atomic-free.
__global__ __launch_bounds__(256) void k_cell_count(const float* __restrict__ xyz, int64_t stride, int64_t n,
  const GridDesc* __restrict__ d, int* __restrict__ cell_of, int* __restrict__ rank_of, int* __restrict__ count)
{
  const GridDesc g = *d;
  const int lane = threadIdx.x & 63;
  const int64_t step = (int64_t) gridDim.x * blockDim.x;
  const int64_t n_up = (n + 63) & ~(int64_t) 63;  // whole waves iterate together (the shuffles below need all lanes)
  for (int64_t i = blockIdx.x * (int64_t) blockDim.x + threadIdx.x; i < n_up; i += step)
  {
    int c = -1;
    if (i < n)
    {
      const float* p = xyz + i * stride;
      const int cx = cell_coord(g, (double) p[0], 0), cy = cell_coord(g, (double) p[1], 1),
                cz = cell_coord(g, (double) p[2], 2);
      c = (cz * g.dim[1] + cy) * g.dim[0] + cx;
    }
    const int prev = __shfl_up(c, 1);
    const bool head = (lane == 0) || (c != prev);
    const unsigned long long hm = __ballot(head);
    const int hl = 63 - __clzll((long long) (hm & ((2ull << lane) - 1ull)));  // head lane of this lane's run
    const unsigned long long after = (hl == 63) ? 0ull : (hm >> (hl + 1));
    const int runlen = after ? (__ffsll((long long) after)) : (64 - hl);
    int base = 0;
    if (head && c >= 0)
      base = atomicAdd(&count[c], runlen);
    base = __shfl(base, hl);
    if (i < n)
    {
      cell_of[i] = c;
      rank_of[i] = base + (lane - hl);
    }
  }
}

// 3-phase exclusive scan over the first ncell (+1) entries; 1024 entries per block.
constexpr int kScanBlock = 1024;

__device__ __forceinline__ int block_scan_excl(int v, int* total)
{
  // 256 threads: wave scan + cross-wave
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
  for (int o = 1; o < 64; o <<= 1)
  {
    const int t = __shfl_up(inc, o);
    if (lane >= o)
      inc += t;
  }
  if (lane == 63)
    wsum[w] = inc;
  __syncthreads();
  int base = 0;
  for (int k = 0; k < w; k++)
    base += wsum[k];
  *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  return base + inc - v;
}

__global__ __launch_bounds__(256) void k_scan_sums(const int* __restrict__ count, const GridDesc* __restrict__ d,
  int* __restrict__ block_sums)
{
  const int ncell = d->ncell;
  const int b0 = blockIdx.x * kScanBlock;
  if (b0 >= ncell)
    return;
  int s = 0;
  for (int k = 0; k < 4; k++)
  {
    const int i = b0 + threadIdx.x * 4 + k;
    if (i < ncell)
      s += count[i];
  }
  int total;
  block_scan_excl(s, &total);
  if (threadIdx.x == 0)
    block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void k_scan_top(int* __restrict__ block_sums, const GridDesc* __restrict__ d)
{
  const int nb = (d->ncell + kScanBlock - 1) / kScanBlock;  // <= kCellCap / 1024 = 2048
  int carry = 0;
  for (int b0 = 0; b0 < nb; b0 += 256)
  {
    const int i = b0 + threadIdx.x;
    const int v = i < nb ? block_sums[i] : 0;
    int total;
    const int ex = block_scan_excl(v, &total);
    if (i < nb)
      block_sums[i] = carry + ex;
    carry += total;
  }
}

__global__ __launch_bounds__(256) void k_scan_final(const int* __restrict__ count, const GridDesc* __restrict__ d,
  const int* __restrict__ block_sums, int* __restrict__ cell_start, int n)
{
  const int ncell = d->ncell;
  const int b0 = blockIdx.x * kScanBlock;
  if (b0 >= ncell)
    return;
  int v[4], s = 0;
  for (int k = 0; k < 4; k++)
  {
    const int i = b0 + threadIdx.x * 4 + k;
    v[k] = i < ncell ? count[i] : 0;
    s += v[k];
  }
  int total;
  int ex = block_scan_excl(s, &total) + block_sums[blockIdx.x];
  for (int k = 0; k < 4; k++)
  {
    const int i = b0 + threadIdx.x * 4 + k;
    if (i < ncell)
      cell_start[i] = ex;
    ex += v[k];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    cell_start[ncell] = n;
}

__global__ __launch_bounds__(256) void k_scatter(const float* __restrict__ xyz, int64_t stride,
  const int32_t* __restrict__ cam, int64_t n, const int* __restrict__ cell_of, const int* __restrict__ rank_of,
  const int* __restrict__ cell_start, float4* __restrict__ sorted)
{
  for (int64_t i = blockIdx.x * (int64_t) blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x)
  {
    const int pos = cell_start[cell_of[i]] + rank_of[i];
    const float* p = xyz + i * stride;
    const unsigned w = ((unsigned) i << 1) | (cam ? (unsigned) (cam[i] & 1) : 0u);
    sorted[pos] = make_float4(p[0], p[1], p[2], __uint_as_float(w));
  }
}

int grid_build(Ctx* c, hipStream_t st)
{
  const int64_t n = c->n;
  const int nblk = (int) std::min<int64_t>((n + 255) / 256, 2048);
  hipMemsetAsync(c->d_cell_count, 0, sizeof(int) * kCellCap, st);
  hipLaunchKernelGGL(k_desc_init, dim3(1), dim3(1), 0, st, c->d_desc);
  if (n > 0)
    hipLaunchKernelGGL(k_bbox, dim3(std::min(nblk, 128)), dim3(256), 0, st, c->d_xyz, c->stride_floats, n, c->d_desc);
  // cell >= r_hands/4 keeps a ball query within 9 x 9 rows
  const double base_cell = std::max(0.02, c->p.nn_radius_hands / 4.0);
  hipLaunchKernelGGL(k_desc_finish, dim3(1), dim3(1), 0, st, c->d_desc, base_cell, n);
  if (n > 0)
    hipLaunchKernelGGL(k_cell_count, dim3(nblk), dim3(256), 0, st, c->d_xyz, c->stride_floats, n, c->d_desc,
      c->d_cell_of, c->d_rank_of, c->d_cell_count);
  const int sb = kCellCap / kScanBlock;
  hipLaunchKernelGGL(k_scan_sums, dim3(sb), dim3(256), 0, st, c->d_cell_count, c->d_desc, c->d_block_sums);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, st, c->d_block_sums, c->d_desc);
  hipLaunchKernelGGL(k_scan_final, dim3(sb), dim3(256), 0, st, c->d_cell_count, c->d_desc, c->d_block_sums,
    c->d_cell_start, (int) n);
  if (n > 0)
    hipLaunchKernelGGL(k_scatter, dim3(nblk), dim3(256), 0, st, c->d_xyz, c->stride_floats, c->d_cam, n, c->d_cell_of,
      c->d_rank_of, c->d_cell_start, c->d_sorted);
  return hipGetLastError() == hipSuccess ? AGH_OK : AGH_ERR_HIP;
}

}  // namespace agh
