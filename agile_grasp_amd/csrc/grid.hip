// grid.hip -- K0: uniform-grid build (bounding box, cell histogram, scan, cell-major scatter).
//
// Stands in for pcl::KdTreeFLANN::setInputCloud (reference src/agile_grasp/hand_search.cpp:10-11): the search
// structure behind the three radius searches of the hot path.  The cloud is re-laid out cell-major as
// float4 {x, y, z, bits((index << 1) | cam)} so that a ball query is a handful of contiguous runs read with
// coalesced 16-byte loads.  Everything runs on the device without a host round trip: the grid descriptor
// (origin, cell size, dimensions) lives in device memory and later kernels read it from there.
#include "agh_internal.h"

#include <algorithm>
#include <cstdlib>

namespace agh
{

// Reset state of the self-cleaning fields (also written once by agh_create / after a failed build).
__global__ void k_desc_reset(GridDesc* d)
{
  d += blockIdx.x;
  d->done = 0u;
  d->ticket = 0u;
}

__device__ void desc_finish(GridDesc* d, double base_cell, int64_t n, const float lo[3], const float hi[3])
{
  double mn[3], mx[3];
  for (int a = 0; a < 3; a++)
  {
    // (an empty cloud, or one without a single finite point, leaves the sentinels: a one-cell grid at the origin)
    const bool any = n > 0 && lo[a] <= hi[a];
    mn[a] = any ? (double) lo[a] : 0.0;
    mx[a] = any ? (double) hi[a] : 0.0;
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

// Bounding box of the cloud, first half: every work-group leaves its six extrema in its own slot of `part` (plain stores).
// The second half -- the reduction of the (<= 128) slots and the grid descriptor -- is done by EVERY work-group of
// k_cell_count for itself, behind its own point load: the "last work-group finishes" form of this kernel (slot stores, fence,
// arrival counter, fence, agent-scope reads of the slots, descriptor) was a chain of four memory round trips inside one
// launch, 8.2 us for a 300k-point cloud of which the point loop is two.  (Six atomicMin/Max per work-group on one cache
// line, the form before that: 9.8 us.)
constexpr int kBboxThreads = 1024;
__global__ __launch_bounds__(kBboxThreads) void k_bbox(const float* __restrict__ xyz, int64_t stride, const int* __restrict__ cloud_off,
  float* __restrict__ part)
{
  // blockIdx.y = cloud of the batch
  const int64_t p0 = cloud_off[blockIdx.y], n = cloud_off[blockIdx.y + 1] - p0;
  xyz += p0 * stride;
  part += (int64_t) blockIdx.y * kBboxBlocks * 6;
  float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
  // (the loop is latency bound: 131072 threads, two points in flight per thread, so a 300k-point cloud takes two rounds)
  const int64_t step = (int64_t) gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t) blockDim.x + threadIdx.x; i < n; i += 2 * step)
  {
    const float* p = xyz + i * stride;
    const float* q = xyz + (i + step < n ? i + step : i) * stride;
    const float pv[3] = { p[0], p[1], p[2] }, qv[3] = { q[0], q[1], q[2] };
    // a point with a non-finite coordinate is no point of the search structure (pcl::KdTreeFLANN::setInputCloud drops it:
    // it can be nobody's neighbour) and does not stretch the box
    const bool pf = isfinite(pv[0]) && isfinite(pv[1]) && isfinite(pv[2]), qf = isfinite(qv[0]) && isfinite(qv[1]) && isfinite(qv[2]);
    for (int a = 0; a < 3; a++)
    {
      mn[a] = fminf(mn[a], fminf(pf ? pv[a] : INFINITY, qf ? qv[a] : INFINITY));
      mx[a] = fmaxf(mx[a], fmaxf(pf ? pv[a] : -INFINITY, qf ? qv[a] : -INFINITY));
    }
  }
  constexpr int kW = kBboxThreads / 64;
  __shared__ float smn[kW][3], smx[kW][3];
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
  if (threadIdx.x < 3)
  {
    const int a = threadIdx.x;
    float lo = smn[0][a], hi = smx[0][a];
    for (int w = 1; w < kW; w++)
    {
      lo = fminf(lo, smn[w][a]);
      hi = fmaxf(hi, smx[w][a]);
    }
    part[blockIdx.x * 6 + a] = lo;
    part[blockIdx.x * 6 + 3 + a] = hi;
  }
}

// The reduction of k_bbox's slots and the grid descriptor, by one work-group for itself (256 threads; `g` in LDS).
__device__ __forceinline__ void desc_from_parts(const float* __restrict__ part, int nparts, double base_cell, int64_t n, GridDesc* g)
{
  __shared__ float red[4][6];
  const int tid = threadIdx.x, lane = tid & 63;
  float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
  for (int b = tid; b < nparts; b += blockDim.x)
    for (int a = 0; a < 3; a++)
    {
      lo[a] = fminf(lo[a], part[b * 6 + a]);
      hi[a] = fmaxf(hi[a], part[b * 6 + 3 + a]);
    }
  for (int a = 0; a < 3; a++)
    for (int o = 32; o > 0; o >>= 1)
    {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
    }
  if (lane == 0 && tid < 256)
    for (int a = 0; a < 3; a++)
    {
      red[tid >> 6][a] = lo[a];
      red[tid >> 6][3 + a] = hi[a];
    }
  __syncthreads();
  if (tid == 0)
  {
    for (int a = 0; a < 3; a++)
      for (int w = 1; w < 4; w++)
      {
        lo[a] = fminf(lo[a], red[w][a]);
        hi[a] = fmaxf(hi[a], red[w][3 + a]);
      }
    desc_finish(g, base_cell, n, lo, hi);
  }
  __syncthreads();
}

// Cell histogram.  Clouds arrive in voxel order (localization.cpp:282-351), so consecutive points mostly share a cell:
// each run of equal cells inside a wave issues ONE atomic (with return), and every point remembers its rank inside its
// cell, which makes the scatter below atomic-free.
__global__ __launch_bounds__(256) void k_cell_count(const float* __restrict__ xyz, int64_t stride,
  const int* __restrict__ cloud_off, GridDesc* __restrict__ d, int* __restrict__ cell_of, int* __restrict__ rank_of,
  int* __restrict__ count, const float* __restrict__ part, int nparts, double base_cell)
{
  const int64_t p0 = cloud_off[blockIdx.y], n = cloud_off[blockIdx.y + 1] - p0;
  xyz += p0 * stride;
  cell_of += p0;
  rank_of += p0;
  count += (int64_t) blockIdx.y * kCellCap;
  // the grid descriptor from k_bbox's slots: every work-group for itself; the first one of a cloud publishes it for the
  // kernels after this one
  __shared__ GridDesc gs;
  desc_from_parts(part + (int64_t) blockIdx.y * kBboxBlocks * 6, nparts, base_cell, n, &gs);
  const GridDesc g = gs;
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    GridDesc* o = d + blockIdx.y;  // (field by field: `ticket` belongs to k_cell_scan)
    for (int a = 0; a < 3; a++)
    {
      o->mn[a] = g.mn[a];
      o->dim[a] = g.dim[a];
    }
    o->cell = g.cell;
    o->inv_cell = g.inv_cell;
    o->ncell = g.ncell;
  }
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
      // A non-finite point keeps its place in the index space (the callers' indices do not move) and its own coordinates in the
      // sorted array, where every query's float32 distance test rejects it (NaN and Inf compare false with `< r^2`); which cell
      // it lies in is therefore free -- spread over the cells by index, so that no cell collects a sensor's drop-outs.
      if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2])))
        c = (int) (i % (int64_t) g.ncell);
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

// Exclusive scan of the cell histogram in ONE launch (decoupled look-back): tiles of 1024 cells are walked round-robin
// by 256 resident work-groups, so a tile's predecessors are always running or done and the look-back cannot deadlock.  A tile
// publishes (build number, flag, value) in one 64-bit word: flag 1 = the tile's own total, 2 = its inclusive prefix;
// the build number makes stale words from earlier builds invisible, so the descriptors never need clearing.  The kernel
// also zeroes the histogram it has read: the next build starts from a clean one without a memset.
constexpr int kScanBlock = 4096;  // cells per tile: 16 per thread, four 16-byte accesses each way

__device__ __forceinline__ int block_scan_excl(int v, int* total)
{
  // 256 threads: wave scan + cross-wave
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int inc = wave_incl_scan_i32(v);
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

__global__ __launch_bounds__(256) void k_cell_scan(int* __restrict__ count, GridDesc* __restrict__ d,
  unsigned long long* __restrict__ tile_state, unsigned gen, int* __restrict__ cell_start, const int* __restrict__ cloud_off)
{
  __shared__ int s_prefix;
  // blockIdx.y = cloud: its own histogram, tile descriptors and cell table; the table holds positions in the common
  // sorted array, i.e. it starts at the cloud's first point.  (Work-groups are dispatched x-fastest, so the predecessors a
  // tile waits for were dispatched before it, for every cloud.)
  const int p0 = cloud_off[blockIdx.y], n = cloud_off[blockIdx.y + 1] - p0;
  count += (int64_t) blockIdx.y * kCellCap;
  tile_state += (int64_t) blockIdx.y * (kCellCap / kScanBlock);
  cell_start += (int64_t) blockIdx.y * kCellStride;
  d += blockIdx.y;
  const int ncell = d->ncell;
  // 256 resident work-groups walk the tiles round-robin: tile t only ever waits for tiles < t, which belong to the
  // first pass of lower-numbered groups or to an earlier pass, so the look-back cannot deadlock (and no same-address
  // ticket atomics, ~12 ns each, are needed to order the tiles)
  for (int tile = blockIdx.x; tile * kScanBlock < ncell; tile += gridDim.x)
  {
  const int b0 = tile * kScanBlock;
  // thread t owns the cells b0 + 16 t .. + 15 (the table is padded to whole tiles, and cells past ncell hold zero counts)
  int4 v[4];
  int sum = 0;
  for (int k = 0; k < 4; k++)
  {
    v[k] = reinterpret_cast<const int4*>(count + b0)[threadIdx.x * 4 + k];
    sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
  int total;
  int ex = block_scan_excl(sum, &total);
  const unsigned long long tag = (unsigned long long) gen << 34;
  if (threadIdx.x < 64)  // wave 0 publishes and looks back, 64 predecessors per round (one serial load per
  {                      // predecessor would make the last tile wait ~0.6 us x its index)
    const int lane = threadIdx.x;
    int prefix = 0;
    if (lane == 0)
      __hip_atomic_store(&tile_state[tile], tag | ((tile == 0 ? 2ull : 1ull) << 32) | (unsigned) total, __ATOMIC_RELEASE,
        __HIP_MEMORY_SCOPE_AGENT);
    for (int j0 = tile - 1; j0 >= 0; j0 -= 64)
    {
      const int j = j0 - lane;
      unsigned long long w = 0;
      for (;;)
      {
        bool ready = true;
        if (j >= 0)
        {
          w = __hip_atomic_load(&tile_state[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
          ready = (w >> 34) == gen && ((w >> 32) & 3ull) != 0ull;
        }
        if (__ballot(!ready) == 0ull)
          break;
        __builtin_amdgcn_s_sleep(1);  // a predecessor has not published yet
      }
      const unsigned long long incl = __ballot(j >= 0 && ((w >> 32) & 3ull) == 2ull);
      const int stop = incl ? __ffsll((long long) incl) - 1 : 63;  // nearest predecessor with an inclusive prefix
      int val = (j >= 0 && lane <= stop) ? (int) (unsigned) w : 0;
      for (int o = 32; o > 0; o >>= 1)
        val += __shfl_xor(val, o);
      prefix += val;
      if (incl)
        break;
    }
    if (lane == 0)
    {
      if (tile != 0)
        __hip_atomic_store(&tile_state[tile], tag | (2ull << 32) | (unsigned) (prefix + total), __ATOMIC_RELEASE,
          __HIP_MEMORY_SCOPE_AGENT);
      s_prefix = prefix;
    }
  }
  __syncthreads();
  ex += s_prefix + p0;
  for (int k = 0; k < 4; k++)
  {
    int4 o;
    o.x = ex;
    o.y = o.x + v[k].x;
    o.z = o.y + v[k].y;
    o.w = o.z + v[k].z;
    ex = o.w + v[k].w;
    // (entries past ncell are scratch of the padded table; the one AT ncell is written below, after everybody's stores)
    reinterpret_cast<int4*>(cell_start + b0)[threadIdx.x * 4 + k] = o;
    reinterpret_cast<int4*>(count + b0)[threadIdx.x * 4 + k] = make_int4(0, 0, 0, 0);
  }
  if (tile == 0 && threadIdx.x == 0)
    cell_start[ncell] = p0 + n;
  __syncthreads();  // s_prefix is reused by the next tile of this group
  }
}

__global__ __launch_bounds__(256) void k_scatter(const float* __restrict__ xyz, int64_t stride,
  const int32_t* __restrict__ cam, const int* __restrict__ cloud_off, const int* __restrict__ cell_of,
  const int* __restrict__ rank_of, const int* __restrict__ cell_start, float4* __restrict__ sorted)
{
  const int64_t p0 = cloud_off[blockIdx.y], p1 = cloud_off[blockIdx.y + 1];
  cell_start += (int64_t) blockIdx.y * kCellStride;
  for (int64_t i = p0 + blockIdx.x * (int64_t) blockDim.x + threadIdx.x; i < p1; i += (int64_t) gridDim.x * blockDim.x)
  {
    const int pos = cell_start[cell_of[i]] + rank_of[i];
    const float* p = xyz + i * stride;
    const unsigned w = ((unsigned) i << 1) | (cam ? (unsigned) (cam[i] & 1) : 0u);
    sorted[pos] = make_float4(p[0], p[1], p[2], __uint_as_float(w));
  }
}

int grid_build(Ctx* c, hipStream_t st)
{
  // one launch per stage for the whole batch: blockIdx.y = cloud
  const int C = c->n_clouds;
  int64_t nmax = 0;
  for (int k = 0; k < C; k++)
    nmax = std::max<int64_t>(nmax, c->cloud_off[(size_t) k + 1] - c->cloud_off[(size_t) k]);
  const int nblk = (int) std::max<int64_t>(1, std::min<int64_t>((nmax + 255) / 256, 2048));
  if (!c->grid_clean)  // first build of the context, or the previous one failed half-way
  {
    hipMemsetAsync(c->d_cell_count, 0, sizeof(int) * (size_t) kCellCap * c->clouds_cap, st);
    hipMemsetAsync(c->d_tile_state, 0, sizeof(unsigned long long) * (size_t) (kCellCap / kScanBlock) * c->clouds_cap, st);
    hipLaunchKernelGGL(k_desc_reset, dim3(c->clouds_cap), dim3(1), 0, st, c->d_desc);
    c->build_gen = 0;
  }
  c->grid_clean = false;
  if (++c->build_gen >= (1u << 30))  // the tag has 30 bits: start over long before it wraps
  {
    hipMemsetAsync(c->d_tile_state, 0, sizeof(unsigned long long) * (size_t) (kCellCap / kScanBlock) * c->clouds_cap, st);
    c->build_gen = 1;
  }
  // host-buffer agh_set_cloud: the camera ids go up on a stream of their own beside the coordinate-only kernels below.  That
  // stream must not write own_cam while work queued EARLIER on `st` (the previous build's k_scatter, a search) still reads it:
  // with a pinned source nothing else orders the two (a pageable copy happened to, by blocking).  The gate is recorded here,
  // before this build's kernels, so the copy waits for the earlier work only.
  bool cam_gate = false;
  if (c->pending_cam_host)
  {
    if (!c->copy_stream && !c->copy_stream_failed &&
        (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess ||
         hipEventCreateWithFlags(&c->copy_done, hipEventDisableTiming) != hipSuccess ||
         hipEventCreateWithFlags(&c->copy_gate, hipEventDisableTiming) != hipSuccess))
    {
      // whatever was created goes again, and the failure is remembered: the camera ids then go up on the build's own stream
      // for the rest of the context's life instead of retrying (and leaking) on every build
      if (c->copy_gate)
        (void) hipEventDestroy(c->copy_gate);
      if (c->copy_done)
        (void) hipEventDestroy(c->copy_done);
      if (c->copy_stream)
        (void) hipStreamDestroy(c->copy_stream);
      c->copy_gate = nullptr;
      c->copy_done = nullptr;
      c->copy_stream = nullptr;
      c->copy_stream_failed = true;
    }
    if (c->copy_stream)
      cam_gate = hipEventRecord(c->copy_gate, st) == hipSuccess && hipStreamWaitEvent(c->copy_stream, c->copy_gate, 0) == hipSuccess;
  }
  // cell >= r_hands/4 keeps a ball query within 9 x 9 rows
  const double base_cell = std::max(0.02, c->p.nn_radius_hands / 4.0);
  const int nparts = std::max(1, std::min(nblk, kBboxBlocks));
  hipLaunchKernelGGL(k_bbox, dim3(nparts, C), dim3(kBboxThreads), 0, st, c->d_xyz, c->stride_floats,
    (const int*) c->d_cloud_off, c->d_bbox_part);
  // (also for an empty cloud: one work-group writes the descriptor of an empty grid)
  hipLaunchKernelGGL(k_cell_count, dim3(nblk, C), dim3(256), 0, st, c->d_xyz, c->stride_floats, (const int*) c->d_cloud_off,
    c->d_desc, c->d_cell_of, c->d_rank_of, c->d_cell_count, (const float*) c->d_bbox_part, nparts, base_cell);
  hipLaunchKernelGGL(k_cell_scan, dim3(C == 1 ? 256 : 64, C), dim3(256), 0, st, c->d_cell_count, c->d_desc, c->d_tile_state,
    c->build_gen, c->d_cell_start, (const int*) c->d_cloud_off);
  if (c->pending_cam_host)
  {
    // host-buffer agh_set_cloud: the camera ids (a blocking pageable copy, ~1.2 MB for 300k points) go up while the three
    // kernels above run -- they read coordinates only.  On a stream of its own: on `st` the transfer would queue behind those
    // kernels.  A pageable copy returns when the SOURCE has been read (into the runtime's staging buffers), not necessarily
    // when the data is on the device, so k_scatter waits for an event behind the copy; the coordinates' copy, which was
    // ordered behind everything earlier on the context's stream, has already returned when this one is issued.
    hipStream_t cs = (c->copy_stream && cam_gate) ? c->copy_stream : st;  // (no gate: in order on `st`, behind the kernels above)
    hipError_t e = hipMemcpyAsync(c->own_cam, c->pending_cam_host, sizeof(int32_t) * (size_t) c->pending_cam_n,
      hipMemcpyHostToDevice, cs);
    c->pending_cam_host = nullptr;
    c->cam_copy_on_copy_stream = cs != st;
    if (e == hipSuccess && cs != st)
    {
      e = hipEventRecord(c->copy_done, cs);
      if (e == hipSuccess)
        e = hipStreamWaitEvent(st, c->copy_done, 0);
    }
    if (e != hipSuccess)
      return AGH_ERR_HIP;
  }
  if (nmax > 0)
    hipLaunchKernelGGL(k_scatter, dim3(nblk, C), dim3(256), 0, st, c->d_xyz, c->stride_floats, c->d_cam,
      (const int*) c->d_cloud_off, c->d_cell_of, c->d_rank_of, c->d_cell_start, c->d_sorted);
  if (hipGetLastError() != hipSuccess)
    return AGH_ERR_HIP;
  c->grid_clean = true;
  return AGH_OK;
}

}  // namespace agh
