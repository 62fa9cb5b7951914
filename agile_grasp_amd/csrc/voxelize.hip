// voxelize.hip -- K-1: the preprocessing in front of the search.
//
// Stands in for the head of Localization::localizeHands (reference src/agile_grasp/localization.cpp):
//   17-24    camera id of raw point i = (i >= size_left), assigned before the NaN removal and never re-indexed
//   25-27    pcl::removeNaNFromPointCloud (order preserving; a cloud flagged is_dense is passed through)
//   216-245  filterWorkspace: min <= p <= max per axis
//   247-355  voxelizeCloud: per-camera minimum, floor((p - min) / cell), std::set in lexicographic (x, y, z) order,
//            coordinates back as v * cell + min, camera 0 block then camera 1 block.
//
// The std::set is replaced by a bitmap over each camera's voxel lattice laid out x-major, z-fastest: a set bit's
// position IS its lexicographic rank order, so "insert" is an atomicOr and "iterate in order" is a popcount scan --
// no sort.  A 1.3 m x 1.0 m x 0.5 m scene at 3 mm is 24 Mbit = 3 MB per camera (L2 / Infinity-Cache resident).
// The arithmetic that defines the voxel of a point and the coordinates of a voxel is the reference's, in double,
// without contraction.
#include "agh_internal.h"

namespace agh
{

constexpr int kPreBlock = 256;
constexpr int kPrePerBlock = 1024;   // points per block (4 rounds of 256)
constexpr int kWordsPerBlock = 4096; // bitmap words per block in the popcount passes (16 per thread)

__device__ __forceinline__ void vox_desc_init(VoxDesc* d)
{
  for (int c = 0; c < 2; c++)
    for (int a = 0; a < 3; a++)
    {
      d->mn_enc[c][a] = enc_float(10000.0f);  // the reference's initial minimum (localization.cpp:250-252)
      d->mx_enc[c][a] = 0u;
    }
  d->n_kept[0] = d->n_kept[1] = 0;
  d->n_vox[0] = d->n_vox[1] = 0;
  d->error = 0;
}
__global__ void k_vox_init(VoxDesc* d)
{
  vox_desc_init(d);
}

__device__ __forceinline__ bool finite3(float x, float y, float z)
{
  return isfinite(x) && isfinite(y) && isfinite(z);
}

// Finite points per block of 1024 raw points (rank base of the NaN-free cloud).
__global__ __launch_bounds__(kPreBlock) void k_vox_count(const float* __restrict__ xyz, int64_t stride, int64_t n,
  int* __restrict__ blk_cnt)
{
  const int64_t base = (int64_t) blockIdx.x * kPrePerBlock;
  int cnt = 0;
  for (int r = 0; r < 4; r++)
  {
    const int64_t i = base + r * kPreBlock + threadIdx.x;
    if (i < n)
    {
      const float* p = xyz + i * stride;
      cnt += finite3(p[0], p[1], p[2]) ? 1 : 0;
    }
  }
  for (int o = 32; o > 0; o >>= 1)
    cnt += __shfl_down(cnt, o);
  __shared__ int s[4];
  if ((threadIdx.x & 63) == 0)
    s[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0)
    blk_cnt[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__device__ void vox_totals(VoxDesc* d, const int* __restrict__ blk_prefix, long long total, VoxDesc* host_desc, int* cloud_off_out);

// Exclusive scan of an int array with one block (in place); total to *total if given.  Two one-thread kernels ride along
// (each was a ~5 us launch of its own: a few dependent global accesses by one thread): init_desc -- the descriptor's
// reset in front of k_vox_classify --, and totals_desc -- the voxel counts per camera and the host mirror behind the second scan.
// cloud_off_out: the context's device-side cloud offsets {0, voxel count} -- so that a grid build queued behind this kernel
// finds the count on the device, without the host having to wait for it (agh_localize).
__global__ __launch_bounds__(1024) void k_vox_scan(int* __restrict__ v, int64_t nb, long long* total, VoxDesc* init_desc,
  VoxDesc* totals_desc, VoxDesc* host_desc, int* cloud_off_out)
{
  __shared__ long long carry;
  __shared__ int wsum[16];
  if (threadIdx.x == 0)
  {
    carry = 0;
    if (init_desc)
      vox_desc_init(init_desc);
  }
  __syncthreads();
  for (int64_t b0 = 0; b0 < nb; b0 += 1024)
  {
    const int64_t i = b0 + threadIdx.x;
    const int x = i < nb ? v[i] : 0;
    int incl = x;
    for (int o = 1; o < 64; o <<= 1)
    {
      const int y = __shfl_up(incl, o);
      if ((int) (threadIdx.x & 63) >= o)
        incl += y;
    }
    if ((threadIdx.x & 63) == 63)
      wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < (int) (threadIdx.x >> 6); w++)
      wbase += wsum[w];
    if (i < nb)
      v[i] = (int) (carry + wbase + incl - x);  // (ranks < 2^30 by the API's bound on n)
    __syncthreads();
    if (threadIdx.x == 1023)
      carry += wbase + incl;
    __syncthreads();
  }
  if (total && threadIdx.x == 0)
    *total = carry;
  if (totals_desc && threadIdx.x == 0)  // (v was written by this work-group: the barriers above order it)
    vox_totals(totals_desc, v, carry, host_desc, cloud_off_out);
}

// Camera id (rank in the NaN-free cloud >= size_left), workspace test, per-camera minimum and maximum.
__global__ __launch_bounds__(kPreBlock) void k_vox_classify(const float* __restrict__ xyz, int64_t stride, int64_t n,
  const int* __restrict__ blk_prefix, int64_t size_left, VoxWorkspace ws, uint8_t* __restrict__ code, VoxDesc* d)
{
  __shared__ int wcnt[4];
  __shared__ unsigned smn[2][3], smx[2][3];
  __shared__ int skept[2];
  if (threadIdx.x < 6)
  {
    smn[threadIdx.x / 3][threadIdx.x % 3] = 0xffffffffu;
    smx[threadIdx.x / 3][threadIdx.x % 3] = 0u;
  }
  if (threadIdx.x < 2)
    skept[threadIdx.x] = 0;
  const int64_t base = (int64_t) blockIdx.x * kPrePerBlock;
  int64_t running = blk_prefix ? (int64_t) blk_prefix[blockIdx.x] : base;  // finite points before this round
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned mn[2][3], mx[2][3];
  for (int c = 0; c < 2; c++)
    for (int a = 0; a < 3; a++)
    {
      mn[c][a] = 0xffffffffu;
      mx[c][a] = 0u;
    }
  int kept[2] = { 0, 0 };
  for (int r = 0; r < 4; r++)
  {
    const int64_t i = base + r * kPreBlock + threadIdx.x;
    float p[3] = { 0.f, 0.f, 0.f };
    bool fin = false;
    if (i < n)
    {
      const float* q = xyz + i * stride;
      p[0] = q[0];
      p[1] = q[1];
      p[2] = q[2];
      fin = blk_prefix ? finite3(p[0], p[1], p[2]) : true;
    }
    const unsigned long long m = __ballot(fin);
    __syncthreads();  // (also orders the LDS initialisation above before its first use)
    if (lane == 0)
      wcnt[wave] = __popcll(m);
    __syncthreads();
    int before = __popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; w++)
      before += wcnt[w];
    const int64_t rank = running + before;
    running += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    if (i < n)
    {
      const int cam = rank >= size_left ? 1 : 0;
      const bool in = fin && (double) p[0] >= ws.lo[0] && (double) p[0] <= ws.hi[0] && (double) p[1] >= ws.lo[1] &&
                      (double) p[1] <= ws.hi[1] && (double) p[2] >= ws.lo[2] && (double) p[2] <= ws.hi[2];
      code[i] = in ? (uint8_t) (1 | (cam << 1)) : (uint8_t) 0;
      if (in)
      {
        kept[cam]++;
        for (int a = 0; a < 3; a++)
        {
          const unsigned e = enc_float(p[a]);
          mn[cam][a] = min(mn[cam][a], e);
          mx[cam][a] = max(mx[cam][a], e);
        }
      }
    }
  }
  for (int c = 0; c < 2; c++)
  {
    for (int a = 0; a < 3; a++)
    {
      unsigned lo = mn[c][a], hi = mx[c][a];
      for (int o = 32; o > 0; o >>= 1)
      {
        lo = min(lo, (unsigned) __shfl_down(lo, o));
        hi = max(hi, (unsigned) __shfl_down(hi, o));
      }
      if (lane == 0)
      {
        atomicMin(&smn[c][a], lo);
        atomicMax(&smx[c][a], hi);
      }
    }
    int k = kept[c];
    for (int o = 32; o > 0; o >>= 1)
      k += __shfl_down(k, o);
    if (lane == 0 && k)
      atomicAdd(&skept[c], k);
  }
  __syncthreads();
  if (threadIdx.x < 6)  // one global atomic pair per block, camera and axis
  {
    const int c = threadIdx.x / 3, a = threadIdx.x % 3;
    if (skept[c] > 0)
    {
      atomicMin(&d->mn_enc[c][a], smn[c][a]);
      atomicMax(&d->mx_enc[c][a], smx[c][a]);
    }
  }
  if (threadIdx.x < 2 && skept[threadIdx.x] > 0)
    atomicAdd((unsigned long long*) &d->n_kept[threadIdx.x], (unsigned long long) skept[threadIdx.x]);
}

// Voxel index along one axis exactly as localization.cpp:288: floor((double(p) - min) / cell).
__device__ __forceinline__ long long vox_index(float p, double mn, double cell)
{
  return (long long) floor(((double) p - mn) / cell);
}

// error: 1 = the lattice exceeds max_words (the hard limit), 2 = it exceeds cap_words, the bitmap the host has allocated from an
// earlier cloud (stage 2 was launched speculatively for that size and does nothing; the host enlarges and repeats).
__device__ void vox_lattice(VoxDesc* d, double cell, unsigned long long max_words, unsigned long long cap_words, VoxDesc* host_desc)
{
  unsigned long long ofs = 0;
  for (int c = 0; c < 2; c++)
  {
    unsigned long long bits = 0;
    for (int a = 0; a < 3; a++)
    {
      d->mn[c][a] = (double) dec_float(d->mn_enc[c][a]);
      d->dim[c][a] = 0;
    }
    if (d->n_kept[c] > 0)
    {
      bits = 1;
      for (int a = 0; a < 3; a++)
      {
        const long long top = vox_index(dec_float(d->mx_enc[c][a]), d->mn[c][a], cell);
        if (top < 0 || top >= (1ll << 31) - 1)
        {
          d->error = 1;
          bits = 0;
          break;
        }
        d->dim[c][a] = (int) top + 1;
        // (the product is checked against max_words step by step so that it cannot wrap)
        if (bits > (max_words * 32ull) / (unsigned long long) d->dim[c][a])
        {
          d->error = 1;
          bits = 0;
          break;
        }
        bits *= (unsigned long long) d->dim[c][a];
      }
    }
    d->bits[c] = bits;
    d->word_ofs[c] = ofs;
    unsigned long long words = (bits + 31ull) / 32ull;
    words = (words + kWordsPerBlock - 1ull) / kWordsPerBlock * kWordsPerBlock;  // camera 1 starts on a block boundary
    ofs += words;
  }
  d->n_words = ofs;
  if (ofs > max_words)
    d->error = 1;
  else if (!d->error && ofs > cap_words)
    d->error = 2;
  if (host_desc)
    *host_desc = *d;
}
__global__ void k_vox_lattice(VoxDesc* d, double cell, unsigned long long max_words, unsigned long long cap_words, VoxDesc* host_desc)
{
  vox_lattice(d, cell, max_words, cap_words, host_desc);
}
// The speculative pass (agh_preprocess_device from the second cloud on): the bitmap of the size the context already has is
// cleared by all work-groups while one thread computes the lattice -- one launch for what were a runtime fill kernel and a
// one-thread kernel, ~5 us each.  (k_vox_mark, the next kernel, is the first reader of both.)
__global__ __launch_bounds__(256) void k_vox_clear_lattice(uint4* __restrict__ bitmap16, int64_t n16, VoxDesc* d, double cell,
  unsigned long long max_words, unsigned long long cap_words, VoxDesc* host_desc)
{
  if (blockIdx.x == 0 && threadIdx.x == 0)
    vox_lattice(d, cell, max_words, cap_words, host_desc);
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t) gridDim.x * 256)
    bitmap16[i] = z;
}

__global__ __launch_bounds__(256) void k_vox_mark(const float* __restrict__ xyz, int64_t stride, int64_t n,
  const uint8_t* __restrict__ code, const VoxDesc* __restrict__ d, double cell, unsigned* __restrict__ bitmap)
{
  const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const unsigned cd = code[i];
  if (!cd || d->error)
    return;
  const int c = (int) (cd >> 1);
  const float* p = xyz + i * stride;
  const unsigned long long ix = (unsigned long long) vox_index(p[0], d->mn[c][0], cell);
  const unsigned long long iy = (unsigned long long) vox_index(p[1], d->mn[c][1], cell);
  const unsigned long long iz = (unsigned long long) vox_index(p[2], d->mn[c][2], cell);
  const unsigned long long pos = (ix * (unsigned long long) d->dim[c][1] + iy) * (unsigned long long) d->dim[c][2] + iz;
  atomicOr(&bitmap[d->word_ofs[c] + (pos >> 5)], 1u << (unsigned) (pos & 31ull));
}

__global__ __launch_bounds__(256) void k_vox_popcount(const unsigned* __restrict__ bitmap, int* __restrict__ blk_cnt)
{
  const uint4* w = (const uint4*) (bitmap + (size_t) blockIdx.x * kWordsPerBlock) + threadIdx.x * 4;
  int cnt = 0;
  for (int k = 0; k < 4; k++)
  {
    const uint4 v = w[k];
    cnt += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
  }
  for (int o = 32; o > 0; o >>= 1)
    cnt += __shfl_down(cnt, o);
  __shared__ int s[4];
  if ((threadIdx.x & 63) == 0)
    s[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0)
    blk_cnt[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__device__ void vox_totals(VoxDesc* d, const int* __restrict__ blk_prefix, long long total_v, VoxDesc* host_desc, int* cloud_off_out)
{
  const long long* total = &total_v;
  if (d->error)
  {
    d->n_vox[0] = d->n_vox[1] = 0;
    if (host_desc)
      *host_desc = *d;
    if (cloud_off_out)
      cloud_off_out[0] = cloud_off_out[1] = 0;
    return;
  }
  const long long first1 = d->word_ofs[1] / kWordsPerBlock < d->n_words / kWordsPerBlock
                             ? (long long) blk_prefix[d->word_ofs[1] / kWordsPerBlock]
                             : *total;
  d->n_vox[0] = first1;
  d->n_vox[1] = *total - first1;
  if (host_desc)
    *host_desc = *d;
  if (cloud_off_out)
  {
    cloud_off_out[0] = 0;
    cloud_off_out[1] = (int) *total;
  }
}

// Emit the voxels in bitmap order = (camera, x, y, z) lexicographic order; coordinates as localization.cpp:313-324.
__global__ __launch_bounds__(256) void k_vox_emit(const unsigned* __restrict__ bitmap, const int* __restrict__ blk_prefix,
  const VoxDesc* __restrict__ d, double cell, float* __restrict__ out_xyz, int32_t* __restrict__ out_cam)
{
  const size_t w0 = (size_t) blockIdx.x * kWordsPerBlock + (size_t) threadIdx.x * 16;
  unsigned w[16];
  const uint4* src = (const uint4*) (bitmap + w0);
  int cnt = 0;
  for (int k = 0; k < 4; k++)
  {
    const uint4 v = src[k];
    w[4 * k] = v.x;
    w[4 * k + 1] = v.y;
    w[4 * k + 2] = v.z;
    w[4 * k + 3] = v.w;
    cnt += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
  }
  int incl = cnt;
  for (int o = 1; o < 64; o <<= 1)
  {
    const int y = __shfl_up(incl, o);
    if ((int) (threadIdx.x & 63) >= o)
      incl += y;
  }
  __shared__ int wsum[4];
  if ((threadIdx.x & 63) == 63)
    wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  int loff = incl - cnt;  // this thread's first voxel among the work-group's
  for (int q = 0; q < (int) (threadIdx.x >> 6); q++)
    loff += wsum[q];
  const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  if (!total || d->error)
    return;
  const int64_t k0 = (int64_t) blk_prefix[blockIdx.x];
  const size_t wg0 = (size_t) blockIdx.x * kWordsPerBlock;
  const int c = wg0 >= d->word_ofs[1] ? 1 : 0;  // (camera 1 starts on a block boundary: a work-group is of one camera)
  const unsigned long long ny = (unsigned long long) d->dim[c][1], nz = (unsigned long long) d->dim[c][2];
  const double m0 = d->mn[c][0], m1 = d->mn[c][1], m2 = d->mn[c][2];
  // (ix, iy, iz) of a bit position: two exact integer divisions by way of double reciprocals (positions are below 2^33, the
  // quotient estimate is off by at most one and is corrected) -- three 64-bit integer divisions per VOXEL, ~100 instructions
  // each on this part, made the kernel 52 us for 250k voxels
  const double inv_nz = 1.0 / (double) nz, inv_ny = 1.0 / (double) ny;
  auto divmod = [](unsigned long long a, unsigned long long b, double inv_b, unsigned long long& q, unsigned long long& r) {
    q = (unsigned long long) ((double) a * inv_b);
    long long rr = (long long) a - (long long) (q * b);
    if (rr < 0)
    {
      q--;
      rr += (long long) b;
    }
    else if (rr >= (long long) b)
    {
      q++;
      rr -= (long long) b;
    }
    r = (unsigned long long) rr;
  };
  auto emit = [&](unsigned rel_bit, int64_t k) {  // rel_bit: bit position inside the work-group's 4096 words
    const unsigned long long pos = ((unsigned long long) wg0 - d->word_ofs[c]) * 32ull + rel_bit;
    unsigned long long t, uz, ux, uy;
    divmod(pos, nz, inv_nz, t, uz);
    divmod(t, ny, inv_ny, ux, uy);
    const long long iz = (long long) uz, iy = (long long) uy, ix = (long long) ux;
    out_xyz[3 * k] = (float) ((double) ix * cell + 1.0 * m0);
    out_xyz[3 * k + 1] = (float) ((double) iy * cell + 1.0 * m1);
    out_xyz[3 * k + 2] = (float) ((double) iz * cell + 1.0 * m2);
    out_cam[k] = c;
  };
  // The lattice is sparse (about one bit in a hundred is set) and the set bits sit unevenly in the threads' words: a thread that
  // walked its own bits and stored its own voxels kept a wave waiting for its fullest lane and scattered 12-byte stores over the
  // output (47 us for 250k voxels).  The work-group's set bits are LISTED in LDS first, in bitmap order, and then emitted a
  // voxel per thread: every lane busy, consecutive lanes write consecutive voxels.
  constexpr int kListCap = 4096;
  __shared__ unsigned list[kListCap];
  if (total <= kListCap)
  {
    int q = loff;
    for (int j = 0; j < 16; j++)
    {
      unsigned bits = w[j];
      while (bits)
      {
        const int b = __ffs(bits) - 1;
        bits &= bits - 1;
        list[q++] = (unsigned) (threadIdx.x * 16 + j) * 32u + (unsigned) b;
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < total; i += 256)
      emit(list[i], k0 + i);
    return;
  }
  // (a dense block -- more than a set bit per word on average: every thread walks its own words)
  int64_t k = k0 + loff;
  for (int j = 0; j < 16; j++)
  {
    unsigned bits = w[j];
    while (bits)
    {
      const int b = __ffs(bits) - 1;
      bits &= bits - 1;
      emit((unsigned) (threadIdx.x * 16 + j) * 32u + (unsigned) b, k);
      k++;
    }
  }
}

int vox_stage1(Ctx* c, const float* d_xyz, int64_t stride_floats, int64_t n, int64_t size_left, int dense,
  const double workspace[6], double cell, hipStream_t st, int64_t cap_words, VoxDesc* host_desc, bool with_lattice)
{
  VoxWorkspace ws;
  for (int a = 0; a < 3; a++)
  {
    ws.lo[a] = workspace[2 * a];
    ws.hi[a] = workspace[2 * a + 1];
  }
  const int64_t nb = (n + kPrePerBlock - 1) / kPrePerBlock;
  const bool scan_first = n > 0 && !dense;  // (then the descriptor's reset rides on that one-block scan)
  if (!scan_first)
    hipLaunchKernelGGL(k_vox_init, dim3(1), dim3(1), 0, st, c->d_vox_desc);
  if (n > 0)
  {
    if (!dense)
    {
      hipLaunchKernelGGL(k_vox_count, dim3((unsigned) nb), dim3(kPreBlock), 0, st, d_xyz, stride_floats, n, c->d_vox_blk);
      hipLaunchKernelGGL(k_vox_scan, dim3(1), dim3(1024), 0, st, c->d_vox_blk, nb, (long long*) nullptr, c->d_vox_desc,
        (VoxDesc*) nullptr, (VoxDesc*) nullptr, (int*) nullptr);
    }
    hipLaunchKernelGGL(k_vox_classify, dim3((unsigned) nb), dim3(kPreBlock), 0, st, d_xyz, stride_floats, n,
      dense ? (const int*) nullptr : (const int*) c->d_vox_blk, size_left, ws, c->d_vox_code, c->d_vox_desc);
  }
  if (with_lattice)  // (the speculative pass computes the lattice inside stage 2's first kernel)
    hipLaunchKernelGGL(k_vox_lattice, dim3(1), dim3(1), 0, st, c->d_vox_desc, cell, (unsigned long long) kVoxMaxWords,
      (unsigned long long) cap_words, host_desc);
  return hipGetLastError() == hipSuccess ? AGH_OK : AGH_ERR_HIP;
}

// n_words: the lattice's size, or any multiple of kWordsPerBlock above it that the bitmap has room for (the blocks beyond the
// lattice hold no bits and emit nothing)
int vox_stage2(Ctx* c, const float* d_xyz, int64_t stride_floats, int64_t n, double cell, int64_t n_words, hipStream_t st,
  VoxDesc* host_desc, bool with_lattice, int* cloud_off_out)
{
  const int64_t cap_words = n_words;
  if (n == 0)
    n_words = 0;  // (no point, no bit: nothing to clear, count or emit; the block counts are not even written)
  const int64_t nb2 = n_words / kWordsPerBlock;
  if (with_lattice)  // speculative pass: clear + lattice in one launch (n_words is a multiple of 4096 words)
    hipLaunchKernelGGL(k_vox_clear_lattice, dim3((unsigned) std::max<int64_t>(1, std::min<int64_t>(n_words / 4 / 256, 2048))), dim3(256), 0,
      st, reinterpret_cast<uint4*>(c->d_vox_bitmap), n_words / 4, c->d_vox_desc, cell, (unsigned long long) kVoxMaxWords,
      (unsigned long long) cap_words, host_desc);
  else if (hipMemsetAsync(c->d_vox_bitmap, 0, (size_t) n_words * 4, st) != hipSuccess)
    return AGH_ERR_HIP;
  if (n > 0 && n_words > 0)
  {
    hipLaunchKernelGGL(k_vox_mark, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, d_xyz, stride_floats, n,
      (const uint8_t*) c->d_vox_code, (const VoxDesc*) c->d_vox_desc, cell, c->d_vox_bitmap);
    hipLaunchKernelGGL(k_vox_popcount, dim3((unsigned) nb2), dim3(256), 0, st, (const unsigned*) c->d_vox_bitmap,
      c->d_vox_blk2);
  }
  hipLaunchKernelGGL(k_vox_scan, dim3(1), dim3(1024), 0, st, c->d_vox_blk2, nb2, c->d_vox_total, (VoxDesc*) nullptr, c->d_vox_desc,
    host_desc, cloud_off_out);  // (+ the voxel counts per camera, the host mirror, the device-side cloud offsets)
  if (n > 0 && n_words > 0)
    hipLaunchKernelGGL(k_vox_emit, dim3((unsigned) nb2), dim3(256), 0, st, (const unsigned*) c->d_vox_bitmap,
      (const int*) c->d_vox_blk2, (const VoxDesc*) c->d_vox_desc, cell, c->d_vox_xyz, c->d_vox_cam);
  return hipGetLastError() == hipSuccess ? AGH_OK : AGH_ERR_HIP;
}

}  // namespace agh
