// taubin.hip -- K1: radius-neighbour gather + Taubin quadric fit + local Darboux frame, per sample.
//
// Reference path (src/agile_grasp/...): HandSearch::findQuadrics hand_search.cpp:65-113 (OMP loop A) ->
// kdtree.radiusSearch (85) -> Quadric::fitQuadric quadric.cpp:14-157 -> findTaubinNormalAxis 159-251 ->
// findAverageNormalAxis 263-305.
//
// Three kernels, one launch each per batch of samples:
//   k_taubin_moments<CAP>  one 256-thread workgroup per sample: coalesced float4 reads of the cell-sorted cloud
//                          for the <= 16 chord-clipped grid rows the ball touches (row per wave, loads one segment
//                          ahead), FLANN float32 distance filter, LDS compaction, LDS bucket sort into the radius
//                          search's (d2, index) order, then the 37 distinct sums behind M and N accumulated
//                          SEQUENTIALLY in that order (56-neighbour chunks: all four waves form the products, 37 lanes
//                          run the 37 dependent add chains) -- the same fp64 operation order as the reference loop, so
//                          the sums are bit-identical to the CPU path.
//   k_taubin_eigen         one sample per LANE (taubin_eigen.h): eliminates the 10th unknown, Cholesky of N9 with deflation of
//                          rank-deficient coordinates, in-place reduction, Householder tridiagonalisation, bisection,
//                          twisted factorisation -> the ONE eigenpair the reference uses -> quadric parameters; every
//                          multiply-add fused, the oracle's loop nest verbatim.  One extra work-group sorts the samples
//                          longest-first.
//   k_taubin_frame         one workgroup per sample: quadric-gradient normals; the 3x3 scatter of the normals and the
//                          n x n (n_i . n_j)^6 column sums in the oracle's LaneSum64 order (64 interleaved partials +
//                          tree), the latter only for the columns a moment-based estimate cannot rule out; argmax,
//                          projection, camera orientation.
// No MFMA: fp64 separately-rounded mul/add is required for bit parity (MFMA fuses), and the work is LDS/VALU bound.
#include "agh_internal.h"

#include <cstring>
#include <utility>

#include "taubin_eigen.h"

namespace agh
{

constexpr int kSortBins = 256;  // distance buckets of the neighbour sort (= workgroup size)
constexpr int kChunk = 56;      // neighbours per summation chunk (56 x 37 doubles: the term tile shares the LDS of the dead sort scratch)

// ---------------------------------------------------------------------------------------------------------------
// K1a
// ---------------------------------------------------------------------------------------------------------------
// Capacity classes: 1152 neighbours is what a two-view cloud voxelised at the reference's 3 mm holds in a 3 cm ball at most
// (C2 / C4 / C5: median 560, 99th percentile 980, maximum 1132), and it is the largest class whose 38.4 KB of LDS and 128
// VGPRs leave room for FOUR work-groups per CU (the 1536 class it replaces ran three); denser clouds fall through to 4096.
// (the work of one sample: the kernel below calls it once -- or, in the 4096 class, for every sample of a list)
template <int CAP>
__device__ __forceinline__ void taubin_moments_sample(const int s, GridView gv, const float* __restrict__ xyz, int64_t stride,
  const int32_t* __restrict__ samples, int S, float r2f, double rpad, int first_class, double* __restrict__ sums,
  int32_t* __restrict__ nt, int32_t* __restrict__ status, float4* __restrict__ nbr, int64_t nbr_stride, int debug_stop,
  int n_points, int32_t* __restrict__ zero_flags, int32_t* __restrict__ scloud, long long* __restrict__ dbg,
  int32_t* __restrict__ ovf_out)
{
#ifdef AGH_DEBUG_HOOKS  // scripts/moments_clocks.py: per-work-group phase timestamps (AGH_DEBUG_CLOCKS_KERNEL=moments)
#define AGH_MSTAMP(i) do { if (dbg && threadIdx.x == 0) dbg[(int64_t) blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
  long long mt_prod = 0, mt_cons = 0, mt_wait = 0, mt_last = 0;
#define AGH_MLAP(acc) do { if (dbg) { const long long now_ = wall_clock64(); acc += now_ - mt_last; mt_last = now_; } } while (0)
#else
#define AGH_MSTAMP(i) do { } while (0)
#define AGH_MLAP(acc) do { } while (0)
#endif
  AGH_MSTAMP(0);
  // LDS: the staged neighbours and their sorted order live for the whole kernel; the sort scratch (keys, bucket
  // permutation, histogram) is dead once `slot` is known, so the term tile of the summation phase reuses its space.
  constexpr int kSortBytes = CAP * 8 + CAP * 2 + (kSortBins + 1) * 4 + kSortBins * 4;
  constexpr int kTS = kChunk + 2;  // row stride of the term-major tile (doubles): spreads the chain's lanes over the LDS banks
  constexpr int kTermBytes = kNumSums * kTS * 8;
  constexpr int kScratch = ((kSortBytes > kTermBytes ? kSortBytes : kTermBytes) + 15) & ~15;
  __shared__ float4 stage[CAP];
  __shared__ unsigned short slot[CAP];
  __shared__ __attribute__((aligned(16))) unsigned char scratch[kScratch];
  __shared__ RowTable rt;
  __shared__ int count;
  __shared__ int wsum[4];
  unsigned long long* key = reinterpret_cast<unsigned long long*>(scratch);
  unsigned short* perm = reinterpret_cast<unsigned short*>(scratch + CAP * 8);
  int* hist = reinterpret_cast<int*>(scratch + CAP * 8 + CAP * 2);
  int* fillc = hist + (kSortBins + 1);
  double* termbuf = reinterpret_cast<double*>(scratch);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (zero_flags && blockIdx.x == 0 && tid < 8)
    zero_flags[tid] = 0;  // first kernel of an agh_find_hands call: the flags are only set by later kernels
  if (!first_class && status[s] != kStatusOverflow)
    return;  // an earlier (smaller) capacity class already handled this sample
  // (n_points < 0: the host only knows a bound -- agh_localize, whose cloud is still being voxelised when this launch is queued --
  // and the count is the last of the device-side cloud offsets)
  const int n_pts = n_points >= 0 ? n_points : gv.cloud_off[gv.n_clouds];
  if (samples[s] < 0 || samples[s] >= n_pts)
  {
    if (threadIdx.x == 0)
    {
      // kSampleSkip: a slot of a device-drawn sample list beyond what the cloud could fill (fewer points than samples): no
      // frame, no hypotheses, no error
      status[s] = samples[s] == kSampleSkip ? kStatusSkipped : kStatusBadIndex;
      nt[s] = 0;
    }
    return;
  }
  const float* qp = xyz + (int64_t) samples[s] * stride;
  const float qx = qp[0], qy = qp[1], qz = qp[2];
  {  // the sample's cloud of the batch: its grid is the one every query of this sample walks
    const int cloud = cloud_of_point(gv, samples[s]);
    gv = grid_of_cloud(gv, cloud);
    if (first_class && tid == 0)
      scloud[s] = cloud;
  }
  // A sample AT a non-finite point (reachable only by calling the search on a raw capture: Localization removes NaNs first,
  // localization.cpp:27): PCL's kd-tree holds no such point and its radiusSearch asserts on such a query; here, as for an empty
  // neighbourhood: no frame, no hypotheses, no error.
  if (!(isfinite(qx) && isfinite(qy) && isfinite(qz)))
  {
    if (tid == 0)
    {
      status[s] = kStatusOk;
      nt[s] = 0;
    }
    return;
  }
  if (tid == 0)
    count = 0;
  for (int k = tid; k < kSortBins; k += 256)
  {
    hist[k] = 0;
    fillc[k] = 0;
  }
  build_rows(gv, qx, qy, qz, rpad, rt);
  if (rt.bad)
  {
    if (tid == 0)
    {
      status[s] = kStatusRows;
      nt[s] = 0;
    }
    return;
  }
  AGH_MSTAMP(1);
  const float binscale = (float) kSortBins / r2f;  // monotone map of d2 in [0, r2f) onto the sort bins
  // ---- gather + FLANN distance filter + compaction into LDS ----
  // every wave owns the 128-candidate segments j = wave, wave + 4, ... of the concatenated grid rows (FlatRows,
  // agh_internal.h: full segments, an equal share per wave, the row table in lane registers) and the reads are coalesced
  // runs.  The walk is software-pipelined like k_hand_sweep's: the (unconditional) loads of the next 128 candidates are
  // issued before the current 128 are filtered, and the two candidates of a lane share one reservation.
  {
    FlatRows flat;
    flat.init(rt, lane);
    const int nseg = flat.segments();
    auto seg_load = [&](int j, int& r0, float4& p0, float4& p1, bool& h0, bool& h1) {
      int a0, a1;
      flat.locate(j, r0, lane, a0, a1, h0, h1);
      p0 = gv.sorted[h0 ? a0 : 0];  // a lane without a candidate reads element 0 and ignores it
      p1 = gv.sorted[h1 ? a1 : 0];
    };
    auto take2 = [&](const float4& p0, bool h0, const float4& p1, bool h1) {
      const float d0 = flann_d2(qx, qy, qz, p0.x, p0.y, p0.z), d1 = flann_d2(qx, qy, qz, p1.x, p1.y, p1.z);
      const bool s0 = h0 & (d0 < r2f), s1 = h1 & (d1 < r2f);  // (bitwise: no exec-mask regions around a compare)
      const unsigned long long m0 = __ballot(s0), m1 = __ballot(s1);
      if (m0 | m1)
      {
        const int c0 = __popcll(m0);
        int base = 0;
        if (lane == 0)
          base = atomicAdd(&count, c0 + __popcll(m1));
        base = __builtin_amdgcn_readfirstlane(base);
        const unsigned long long below = (1ull << lane) - 1ull;
        const int k0 = base + __popcll(m0 & below), k1 = base + c0 + __popcll(m1 & below);
        if (s0 && k0 < CAP)
        {
          stage[k0] = p0;
          key[k0] = ((unsigned long long) __float_as_uint(d0) << 32) | (unsigned long long) __float_as_uint(p0.w);
          atomicAdd(&hist[min(kSortBins - 1, (int) (d0 * binscale))], 1);
        }
        if (s1 && k1 < CAP)
        {
          stage[k1] = p1;
          key[k1] = ((unsigned long long) __float_as_uint(d1) << 32) | (unsigned long long) __float_as_uint(p1.w);
          atomicAdd(&hist[min(kSortBins - 1, (int) (d1 * binscale))], 1);
        }
      }
    };
    int cur_j = wave, r0 = 0;
    float4 p0, p1;
    bool h0, h1;
    seg_load(cur_j, r0, p0, p1, h0, h1);
    while (cur_j < nseg)
    {
      float4 q0, q1;
      bool g0, g1;
      seg_load(cur_j + 4, r0, q0, q1, g0, g1);  // in flight while the current segment is filtered
      take2(p0, h0, p1, h1);
      cur_j += 4;
      p0 = q0;
      p1 = q1;
      h0 = g0;
      h1 = g1;
    }
  }
  __syncthreads();
  const int n = count;
  if (n > CAP)
  {
    if (tid == 0)
    {
      status[s] = kStatusOverflow;
      nt[s] = n;
      if (ovf_out)
        ovf_out[1 + atomicAdd(&ovf_out[0], 1)] = s;
    }
    return;
  }
  AGH_MSTAMP(2);
  if (debug_stop == 1)
    return;
  // ---- sort into FLANN's sorted radius-search order: ascending (d2, index) ----
  // Distance-bucket counting sort (d2 is ~uniform over the ball's bins for surface data), then an exact rank inside
  // each bucket by comparing the full 64-bit keys (d2 bits, index) -- a few comparisons per point, three barriers.
  {
    const int lane_ = tid & 63, w_ = tid >> 6;
    const int v = hist[tid];  // kSortBins == blockDim.x
    const int inc = wave_incl_scan_i32(v);
    if (lane_ == 63)
      wsum[w_] = inc;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < w_; k++)
      base += wsum[k];
    hist[tid] = base + inc - v;  // exclusive start of bin tid
    if (tid == 0)
      hist[kSortBins] = n;
    __syncthreads();
  }
  {
    // (a thread's <= CAP / 256 elements together: key reads, then the reservations, then the writes -- three LDS round trips in all
    // instead of three per element)
    constexpr int kPerThread = CAP <= 1152 ? (CAP + 255) / 256 : 1;
    for (int k0 = tid; k0 < n; k0 += 256 * kPerThread)
    {
      int bb[kPerThread], pos[kPerThread];
#pragma unroll
      for (int u = 0; u < kPerThread; u++)
      {
        const int k = k0 + 256 * u;
        const float d2 = __uint_as_float((unsigned) (key[k < n ? k : k0] >> 32));
        bb[u] = min(kSortBins - 1, (int) (d2 * binscale));
      }
#pragma unroll
      for (int u = 0; u < kPerThread; u++)
        pos[u] = (k0 + 256 * u < n) ? hist[bb[u]] + atomicAdd(&fillc[bb[u]], 1) : 0;
#pragma unroll
      for (int u = 0; u < kPerThread; u++)
        if (k0 + 256 * u < n)
          perm[pos[u]] = (unsigned short) (k0 + 256 * u);
    }
  }
  __syncthreads();
  for (int g = tid; g < n; g += 256)
  {
    const int k = perm[g];
    const unsigned long long mine = key[k];
    const float d2 = __uint_as_float((unsigned) (mine >> 32));
    const int b = min(kSortBins - 1, (int) (d2 * binscale));
    const int bs = hist[b], be = hist[b + 1];
    int rank = bs;
    for (int g2 = bs; g2 < be; g2 += 4)  // four members of the bucket at a time: two LDS round trips per four, not per member
    {
      unsigned short pk[4];
#pragma unroll
      for (int u = 0; u < 4; u++)
        pk[u] = perm[min(g2 + u, be - 1)];
      unsigned long long kk[4];
#pragma unroll
      for (int u = 0; u < 4; u++)
        kk[u] = key[pk[u]];
#pragma unroll
      for (int u = 0; u < 4; u++)
        rank += (g2 + u < be && kk[u] < mine) ? 1 : 0;
    }
    slot[rank] = (unsigned short) k;
  }
  __syncthreads();
  if (debug_stop == 2)
    return;
  AGH_MSTAMP(3);
  // ---- sorted neighbour list to global (consumed by k_taubin_frame) ----
  {
    constexpr int kPerThread = CAP <= 1152 ? (CAP + 255) / 256 : 1;
    for (int i0 = tid; i0 < n; i0 += 256 * kPerThread)
    {
      unsigned short sl[kPerThread];
#pragma unroll
      for (int u = 0; u < kPerThread; u++)
        sl[u] = slot[min(i0 + 256 * u, n > 0 ? n - 1 : 0)];
      float4 pv[kPerThread];
#pragma unroll
      for (int u = 0; u < kPerThread; u++)
        pv[u] = stage[sl[u]];
#pragma unroll
      for (int u = 0; u < kPerThread; u++)
        if (i0 + 256 * u < n)
          nbr[(int64_t) s * nbr_stride + i0 + 256 * u] = pv[u];
    }
  }
  // ---- 37 sequential sums (quadric.cpp:40-131) ----
  double acc = 0.0;
#ifdef AGH_DEBUG_HOOKS
  mt_last = wall_clock64();
#endif
  // (the chunk's point is fetched a chunk ahead: the two-deep look-up stage[slot[.]] then hides behind the chain of the
  // chunk before instead of opening every chunk)
  const int n_last = n > 0 ? n - 1 : 0;
  float4 p_next = stage[slot[min(lane, n_last)]];
  for (int c0 = 0; c0 < n; c0 += kChunk)
  {
    const int rows = min(kChunk, n - c0);
    const float4 p = p_next;
    p_next = stage[slot[min(c0 + kChunk + lane, n_last)]];
    // TWO waves form the 37 terms of a chunk (19 + 18), not four (10 + 9 + 9 + 9): every producing wave converts the point and
    // forms the six squares and mixed products for itself, so four producers issue 4 x 9 + 31 fp64 instructions per chunk where
    // two issue 2 x 9 + 31 -- and the phase is bound by the fp64 issue slots of a CU that runs sixteen such waves.
    if (lane < rows && wave < 2)
    {
      const double x = (double) p.x, y = (double) p.y, z = (double) p.z;
      const double x2 = x * x, y2 = y * y, z2 = z * z;
      const double xy = x * y, yz = y * z, xz = x * z;
      double* t = &termbuf[lane];  // term-major tile: term k of row r at [k * kTS + r]
      if (wave == 0)
      {
        t[0 * kTS] = x2 * x2;
        t[1 * kTS] = x2 * y2;
        t[2 * kTS] = x2 * z2;
        t[3 * kTS] = x2 * xy;
        t[4 * kTS] = x2 * yz;
        t[5 * kTS] = x2 * xz;
        t[6 * kTS] = x2 * x;
        t[7 * kTS] = x2 * y;
        t[8 * kTS] = x2 * z;
        t[9 * kTS] = x2;
        t[10 * kTS] = y2 * y2;
        t[11 * kTS] = y2 * z2;
        t[12 * kTS] = y2 * xy;
        t[13 * kTS] = y2 * yz;
        t[14 * kTS] = y2 * xz;
        t[15 * kTS] = y2 * x;
        t[16 * kTS] = y2 * y;
        t[17 * kTS] = y2 * z;
        t[18 * kTS] = y2;
      }
      else
      {
        t[19 * kTS] = z2 * z2;
        t[20 * kTS] = z2 * xy;
        t[21 * kTS] = z2 * yz;
        t[22 * kTS] = z2 * xz;
        t[23 * kTS] = z2 * x;
        t[24 * kTS] = z2 * y;
        t[25 * kTS] = z2 * z;
        t[26 * kTS] = z2;
        t[27 * kTS] = x * yz;
        t[28 * kTS] = xy;
        t[29 * kTS] = yz;
        t[30 * kTS] = xz;
        t[31 * kTS] = x;
        t[32 * kTS] = y;
        t[33 * kTS] = z;
        t[34 * kTS] = x2 + y2;
        t[35 * kTS] = y2 + z2;
        t[36 * kTS] = x2 + z2;
      }
    }
    __syncthreads();
    AGH_MLAP(mt_prod);
    if (wave == 0 && lane < kNumSums && rows == kChunk)
    {
      // full chunk: all 56 loads are issued up front (two register halves), so the dependent add chain -- the critical
      // path of the block -- pays one LDS round trip per chunk instead of one per eight rows
      // (term-major tile: a lane's 56 terms are 28 sixteen-byte loads; with the row-major tile they were 56 eight-byte loads
      // and issuing them was half of the chain's time)
      double2 va[kChunk / 2];
#pragma unroll
      for (int u = 0; u < kChunk / 2; u++)
        va[u] = reinterpret_cast<const double2*>(termbuf + lane * kTS)[u];
#pragma unroll
      for (int u = 0; u < kChunk / 2; u++)
      {
        acc += va[u].x;
        acc += va[u].y;
      }
    }
    else if (wave == 0 && lane < kNumSums)
    {
      int k = 0;
      for (; k + 8 <= rows; k += 8)  // 8 LDS loads in flight; the adds stay in neighbour order
      {
        double v_[8];
#pragma unroll
        for (int u = 0; u < 8; u++)
          v_[u] = termbuf[lane * kTS + k + u];
#pragma unroll
        for (int u = 0; u < 8; u++)
          acc += v_[u];
      }
      for (; k < rows; k++)
        acc += termbuf[lane * kTS + k];
    }
#ifdef AGH_DEBUG_HOOKS
    if (dbg && acc == 1.2345e300)  // (keeps the adds in front of the clock read)
      mt_cons++;
#endif
    AGH_MLAP(mt_cons);
    __syncthreads();
    AGH_MLAP(mt_wait);
  }
  if (wave == 0 && lane < kNumSums)
    sums[(int64_t) s * kSumStride + lane] = acc;
  if (tid == 0)
  {
    nt[s] = n;
    status[s] = kStatusOk;
  }
  AGH_MSTAMP(4);
#ifdef AGH_DEBUG_HOOKS
  if (dbg && threadIdx.x == 0)
  {
    dbg[(int64_t) blockIdx.x * 8 + 5] = mt_prod;
    dbg[(int64_t) blockIdx.x * 8 + 6] = mt_cons;
    dbg[(int64_t) blockIdx.x * 8 + 7] = mt_wait;
  }
#endif
#undef AGH_MSTAMP
#undef AGH_MLAP
}

// ovf_out (the 1152 class): the samples it hands on -- more than 1152 neighbours -- are LISTED ({count, samples...}), and the
// 4096 class (ovf_list) is launched as a few hundred work-groups that walk that list instead of one work-group per sample of
// which nearly all return at once.  At 100 KB of LDS a CU holds one such work-group, so the 16 000 empty ones of a batch of
// eight clouds (37 listed samples) were 50 us of the launch, and as many again in k_taubin_frame (round 6).
template <int CAP>
__global__ __launch_bounds__(256, CAP <= 1152 ? 4 : 1) void k_taubin_moments(GridView gv, const float* __restrict__ xyz, int64_t stride,
  const int32_t* __restrict__ samples, int S, float r2f, double rpad, int first_class, double* __restrict__ sums,
  int32_t* __restrict__ nt, int32_t* __restrict__ status, float4* __restrict__ nbr, int64_t nbr_stride, int debug_stop,
  int n_points, int32_t* __restrict__ zero_flags, int32_t* __restrict__ scloud, long long* __restrict__ dbg,
  int32_t* __restrict__ ovf_out, const int32_t* __restrict__ ovf_list)
{
  if constexpr (CAP == 4096)  // (only this instantiation carries the loop: in the others it cost registers they do not have)
  {
    const int n_items = ovf_list ? min(ovf_list[0], S) : S;
    for (int it = blockIdx.x; it < n_items; it += gridDim.x)
    {
      taubin_moments_sample<CAP>(ovf_list ? ovf_list[1 + it] : it, gv, xyz, stride, samples, S, r2f, rpad, first_class, sums, nt, status,
        nbr, nbr_stride, debug_stop, n_points, zero_flags, scloud, dbg, ovf_out);
      if (it + (int) gridDim.x < n_items)
        __syncthreads();  // (the next sample reuses the tiles)
    }
  }
  else
    taubin_moments_sample<CAP>((int) blockIdx.x, gv, xyz, stride, samples, S, r2f, rpad, first_class, sums, nt, status, nbr, nbr_stride,
      debug_stop, n_points, zero_flags, scloud, dbg, ovf_out);
}

// ---------------------------------------------------------------------------------------------------------------
// K1a beyond the LDS-resident classes: neighbourhoods of more than 4096 points (un-voxelised captures: kdtree.radiusSearch has
// max_nn = 0, hand_search.cpp:85).  They do not fit the kernel above (28 bytes of LDS per neighbour), and they are rare, so this
// class trades speed for room: a first walk over the ball counts its points, the work-group reserves that many entries of a
// global pool (kHugeEntries for all such neighbourhoods of a launch), a second walk stores the points and their (d2, index)
// keys there, every point finds its place in FLANN's order by counting the smaller keys (tiles of the key array through LDS:
// n^2 / 256 comparisons per thread, ~0.3 ms for 6000 points), the sorted list goes to the sample's neighbour list if it fits
// (n <= nbr_stride: k_taubin_frame<6144> then works on it like the other classes do) or to the pool (k_taubin_frame_huge), and the
// 37 sums run over it in that order with the term tile and the one-wave add chain of the kernel above -- the same products, the
// same additions in the same order: bit-identical sums.  Launched only once a call has met such a neighbourhood
// (Ctx::huge_classes), one work-group per sample, of which all but the flagged ones return at once.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_taubin_moments_huge(GridView gv, const float* __restrict__ xyz, int64_t stride,
  const int32_t* __restrict__ samples, int S, float r2f, double rpad, double* __restrict__ sums, int32_t* __restrict__ nt,
  int32_t* __restrict__ status, float4* __restrict__ nbr, int64_t nbr_stride, float4* __restrict__ pool_stage,
  unsigned long long* __restrict__ pool_key, float4* __restrict__ pool_sorted, long long* __restrict__ huge_base,
  unsigned long long* __restrict__ pool_count)
{
  constexpr int kTS = kChunk + 2;
  constexpr int kKeyTile = 2048;
  __shared__ __attribute__((aligned(16))) double termbuf[kNumSums * kTS];
  __shared__ unsigned long long ktile[kKeyTile];
  __shared__ RowTable rt;
  __shared__ int count, n_s;
  __shared__ long long base_s;
  const int s = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0)
    huge_base[s] = -1;
  if (status[s] != kStatusOverflow)
    return;  // (an earlier class handled this sample, or it has no neighbourhood)
  if (tid == 0)
    count = 0;
  const float* qp = xyz + (int64_t) samples[s] * stride;
  const float qx = qp[0], qy = qp[1], qz = qp[2];
  gv = grid_of_cloud(gv, cloud_of_point(gv, samples[s]));
  build_rows(gv, qx, qy, qz, rpad, rt);  // (ends with barriers)
  if (rt.bad)
    return;  // (the status stays kStatusOverflow: loud)
  // ---- how many points the ball holds, then that many entries of the pool ----
  {
    int mine = 0;
    for (int j = tid; j < rt.total; j += 256)
    {
      const float4 p = gv.sorted[row_lookup(rt, j)];
      mine += flann_d2(qx, qy, qz, p.x, p.y, p.z) < r2f ? 1 : 0;
    }
    mine = wave_allsum_i32(mine);
    if (lane == 0 && mine)
      atomicAdd(&count, mine);
  }
  __syncthreads();
  // (`count` becomes the second walk's cursor below: the first walk's total is published through a word of its own, so that no
  // wave can read the counter after thread 0 has cleared it -- the barrier above orders the atomics, not that store)
  if (tid == 0)
  {
    const int n0 = count;
    n_s = n0;
    const unsigned long long b0 = atomicAdd(pool_count, (unsigned long long) n0);
    base_s = b0 + (unsigned long long) n0 <= (unsigned long long) kHugeEntries ? (long long) b0 : -1;
    count = 0;
  }
  __syncthreads();
  const int n = n_s;
  const long long base = base_s;
  if (base < 0)
  {
    if (tid == 0)
      nt[s] = n;  // (the pool is exhausted: the status stays kStatusOverflow)
    return;
  }
  float4* stage = pool_stage + base;
  unsigned long long* key = pool_key + base;
  // ---- the ball's points and their keys, in any order ----
  for (int j = tid; j < rt.total; j += 256)
  {
    const float4 p = gv.sorted[row_lookup(rt, j)];
    const float d2 = flann_d2(qx, qy, qz, p.x, p.y, p.z);
    if (d2 < r2f)
    {
      const int k = atomicAdd(&count, 1);
      stage[k] = p;
      key[k] = ((unsigned long long) __float_as_uint(d2) << 32) | (unsigned long long) __float_as_uint(p.w);
    }
  }
  __threadfence();
  __syncthreads();
  // the sorted list: the sample's own neighbour list if it fits, else the pool
  const bool in_pool = n > nbr_stride;
  float4* const list_w = in_pool ? pool_sorted + base : nbr + (int64_t) s * nbr_stride;
  if (tid == 0)
    huge_base[s] = in_pool ? base : -1;
  // ---- FLANN's sorted order, ascending (d2, index): a point's place is the number of smaller keys (the keys are distinct) ----
  for (int g0 = 0; g0 < n; g0 += 256 * 8)  // eight points per thread at a time
  {
    constexpr int kPer = 8;
    unsigned long long mine[kPer];
    int rank[kPer];
#pragma unroll
    for (int u = 0; u < kPer; u++)
    {
      const int g = g0 + tid + 256 * u;
      mine[u] = g < n ? key[g] : ~0ull;
      rank[u] = 0;
    }
    for (int t0 = 0; t0 < n; t0 += kKeyTile)
    {
      const int tn = min(kKeyTile, n - t0);
      __syncthreads();
      for (int k = tid; k < tn; k += 256)
        ktile[k] = key[t0 + k];
      __syncthreads();
      for (int k = 0; k < tn; k++)
      {
        const unsigned long long kk = ktile[k];  // (the same word for every lane: an LDS broadcast)
#pragma unroll
        for (int u = 0; u < kPer; u++)
          rank[u] += kk < mine[u] ? 1 : 0;
      }
    }
#pragma unroll
    for (int u = 0; u < kPer; u++)
    {
      const int g = g0 + tid + 256 * u;
      if (g < n)
        list_w[rank[u]] = stage[g];
    }
  }
  __threadfence();
  __syncthreads();
  // ---- 37 sequential sums (quadric.cpp:40-131) over the sorted list: the term tile and add chain of k_taubin_moments ----
  const float4* list = list_w;
  double acc = 0.0;
  for (int c0 = 0; c0 < n; c0 += kChunk)
  {
    const int rows = min(kChunk, n - c0);
    if (lane < rows && wave < 2)
    {
      const float4 p = list[c0 + lane];
      const double x = (double) p.x, y = (double) p.y, z = (double) p.z;
      const double x2 = x * x, y2 = y * y, z2 = z * z;
      const double xy = x * y, yz = y * z, xz = x * z;
      double* t = &termbuf[lane];  // term-major tile: term k of row r at [k * kTS + r]
      if (wave == 0)
      {
        t[0 * kTS] = x2 * x2;
        t[1 * kTS] = x2 * y2;
        t[2 * kTS] = x2 * z2;
        t[3 * kTS] = x2 * xy;
        t[4 * kTS] = x2 * yz;
        t[5 * kTS] = x2 * xz;
        t[6 * kTS] = x2 * x;
        t[7 * kTS] = x2 * y;
        t[8 * kTS] = x2 * z;
        t[9 * kTS] = x2;
        t[10 * kTS] = y2 * y2;
        t[11 * kTS] = y2 * z2;
        t[12 * kTS] = y2 * xy;
        t[13 * kTS] = y2 * yz;
        t[14 * kTS] = y2 * xz;
        t[15 * kTS] = y2 * x;
        t[16 * kTS] = y2 * y;
        t[17 * kTS] = y2 * z;
        t[18 * kTS] = y2;
      }
      else
      {
        t[19 * kTS] = z2 * z2;
        t[20 * kTS] = z2 * xy;
        t[21 * kTS] = z2 * yz;
        t[22 * kTS] = z2 * xz;
        t[23 * kTS] = z2 * x;
        t[24 * kTS] = z2 * y;
        t[25 * kTS] = z2 * z;
        t[26 * kTS] = z2;
        t[27 * kTS] = x * yz;
        t[28 * kTS] = xy;
        t[29 * kTS] = yz;
        t[30 * kTS] = xz;
        t[31 * kTS] = x;
        t[32 * kTS] = y;
        t[33 * kTS] = z;
        t[34 * kTS] = x2 + y2;
        t[35 * kTS] = y2 + z2;
        t[36 * kTS] = x2 + z2;
      }
    }
    __syncthreads();
    if (wave == 0 && lane < kNumSums)
      for (int k = 0; k < rows; k++)  // (the adds stay in neighbour order)
        acc += termbuf[lane * kTS + k];
    __syncthreads();
  }
  if (wave == 0 && lane < kNumSums)
    sums[(int64_t) s * kSumStride + lane] = acc;
  if (tid == 0)
  {
    nt[s] = n;
    status[s] = kStatusOk;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// K1b: quadric.cpp:134-153 + solveGeneralizedEigenProblem (330-363): the ONE eigenpair the reference uses of
// M v = lambda N v (the smallest of "the first nine"), by the scheme the oracle states (solve_taubin): elimination of
// the 10th unknown, Cholesky of N9 with deflation of rank-deficient coordinates, in-place reduction C = L^-1 S L^-T,
// Householder tridiagonalisation, bisection on the Sturm sequence, twisted factorisation, back-transformation.
//
// ONE SAMPLE PER LANE.  The solve is a chain of ~4000 dependent-ish fp64 operations on a 9 x 9 matrix; spreading one
// matrix over lanes (round 2: nine lanes per sample, DPP broadcasts) bought a 4x shorter chain at 16x the instruction
// count, and left 63 of 64 lanes idle in the scalar phases.  With a sample per lane every instruction does 64 samples'
// work, every index is a compile-time constant (the matrices live in registers: <= 256 VGPRs), there is no LDS and no
// cross-lane traffic, and the arithmetic is the oracle's loop nest verbatim -- the same roundings by construction.
// 2000 samples are 32 waves; the kernel's time is one lane's chain, whatever the sample count (until the waves outnumber
// the SIMDs several times over: the 300 000 samples of the all-points pass are 4688 waves).
// ---------------------------------------------------------------------------------------------------------------
// Longest-first scheduling of k_taubin_frame: a sample's cost there grows with its Taubin neighbourhood n_t, and the
// samples arrive sorted by index, i.e. spatially coherent, so dense regions form runs of slow work-groups and a run
// that starts late is a long tail.  Work-groups are dispatched in blockIdx order; `order` maps blockIdx -> sample by
// descending n_t (counting sort, bins of 16 neighbours).  It is computed by ONE EXTRA work-group of this kernel:
// k_taubin_eigen leaves half the SIMDs idle, so the sort costs no time and no launch.  Results go to per-sample slots, so
// the processing order (and the arbitrary order inside a bin) is invisible in the output.  (k_hand_sweep used the same
// scheme with the candidate count of the hand-search ball as weight until its gather was clipped to the hand's slab; it
// now runs in sample order, see there.)
constexpr int kOrderBins = 2048;
constexpr int kOrderMaxSamples = 4096;  // scheduling orders are made for launches of at most this many samples
__device__ __forceinline__ int order_bin(int w) { return kOrderBins - 1 - min(w >> 4, kOrderBins - 1); }  // 0 = heaviest

__device__ void sample_order_block(const int* __restrict__ weight, int S, int* __restrict__ order, int* hist, int* wave_tot)
{
  // One work-group of 256 threads walks the S weights twice.  Every step of the walk is a global-load round trip, so a
  // thread takes kPer weights per step with all its loads in flight together (one weight per step and one wave took
  // 2 x S / 64 round trips: 140 us at S = 8000 -- the sorter WAS k_taubin_eigen's duration at C4 and in the batch; with
  // the solver at 10 us it was again, at 61 us, until the walk got four waves).
  constexpr int kPer = 16, kT = 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int b = tid; b < kOrderBins; b += kT)
    hist[b] = 0;
  __syncthreads();
  for (int base = 0; base < S; base += kT * kPer)
  {
    int bin[kPer];
#pragma unroll
    for (int u = 0; u < kPer; u++)
    {
      const int s = base + u * kT + tid;
      bin[u] = s < S ? order_bin(weight[s]) : -1;
    }
#pragma unroll
    for (int u = 0; u < kPer; u++)
      if (bin[u] >= 0)
        atomicAdd(&hist[bin[u]], 1);
  }
  __syncthreads();
  constexpr int per = kOrderBins / kT;  // consecutive bins per thread
  int loc = 0;
  for (int k = 0; k < per; k++)
    loc += hist[tid * per + k];
  int incl = loc;
  incl = wave_incl_scan_i32(incl);
  if (lane == 63)
    wave_tot[wave] = incl;
  __syncthreads();
  int run = incl - loc;
  for (int w = 0; w < wave; w++)
    run += wave_tot[w];
  for (int k = 0; k < per; k++)
  {
    const int c = hist[tid * per + k];
    hist[tid * per + k] = run;
    run += c;
  }
  __syncthreads();
  for (int base = 0; base < S; base += kT * kPer)
  {
    int bin[kPer], pos[kPer];
#pragma unroll
    for (int u = 0; u < kPer; u++)
    {
      const int s = base + u * kT + tid;
      bin[u] = s < S ? order_bin(weight[s]) : -1;
    }
#pragma unroll
    for (int u = 0; u < kPer; u++)
      pos[u] = bin[u] >= 0 ? atomicAdd(&hist[bin[u]], 1) : 0;
#pragma unroll
    for (int u = 0; u < kPer; u++)
      if (bin[u] >= 0)
        order[pos[u]] = base + u * kT + tid;
  }
}

// blockIdx -> sample map of k_hand_sweep (round 4): BLOCKS of 32 consecutive samples, heaviest block first by the sum of the
// Taubin neighbour counts, order kept inside a block.  A sweep work-group's duration follows the local point density (its
// correlation with n_t is 0.56 per sample, 0.9 per block of 64); the samples arrive spatially sorted, so dense regions are runs
// of slow work-groups and a run that is dispatched late is the kernel's tail.  Whole blocks keep what neighbouring
// work-groups share in L2 (a per-sample longest-first order by n_t is no better than sample order; a random one costs 4 us).
// C2: 91.5 -> 85.2 us (scripts/sweep_order_experiment.py; longest-first by the MEASURED durations, which nothing known before
// the launch predicts, would be 76).  At most 1024 blocks (S <= 32768); beyond, the sweep runs in sample order.
constexpr int kSweepBlock = 32, kSweepBlocksMax = 1024;
__device__ void sweep_order_block(const int* __restrict__ weight, int S, int* __restrict__ order2, int* bw /* kSweepBlocksMax */)
{
  const int B = (S + kSweepBlock - 1) / kSweepBlock;
  if (B > kSweepBlocksMax)
    return;
  const int tid = threadIdx.x;
  __syncthreads();
  for (int b = tid; b < B; b += 256)
  {
    int w = 0;
    const int i0 = b * kSweepBlock, i1 = min(S, i0 + kSweepBlock);
    for (int i = i0; i < i1; i++)
      w += weight[i];
    bw[b] = w;
  }
  __syncthreads();
  const int last_size = S - (B - 1) * kSweepBlock;
  int rank_last = 0;  // rank of the (possibly short) last block
  for (int q = 0; q < B - 1; q++)
    rank_last += bw[q] >= bw[B - 1] ? 1 : 0;  // (earlier index wins ties)
  for (int b = tid; b < B; b += 256)
  {
    const int wb = bw[b];
    int rank = 0;
    for (int q = 0; q < B; q++)
      rank += (bw[q] > wb || (bw[q] == wb && q < b)) ? 1 : 0;
    const int pos = rank * kSweepBlock - (rank > rank_last ? kSweepBlock - last_size : 0);
    const int size = b == B - 1 ? last_size : kSweepBlock;
    for (int k = 0; k < size; k++)
      order2[pos + k] = b * kSweepBlock + k;
  }
}

// RAND50: sample i consumes 50 draws iff its neighbourhood has more than 50 points, in sample order (one wave: chunked
// inclusive scan; *total_io carries the draws consumed by earlier passes of this call).
__device__ void draw_offsets_wave(const int32_t* __restrict__ nt, int S, int32_t* __restrict__ draw_ofs,
  int32_t* __restrict__ total_io)
{
  // sixteen consecutive samples per lane and step (one wave scan per 1024 samples: with one sample per lane the 16 000
  // samples of a batch were 250 dependent scans, 100 us on the critical path of k_taubin_eigen's sorter work-group)
  constexpr int kPer = 16;
  const int lane = threadIdx.x;
  int carry = *total_io;
  for (int c0 = 0; c0 < S; c0 += 64 * kPer)
  {
    const int i0 = c0 + lane * kPer;
    int v[kPer], sum = 0;
#pragma unroll
    for (int u = 0; u < kPer; u++)
    {
      v[u] = (i0 + u < S && nt[i0 + u] > 50) ? 50 : 0;
      sum += v[u];
    }
    const int inc = wave_incl_scan_i32(sum);
    int run = carry + inc - sum;
#pragma unroll
    for (int u = 0; u < kPer; u++)
    {
      if (i0 + u < S)
        draw_ofs[i0 + u] = run;
      run += v[u];
    }
    carry += __builtin_amdgcn_readlane(inc, 63);
  }
  if (lane == 0)
    *total_io = carry;
}

template <int LPS>
__global__ __launch_bounds__(256) void k_taubin_eigen(const double* __restrict__ sums, const int32_t* __restrict__ nt,
  const int32_t* __restrict__ status, int S, double* __restrict__ eig, int32_t* __restrict__ flags,
  const int* __restrict__ weight, int* __restrict__ order, int32_t* __restrict__ draw_ofs, int32_t* __restrict__ draw_total_io,
  int* __restrict__ order_sweep)
{
  __shared__ int hist[kOrderBins];
  __shared__ int wave_tot[4];
  static_assert(kSweepBlocksMax <= kOrderBins, "the sweep's block weights reuse the histogram");
  const int n_solver_groups = (S * LPS + 255) / 256;
  // (order == nullptr: no scheduling orders -- beyond ~4000 samples a launch holds many rounds of work-groups, the orders buy
  // nothing there (k_taubin_frame 130.19 against 130.18 us at C4 with and without; the sweep's block order neutral at C4 and in
  // the batch), and the one sorter work-group WAS this kernel's duration: 32 us at C4's 8000 samples, 74 us at the batch's
  // 16 000, against ~17 us for the solver itself)
  // two extra work-groups: the scheduling orders of the two following kernels, side by side (one work-group made both, one after
  // the other, and was then the longest of the launch)
  if (order && (int) blockIdx.x == n_solver_groups)
  {
    sample_order_block(weight, S, order, hist, wave_tot);
    return;
  }
  if (order && (int) blockIdx.x == n_solver_groups + 1)
  {
    if (order_sweep)
      sweep_order_block(weight, S, order_sweep, hist);
    return;
  }
  if ((int) blockIdx.x >= n_solver_groups + (order ? 2 : 0))  // production mode, one more: the draw offsets k_taubin_frame needs ride along here
  {                                        // instead of in a launch of their own (12 us at C2: a launch for one wave)
    if (threadIdx.x < 64)
      draw_offsets_wave(nt, S, draw_ofs, draw_total_io);
    return;
  }
  // LPS consecutive lanes hold the same sample (taubin_eigen.h: they share the bisection); the first of them reports
  const int s = (int) (blockIdx.x * 256 + threadIdx.x) / LPS;
  const bool writer = (threadIdx.x % LPS) == 0;
  if (s >= S)
    return;
  // (the sums are read whatever the status says -- the record exists for every sample -- so that the reads do not wait for it)
  double sv[kNumSums];
#pragma unroll
  for (int k = 0; k < kNumSums; k++)
    sv[k] = sums[(int64_t) s * kSumStride + k];
  const int st_ = status[s];
  // loud capacity / index errors (read back by agh_synchronize and the host entry points)
  if (writer && (st_ == kStatusOverflow || st_ == kStatusRows))
    atomicOr(&flags[0], 1);
  if (writer && st_ == kStatusBadIndex)
    atomicOr(&flags[0], 4);
  const bool live = st_ == kStatusOk && nt[s] > 0;
#pragma unroll
  for (int k = 0; k < kNumSums; k++)
    sv[k] = live ? sv[k] : 0.0;
  double v[10];
  const double lambda = taubin_smallest_eigenpair<LPS>(sv, live ? (double) nt[s] : 1.0, v);
  if (!writer)
    return;
  double* out = eig + (int64_t) s * 12;
#pragma unroll
  for (int k = 0; k < 10; k++)
    out[k] = live ? ((k >= 3 && k < 6) ? v[k] * 0.5 : v[k]) : 0.0;  // quadric.cpp:153
  out[10] = live ? lambda : 0.0;
  out[11] = live ? 1.0 : 0.0;
}

// ---------------------------------------------------------------------------------------------------------------
// K1c
// ---------------------------------------------------------------------------------------------------------------
// 6! / (a! b! c!) in the order the 28 moments are kept (a = 6 .. 0, b = 6 - a .. 0, c = 6 - a - b): exact small integers
__device__ const double kMultinomial6[28] = { 1.0, 6.0, 6.0, 15.0, 30.0, 15.0, 20.0, 60.0, 60.0, 20.0, 15.0, 60.0, 90.0, 60.0, 15.0, 6.0, 30.0, 60.0, 60.0, 30.0, 6.0, 1.0, 6.0, 15.0, 20.0, 15.0, 6.0, 1.0 };

__device__ __forceinline__ double wave_max_f64_(double v)
{
  v = fmax(v, xor_partner_f64<32>(v));
  v = fmax(v, xor_partner_f64<16>(v));
  v = fmax(v, xor_partner_f64<8>(v));
  v = fmax(v, xor_partner_f64<4>(v));
  v = fmax(v, xor_partner_f64<2>(v));
  v = fmax(v, xor_partner_f64<1>(v));
  return v;
}

// THREADS = 256 (one work-group of four waves per sample) or 64 (one wave per sample: the class for at most 128 normals
// of the r = 0.01 all-points pass).
template <int CAP, int THREADS>
__global__ __launch_bounds__(THREADS, (CAP == 1152 && THREADS == 256) ? 5 : ((CAP == 64 && THREADS == 256) ? 8 : 1)) void k_taubin_frame(const float4* __restrict__ nbr, int64_t nbr_stride,
  const int32_t* __restrict__ nt, const double* __restrict__ eig, const int32_t* __restrict__ status,
  const float* __restrict__ xyz, int64_t stride, const int32_t* __restrict__ samples, int S, int rand_mode,
  const int32_t* __restrict__ draw_ofs, const int32_t* __restrict__ draws, double cam0x, double cam0y, double cam0z,
  double cam1x, double cam1y, double cam1z, agh_frame* __restrict__ frames, double* __restrict__ normals_out, int nmin, int debug_stop,
  const int* __restrict__ order, long long* __restrict__ dbg, const float4* __restrict__ pool_sorted,
  const long long* __restrict__ huge_base, const int32_t* __restrict__ ovf_list)
{
  // (ovf_list: the 4096 class walks the list of the samples beyond 1152 neighbours that k_taubin_moments<1152> made, a few hundred
  // work-groups instead of one per sample -- see there)
#ifdef AGH_DEBUG_HOOKS  // scripts/frame_clocks.py: per-work-group phase timestamps (AGH_DEBUG_CLOCKS_KERNEL=frame)
#define AGH_FSTAMP(i, t) do { if (dbg && threadIdx.x == (t)) dbg[(int64_t) (order ? order[blockIdx.x] : (int) blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define AGH_FSTAMP(i, t) do { } while (0)
#endif
  AGH_FSTAMP(0, 0);
  __shared__ double nx[CAP], ny[CAP], nz[CAP];
  __shared__ int camcnt[2];
  __shared__ int next_col;
  __shared__ double sM3[6];
  __shared__ double sAxis[3];
  constexpr int NW = THREADS / 64;
  __shared__ double wbest[NW];
  __shared__ int wbest_j[NW];
  __shared__ double sT[NW][28];
  __shared__ double sW[28];
  __shared__ double wmax_s[NW];
  __shared__ unsigned short cand[CAP];
  __shared__ int ncand;
  __shared__ int bar3cnt;  // arrivals at the barriers of the estimate phase (waves 1..3 of the four-wave variant)
#ifdef AGH_FRAME_PAD  // occupancy experiment: pad the LDS so that one work-group less fits a CU
  __shared__ int pad_[AGH_FRAME_PAD / 4];
  if (threadIdx.x == 0 && nmin == -12345)
    pad_[blockIdx.x % (AGH_FRAME_PAD / 4)] = 1;
#endif

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto body = [&](const int s) {
  const int n = nt[s];
  const double* ev = eig + (int64_t) s * 12;
  const bool ok = status[s] == kStatusOk && ev[11] != 0.0;
  // capacity classes: this instantiation owns the valid samples with nmin < n <= CAP; the first one (nmin == 0) also
  // writes the records of invalid samples
  const int ks_class = (rand_mode && n > 50) ? 50 : n;  // normals the sample needs room for
  if (ok ? (ks_class <= nmin || ks_class > CAP) : (nmin != 0))
    return;  // (deterministic mode, n beyond every class here: k_taubin_frame_huge)
  const bool valid = ok;
  const bool bad_index = status[s] == kStatusBadIndex || status[s] == kStatusSkipped;
  const float* qp = xyz + (int64_t) (bad_index ? 0 : samples[s]) * stride;
  if (!valid)
  {
    if (tid == 0)
    {
      agh_frame f;
      for (int k = 0; k < 3; k++)
      {
        f.sample[k] = bad_index ? 0.0 : (double) qp[k];
        f.normal[k] = f.axis[k] = f.binormal[k] = 0.0;
      }
      for (int k = 0; k < 10; k++)
        f.params[k] = 0.0;
      f.eigenvalue = 0.0;
      f.n_nb = n;
      f.majority_cam = 0;
      f.max_index = 0;
      f.valid = 0;
      frames[s] = f;
    }
    return;
  }
  // quadric.cpp:162-170
  const double a = ev[0], b = ev[1], c = ev[2];
  const double d = 2.0 * ev[3], e = 2.0 * ev[4], f = 2.0 * ev[5];
  const double g = ev[6], h = ev[7], i9 = ev[8];
  AGH_FSTAMP(1, 0);
  const bool sub = rand_mode && n > 50;  // quadric.cpp:177-193
  const int ks = sub ? 50 : n;
  if (tid < 2)
    camcnt[tid] = 0;
  if (tid == 0)
  {
    next_col = 0;
    ncand = 0;
    bar3cnt = 0;
  }
  __syncthreads();
  // (a neighbourhood beyond the per-sample list lives in the pool of k_taubin_moments_huge: only the production mode, which
  // needs 50 of its points, gets here with one)
  const float4* nb = (huge_base && huge_base[s] >= 0) ? pool_sorted + huge_base[s] : nbr + (int64_t) s * nbr_stride;
  int cam1 = 0;  // this thread's neighbours seen by camera 1 (quadric.cpp:215-226)
  for (int t = tid; t < ks; t += THREADS)
  {
    const int pick = sub ? (draws[draw_ofs[s] + t] % n) : t;
    const float4 p = nb[pick];
    const double x = (double) p.x, y = (double) p.y, z = (double) p.z;
    const double fx = (((2.0 * a) * x + d * y) + f * z) + g;  // quadric.cpp:238-247
    const double fy = (((2.0 * b) * y + d * x) + e * z) + h;
    const double fz = (((2.0 * c) * z + e * y) + f * x) + i9;
    const double mag = sqrt((fx * fx + fy * fy) + fz * fz);
    nx[t] = fx / mag;
    ny[t] = fy / mag;
    nz[t] = fz / mag;
    cam1 += (int) (__float_as_uint(p.w) & 1u);
  }
  // one LDS atomic per wave (one per neighbour put up to 64 lanes on two addresses, and an LDS atomic serialises them)
  cam1 = wave_allsum_i32(cam1);
  if (lane == 0 && cam1)
    atomicAdd(&camcnt[1], cam1);
  __syncthreads();
  AGH_FSTAMP(2, 0);
  if (debug_stop == 1)
    return;
  // Division of labour in the four-wave variant when the columns are estimated first (more than 64 normals): waves 1..3 run
  // the estimate phase and synchronise among themselves through an LDS counter, while wave 0 accumulates M3 and runs its
  // 3 x 3 eigen solve -- a chain of dependent divisions and square roots on ONE lane, ~10 us -- at the same time; the four
  // waves meet again at the work queue of the exact column sums.  (A round-2 attempt ran the solve in slices between
  // work-group barriers shared by all four waves; its timing depended on the box.  Here wave 0 meets no barrier at all.)
  constexpr bool kSplit = NW == 4;
  const bool use3 = kSplit && ks > 64;  // uniform over the work-group
  constexpr int ET = kSplit ? THREADS - 64 : THREADS;  // threads of the estimate phase
  constexpr int ENW = kSplit ? NW - 1 : NW;
  const int et = kSplit ? tid - 64 : tid, ew = kSplit ? wave - 1 : wave;
  int bar_phase = 0;
  auto wait3 = [&](int target) {
    while (__hip_atomic_load(&bar3cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target)
      __builtin_amdgcn_s_sleep(1);
    __threadfence_block();
  };
  auto ebar = [&]() {  // barrier of the estimate phase: waves 1..3 only in the split variant
    if (kSplit)
    {
      bar_phase++;
      __threadfence_block();
      __builtin_amdgcn_wave_barrier();
      if (lane == 0)
        __hip_atomic_fetch_add(&bar3cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      wait3(3 * bar_phase);
    }
    else
      __syncthreads();
  };
  auto m3_and_axis = [&]() {  // one wave: M3 = normals * normals^T by sequential sums (quadric.cpp:266), then its eigenvectors
    // M3 in the oracle's LaneSum64 order: lane l owns partial l (terms l, l + 64, ...), butterfly tree at the end
    double m[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
    if (!(debug_stop == 5 || debug_stop == 7))
      for (int t = lane; t < ks; t += 64)
      {
        const double x = nx[t], y = ny[t], z = nz[t];
        m[0] += x * x;
        m[1] += x * y;
        m[2] += x * z;
        m[3] += y * y;
        m[4] += y * z;
        m[5] += z * z;
      }
#pragma unroll
    for (int k = 0; k < 6; k++)
    {
      m[k] = wave_allsum_f64(m[k]);
      if (lane == 0)
        sM3[k] = m[k];
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    if (lane == 0)
    {
      const double m3[6] = { sM3[0], sM3[1], sM3[2], sM3[3], sM3[4], sM3[5] };
      double ax[3] = { 1.0, 0.0, 0.0 };
      if (!(debug_stop == 6 || debug_stop == 7))
        smallest_eigvec3(m3, ax);
      sAxis[0] = ax[0];
      sAxis[1] = ax[1];
      sAxis[2] = ax[2];
    }
  };
  // ---- argmax_j sum_i (n_i . n_j)^6 (quadric.cpp:283-284) by filter-and-refine ----
  // The reference needs only the ARGMAX of the n column sums.  (1) A cheap estimate of every column sum:
  //   sum_i (n_i . n_j)^6 = sum_{a+b+c=6} 6!/(a!b!c!) T_abc jx^a jy^b jz^c,   T_abc = sum_i nx_i^a ny_i^b nz_i^c,
  // i.e. 28 moments of the normals (O(n)) instead of n^2 dot products; its rounding error is < 3e-12 n.
  // (2) Every column whose estimate is within delta = 1e-9 n + 1e-7 max of the best estimate -- hundreds of times
  // the error bound, so the true (sequentially rounded) maximum and all its exact ties are among them -- gets the
  // reference's exact sequential sum; the argmax (first index on ties) is taken over those.  The result is the
  // same index the exhaustive n^2 evaluation yields (asserted against the exhaustive oracle by the parity tests).
  // With at most 64 normals (always in the reference's production mode, which subsamples 50) a column's exact sum is
  // one term per lane plus the butterfly: evaluating all columns exactly is cheaper than estimating them first
  // (measured; with up to 128 it is not).
  // Round 3: with at most 64 normals the columns are not queued at all -- ONE wave evaluates them all at once, a column per
  // lane: lane j walks the 64 leaves of the LaneSum64 tree of ITS column (terms i = 0 .. ks - 1, zeros beyond) in
  // bit-reversed order and combines them pairwise on a six-deep register stack, which is the butterfly's order of
  // additions exactly (p[l] + p[l + o], o = 32 .. 1).  The normals are LDS broadcasts.  The queue gave every column a
  // round trip of its own (an LDS atomic, a 6-step ds_bpermute butterfly): 23 us for 50 columns, the kernel's critical
  // path in the production mode; this is ~1.5 us, and runs on the last wave while wave 0 finds the axis.
  // (only in the classes made for few normals -- the production mode's and the all-points pass's; in the large classes a
  // sample with at most 64 neighbours is rare and takes the queue, so that their register budgets stay what they were)
  constexpr bool kSmallCols = CAP <= 128;
  const bool small_cols = kSmallCols && ks <= 64;
  double small_best = -1.0;
  int small_best_j = 0x7fffffff;
  if (!kSmallCols && ks <= 64)
  {
    for (int j = tid; j < ks; j += THREADS)
      cand[j] = (unsigned short) j;
    if (tid == 0)
      ncand = ks;
  }
  else if (small_cols)
  {
    if (tid == 0)
      ncand = 0;
    if (wave == NW - 1)
    {
      const int j = lane < ks ? lane : 0;
      const double jx = nx[j], jy = ny[j], jz = nz[j];
      // visit counter c = 8 b + k: leaf i = bitrev6(c) = (bitrev3(k) << 3) | bitrev3(b).  The eight leaves of a block close
      // three levels of the tree among themselves (unrolled: a three-deep stack in registers); the block sums are combined
      // by the same counter scheme one level up (b is uniform: plain branches).
      double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll 1
      for (int b = 0; b < 8; b++)
      {
        const int br = ((b & 1) << 2) | (b & 2) | ((b & 4) >> 2);
        double stk[4];
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
          const int i = ((((k & 1) << 2) | (k & 2) | ((k & 4) >> 2)) << 3) | br;
          const int ii = i < ks ? i : 0;
          const double gdot = (nx[ii] * jx + ny[ii] * jy) + nz[ii] * jz;
          const double g2 = gdot * gdot;
          double v = i < ks ? (g2 * g2) * g2 : 0.0;
          int depth = __builtin_popcount(k);
#pragma unroll
          for (int m = 0; m < 3; m++)
            if (((k >> m) & 1) && ((k & ((1 << m) - 1)) == ((1 << m) - 1)))
            {
              depth--;
              v = stk[depth] + v;  // the earlier (lower-index) half is the left operand
            }
          stk[depth] = v;
        }
        double v = stk[0];
        if (b & 1)
        {
          v = s0 + v;
          if (b & 2)
          {
            v = s1 + v;
            if (b & 4)
              v = s2 + v;
            else
              s2 = v;
          }
          else
            s1 = v;
        }
        else
          s0 = v;
        if (b == 7)
          s0 = v;
      }
      if (lane < ks)
      {
        small_best = s0;
        small_best_j = lane;
      }
    }
  }
  else if (CAP <= 64)
    ;  // (this instantiation never estimates: the branch below would only cost it registers)
  else if (kSplit && wave == 0)
    m3_and_axis();
  else
  {
    const double mult_et = kMultinomial6[et < 28 ? et : 0];  // (a table read issued here, used two barriers later)
    {
      double T[28];
  #pragma unroll
      for (int k = 0; k < 28; k++)
        T[k] = 0.0;
      for (int t = et; t < ks; t += ET)
      {
        const double x = nx[t], y = ny[t], z = nz[t];
        double px[7], py[7], pz[7];
        px[0] = py[0] = pz[0] = 1.0;
  #pragma unroll
        for (int k = 1; k < 7; k++)
        {
          px[k] = px[k - 1] * x;
          py[k] = py[k - 1] * y;
          pz[k] = pz[k - 1] * z;
        }
        // (an ESTIMATE: it only selects the columns that get the exact sum, with a margin hundreds of times its error --
        // so its multiply-adds may be fused: one product and one fma per moment)
        int k = 0;
  #pragma unroll
        for (int a = 6; a >= 0; a--)
  #pragma unroll
          for (int b = 6 - a; b >= 0; b--)
          {
            T[k] = fma(px[a], py[b] * pz[6 - a - b], T[k]);
            k++;
          }
      }
      // Wave reduction of the 28 moments by a halving butterfly: in the step with partner distance o a lane keeps one
      // half of its values and receives the partner's copies of that half, so 16 + 8 + 4 + 2 + 1 + 1 exchanges do what 28
      // full butterflies (168 exchanges) would; lane l ends with the total of moment l >> 1.
      double R[32];
  #pragma unroll
      for (int k = 0; k < 32; k++)
        R[k] = k < 28 ? T[k] : 0.0;
  #pragma unroll
      for (int half = 16, o = 32; half >= 1; half >>= 1, o >>= 1)
      {
        const bool upper = (lane & o) != 0;
  #pragma unroll
        for (int k = 0; k < half; k++)
        {
          const double send = upper ? R[k] : R[k + half];
          const double keep = upper ? R[k + half] : R[k];
          R[k] = keep + __shfl_xor(send, o);
        }
      }
      R[0] = R[0] + __shfl_xor(R[0], 1);
      if ((lane & 1) == 0 && (lane >> 1) < 28)
        sT[ew][lane >> 1] = R[0];
    }
    AGH_FSTAMP(3, 64);
    ebar();
    if (et < 28)  // multinomial-weighted moments, once per block (the weight was fetched before the moments were summed)
    {
      double tsum = sT[0][et];
      for (int w = 1; w < ENW; w++)
        tsum = tsum + sT[w][et];
      sW[et] = tsum * mult_et;
    }
    ebar();
    constexpr int EST = (CAP + ET - 1) / ET;
    double est[EST];
    double est_max = -1.0;
    {
      double W[28];
  #pragma unroll
      for (int k = 0; k < 28; k++)
        W[k] = sW[k];
  #pragma unroll
      for (int m = 0; m < EST; m++)
      {
        const int j = et + ET * m;
        double e_ = -2.0;
        if (j < ks)
        {
          const double x = nx[j], y = ny[j], z = nz[j];
          // sum_a x^a sum_b W_ab y^b z^(6-a-b) by nested Horner schemes with fused multiply-adds (an estimate, see above):
          // W[base(a) + j] belongs to y^(6-a-j) z^j, so the inner scheme runs down in y with the powers of z as weights
          double pz[7];
          pz[0] = 1.0;
  #pragma unroll
          for (int k = 1; k < 7; k++)
            pz[k] = pz[k - 1] * z;
          e_ = 0.0;
          int base = 0;
  #pragma unroll
          for (int a = 6; a >= 0; a--)
          {
            double r = W[base];
  #pragma unroll
            for (int jz = 1; jz <= 6 - a; jz++)
              r = fma(r, y, W[base + jz] * pz[jz]);
            e_ = fma(e_, x, r);
            base += 7 - a;
          }
          if (!(e_ == e_))
            e_ = 1e300;  // NaN normals: keep every such column as a candidate (exhaustive fallback)
        }
        est[m] = e_;
        est_max = fmax(est_max, e_);
      }
    }
    est_max = wave_max_f64_(est_max);
    if (lane == 0)
      wmax_s[ew] = est_max;
    ebar();
    est_max = wmax_s[0];
    for (int w = 1; w < ENW; w++)
      est_max = fmax(est_max, wmax_s[w]);
    {
      const double delta = 1e-9 * (double) ks + 1e-7 * fabs(est_max);
  #pragma unroll
      for (int m = 0; m < EST; m++)
      {
        const int j = et + ET * m;
        const bool is_c = j < ks && (est[m] >= est_max - delta || est_max >= 1e299);
        const unsigned long long mk = __ballot(is_c);
        int base = 0;
        if (lane == 0 && mk)
          base = atomicAdd(&ncand, __popcll(mk));
        base = __builtin_amdgcn_readfirstlane(base);
        if (is_c)
          cand[base + __popcll(mk & ((1ull << lane) - 1ull))] = (unsigned short) j;
      }
    }
  }
  AGH_FSTAMP(4, 64);  // waves 1..3: candidate list written
  AGH_FSTAMP(5, 0);   // wave 0: M3 and axis done (split variant)
  if (use3)
  {
    if (wave == 0)
      wait3(3 * 4);  // the candidate list is complete when waves 1..3 have passed their fourth barrier
    else
      ebar();
  }
  else
    __syncthreads();
  const int ncnd = ncand;
  if (debug_stop == 2)
    return;
  // ---- without the split: wave 0 (the other waves go straight to the exact column sums below) ----
  if (wave == 0 && !use3)
    m3_and_axis();
  if (debug_stop == 3)
    return;
  // exact sums of the candidate columns in the oracle's LaneSum64 order, one candidate per wave at a time: lane l adds
  // the terms l, l + 64, ... and a butterfly combines the 64 partials (every lane ends with the same total)
  double best = -1.0;
  int best_j = 0x7fffffff;
  // Many candidates -- an exactly planar neighbourhood, where every normal is +-n and no estimate can tell the columns apart:
  // all n of them, n^2 terms -- take them EIGHT at a time: a term's normal is read from LDS once and serves eight columns (one
  // column at a time the phase is bound by LDS bandwidth: 24 bytes per 9 flops; 275 us at the axis-aligned C2), and the eight
  // LaneSum64 butterflies run as ONE transposed butterfly: at the strides 32, 16 and 8 a lane hands on the half of its partial
  // sums its partner keeps (v_permlane32_swap / v_permlane16_swap / row_ror:8), so 4 + 2 + 1 + 3 additions do what 8 x 6 did,
  // every one of them the same pair (p[l], p[l ^ o]) the separate butterflies add: lane l ends with the total of column l >> 3,
  // bit for bit.  (Round 5 took four columns with four separate butterflies: 201 us at the axis-aligned C2.)
  // What bounds the phase: 9 separately rounded fp64 operations per term (the oracle's (x x' + y y') + z z', squared, cubed,
  // added) -- n^2 terms summed over C2u's planar samples are 5.1 G lane-operations, 131 us at the chip's fp64 issue rate.
  if (CAP > 128 && ncnd >= 16)
  {
    for (;;)
    {
      int c = 0;
      if (lane == 0)
        c = atomicAdd(&next_col, 8);
      c = __builtin_amdgcn_readfirstlane(c);
      if (c >= ncnd)
        break;
      double jx[8], jy[8], jz[8], acc[8];
#pragma unroll
      for (int u = 0; u < 8; u++)
      {
        const int jj = cand[min(c + u, ncnd - 1)];  // (a short last group repeats its last column; the repeats are not compared)
        jx[u] = nx[jj];
        jy[u] = ny[jj];
        jz[u] = nz[jj];
        acc[u] = 0.0;
      }
      for (int t = lane; t < ks; t += 64)
      {
        const double tx = nx[t], ty = ny[t], tz = nz[t];
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
          const double gdot = (tx * jx[u] + ty * jy[u]) + tz * jz[u];
          const double g2 = gdot * gdot;
          acc[u] += (g2 * g2) * g2;
        }
      }
      auto sum32 = [](double a, double b) -> double {  // lanes < 32: a[l] + a[l + 32]; lanes >= 32: b[l - 32] + b[l]
        const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
        return __hiloint2double((int) hi[0], (int) lo[0]) + __hiloint2double((int) hi[1], (int) lo[1]);
      };
      auto sum16 = [](double a, double b) -> double {  // even rows: a over (l, l + 16); odd rows: b over (l - 16, l)
        const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
        return __hiloint2double((int) hi[0], (int) lo[0]) + __hiloint2double((int) hi[1], (int) lo[1]);
      };
      const double v0 = sum32(acc[0], acc[4]), v1 = sum32(acc[1], acc[5]), v2 = sum32(acc[2], acc[6]), v3 = sum32(acc[3], acc[7]);
      const double w0 = sum16(v0, v2), w1 = sum16(v1, v3);  // rows: columns {0, 2, 4, 6} and {1, 3, 5, 7}
      const bool up = (lane & 8) != 0;
      double x = (up ? w1 : w0) + xor_partner_f64<8>(up ? w0 : w1);
      x = x + xor_partner_f64<4>(x);
      x = x + xor_partner_f64<2>(x);
      x = x + xor_partner_f64<1>(x);  // lane l: the total of column c + (l >> 3)
      const int u = lane >> 3;
      const int jcol = cand[min(c + u, ncnd - 1)];
      if (c + u < ncnd && (x > best || (x == best && jcol < best_j) || best_j == 0x7fffffff))
      {
        best = x;
        best_j = jcol;
      }
    }
    // the lanes hold the best of their own columns: the wave's best (first index on ties) on every lane, as the loop below expects
    for (int o = 32; o >= 8; o >>= 1)
    {
      const double ob = __shfl_xor(best, o);
      const int oj = __shfl_xor(best_j, o);
      if (oj != 0x7fffffff && (best_j == 0x7fffffff || ob > best || (ob == best && oj < best_j)))
      {
        best = ob;
        best_j = oj;
      }
    }
  }
  for (;;)
  {
    int c = 0;
    if (lane == 0)
      c = atomicAdd(&next_col, 1);
    c = __builtin_amdgcn_readfirstlane(c);
    if (c >= ncnd)
      break;
    const int j = cand[c];
    const double jx = nx[j], jy = ny[j], jz = nz[j];
    double acc = 0.0;
    for (int t = lane; t < ks; t += 64)
    {
      const double gdot = (nx[t] * jx + ny[t] * jy) + nz[t] * jz;
      const double g2 = gdot * gdot;
      acc += (g2 * g2) * g2;
    }
    acc = wave_allsum_f64(acc);
    if (acc > best || (acc == best && j < best_j) || best_j == 0x7fffffff)
    {
      best = acc;
      best_j = j;
    }
  }
  if (small_cols && wave == NW - 1)
  {
    best = small_best;
    best_j = small_best_j;
  }
  AGH_FSTAMP(6, 0);
  if (debug_stop == 4)
    return;
  // argmax with first-index tie-break (Eigen maxCoeff keeps the first maximum).  Only the one-column-per-lane path has a
  // different (best, best_j) in every lane; the column queues leave the same pair in all lanes of a wave (the sums are
  // butterfly totals), and reducing it again was eighteen ds_bpermute round trips on the work-group's tail.
  if (small_cols && wave == NW - 1)
  for (int o = 32; o > 0; o >>= 1)
  {
    const double ob = __shfl_down(best, o);
    const int oj = __shfl_down(best_j, o);
    const bool take = (oj != 0x7fffffff) && (best_j == 0x7fffffff || ob > best || (ob == best && oj < best_j));
    if (take)
    {
      best = ob;
      best_j = oj;
    }
  }
  if (lane == 0)
  {
    wbest[wave] = best;
    wbest_j[wave] = best_j;
  }
  __syncthreads();
  if (tid == 0)
  {
    for (int w = 1; w < NW; w++)
    {
      const double ob = wbest[w];
      const int oj = wbest_j[w];
      const bool take = (oj != 0x7fffffff) && (best_j == 0x7fffffff || ob > best || (ob == best && oj < best_j));
      if (take)
      {
        best = ob;
        best_j = oj;
      }
    }
    const int max_index = best_j;
    double axis[3] = { sAxis[0], sAxis[1], sAxis[2] };
    const double nm[3] = { nx[max_index], ny[max_index], nz[max_index] };
    // normal = normalise((I - a a^T) n_max) (quadric.cpp:285-288)
    double np_[3];
    for (int r = 0; r < 3; r++)
    {
      double pr[3];
      for (int q = 0; q < 3; q++)
        pr[q] = ((r == q) ? 1.0 : 0.0) - axis[r] * axis[q];
      np_[r] = (pr[0] * nm[0] + pr[1] * nm[1]) + pr[2] * nm[2];
    }
    const double nn = sqrt((np_[0] * np_[0] + np_[1] * np_[1]) + np_[2] * np_[2]);
    double normal[3] = { np_[0] / nn, np_[1] / nn, np_[2] / nn };
    double binormal[3] = { axis[1] * normal[2] - axis[2] * normal[1], axis[2] * normal[0] - axis[0] * normal[2],
      axis[0] * normal[1] - axis[1] * normal[0] };  // quadric.cpp:291
    const int maj = (camcnt[1] > ks - camcnt[1]) ? 1 : 0;  // camera 1's count against camera 0's
    const double sample[3] = { (double) qp[0], (double) qp[1], (double) qp[2] };
    const double s2s[3] = { sample[0] - (maj ? cam1x : cam0x), sample[1] - (maj ? cam1y : cam0y),
      sample[2] - (maj ? cam1z : cam0z) };
    if ((normal[0] * s2s[0] + normal[1] * s2s[1]) + normal[2] * s2s[2] > 0)
      for (int r = 0; r < 3; r++)
        normal[r] *= -1.0;
    if ((binormal[0] * s2s[0] + binormal[1] * s2s[1]) + binormal[2] * s2s[2] > 0)
      for (int r = 0; r < 3; r++)
        binormal[r] *= -1.0;
    axis[0] = normal[1] * binormal[2] - normal[2] * binormal[1];  // quadric.cpp:304
    axis[1] = normal[2] * binormal[0] - normal[0] * binormal[2];
    axis[2] = normal[0] * binormal[1] - normal[1] * binormal[0];
    agh_frame fr;
    for (int r = 0; r < 3; r++)
    {
      fr.sample[r] = sample[r];
      fr.normal[r] = normal[r];
      fr.axis[r] = axis[r];
      fr.binormal[r] = binormal[r];
    }
    for (int k = 0; k < 10; k++)
      fr.params[k] = ev[k];
    fr.eigenvalue = ev[10];
    fr.n_nb = n;
    fr.majority_cam = maj;
    fr.max_index = max_index;
    fr.valid = 1;
    frames[s] = fr;
    if (normals_out)
    {
      double* o = normals_out + 3 * (int64_t) samples[s];  // cloud_normals_.col(indices[i]) (hand_search.cpp:102)
      o[0] = normal[0];
      o[1] = normal[1];
      o[2] = normal[2];
    }
    AGH_FSTAMP(7, 0);
  }
  };  // body
  if constexpr (CAP == 4096)  // (only this instantiation carries the loop: in the others it cost registers they do not have)
  {
    const int n_items = ovf_list ? min(ovf_list[0], S) : S;
    for (int it = blockIdx.x; it < n_items; it += gridDim.x)
    {
      body(ovf_list ? ovf_list[1 + it] : (order ? order[it] : it));
      if (it + (int) gridDim.x < n_items)
        __syncthreads();  // (the next sample reuses the tiles)
    }
  }
  else
    body(order ? order[blockIdx.x] : (int) blockIdx.x);
#undef AGH_FSTAMP
}

// K1c beyond the LDS-resident classes (deterministic normals, a neighbourhood in the pool of k_taubin_moments_huge: more than
// nbr_stride points): the same steps on normals kept in GLOBAL memory -- quadric-gradient normals (quadric.cpp:238-247), the
// majority camera, M3 in LaneSum64 order and its smallest eigenvector, and the column sums of quadric.cpp:283-284 for EVERY column
// (no estimate phase: n^2 terms, ~15 ms for 12 000 points -- a rare path), each in LaneSum64 order, first maximum kept; then the
// frame exactly as k_taubin_frame forms it.
__global__ __launch_bounds__(256) void k_taubin_frame_huge(const float4* __restrict__ pool_sorted, double* __restrict__ pool_normals,
  const long long* __restrict__ huge_base, const int32_t* __restrict__ nt, const double* __restrict__ eig,
  const int32_t* __restrict__ status, const float* __restrict__ xyz, int64_t stride, const int32_t* __restrict__ samples, int S,
  double cam0x, double cam0y, double cam0z, double cam1x, double cam1y, double cam1z, agh_frame* __restrict__ frames,
  double* __restrict__ normals_out)
{
  __shared__ int camcnt1, next_col;
  __shared__ double sM3[6], sAxis[3];
  __shared__ double wbest[4];
  __shared__ int wbest_j[4];
  const int s = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long base = huge_base[s];
  if (base < 0)
    return;
  const int n = nt[s];
  const double* ev = eig + (int64_t) s * 12;
  if (!(status[s] == kStatusOk && ev[11] != 0.0))
    return;  // (k_taubin_frame's first class writes the records of samples without a frame)
  const float4* nb = pool_sorted + base;
  double* const nx = pool_normals + base;
  double* const ny = pool_normals + kHugeEntries + base;
  double* const nz = pool_normals + 2 * kHugeEntries + base;
  const double a = ev[0], b = ev[1], c = ev[2];
  const double d = 2.0 * ev[3], e = 2.0 * ev[4], f = 2.0 * ev[5];
  const double g = ev[6], h = ev[7], i9 = ev[8];
  if (tid == 0)
  {
    camcnt1 = 0;
    next_col = 0;
  }
  __syncthreads();
  int cam1 = 0;
  for (int t = tid; t < n; t += 256)
  {
    const float4 p = nb[t];
    const double x = (double) p.x, y = (double) p.y, z = (double) p.z;
    const double fx = (((2.0 * a) * x + d * y) + f * z) + g;  // quadric.cpp:238-247
    const double fy = (((2.0 * b) * y + d * x) + e * z) + h;
    const double fz = (((2.0 * c) * z + e * y) + f * x) + i9;
    const double mag = sqrt((fx * fx + fy * fy) + fz * fz);
    nx[t] = fx / mag;
    ny[t] = fy / mag;
    nz[t] = fz / mag;
    cam1 += (int) (__float_as_uint(p.w) & 1u);
  }
  cam1 = wave_allsum_i32(cam1);
  if (lane == 0 && cam1)
    atomicAdd(&camcnt1, cam1);
  __threadfence();
  __syncthreads();
  if (wave == 0)  // M3 = normals * normals^T in the oracle's LaneSum64 order (quadric.cpp:266), then its smallest eigenvector
  {
    double m[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
    for (int t = lane; t < n; t += 64)
    {
      const double x = nx[t], y = ny[t], z = nz[t];
      m[0] += x * x;
      m[1] += x * y;
      m[2] += x * z;
      m[3] += y * y;
      m[4] += y * z;
      m[5] += z * z;
    }
#pragma unroll
    for (int k = 0; k < 6; k++)
    {
      m[k] = wave_allsum_f64(m[k]);
      if (lane == 0)
        sM3[k] = m[k];
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    if (lane == 0)
    {
      const double m3[6] = { sM3[0], sM3[1], sM3[2], sM3[3], sM3[4], sM3[5] };
      double ax[3] = { 1.0, 0.0, 0.0 };
      smallest_eigvec3(m3, ax);
      sAxis[0] = ax[0];
      sAxis[1] = ax[1];
      sAxis[2] = ax[2];
    }
  }
  // the exact sum of every column, four columns at a time per wave (a term's normal is read once and serves four columns)
  double best = -1.0;
  int best_j = 0x7fffffff;
  for (;;)
  {
    int c0 = 0;
    if (lane == 0)
      c0 = atomicAdd(&next_col, 4);
    c0 = __builtin_amdgcn_readfirstlane(c0);
    if (c0 >= n)
      break;
    int jj[4];
    double jx[4], jy[4], jz[4], acc[4];
#pragma unroll
    for (int u = 0; u < 4; u++)
    {
      jj[u] = min(c0 + u, n - 1);  // (a short last group repeats its last column; the repeats are not compared)
      jx[u] = nx[jj[u]];
      jy[u] = ny[jj[u]];
      jz[u] = nz[jj[u]];
      acc[u] = 0.0;
    }
    for (int t = lane; t < n; t += 64)
    {
      const double tx = nx[t], ty = ny[t], tz = nz[t];
#pragma unroll
      for (int u = 0; u < 4; u++)
      {
        const double gdot = (tx * jx[u] + ty * jy[u]) + tz * jz[u];
        const double g2 = gdot * gdot;
        acc[u] += (g2 * g2) * g2;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
    {
      const double sum = wave_allsum_f64(acc[u]);
      if (c0 + u < n && (sum > best || (sum == best && jj[u] < best_j) || best_j == 0x7fffffff))
      {
        best = sum;
        best_j = jj[u];
      }
    }
  }
  if (lane == 0)
  {
    wbest[wave] = best;
    wbest_j[wave] = best_j;
  }
  __syncthreads();
  if (tid == 0)
  {
    for (int w = 1; w < 4; w++)  // argmax with first-index tie-break (Eigen maxCoeff keeps the first maximum)
    {
      const double ob = wbest[w];
      const int oj = wbest_j[w];
      const bool take = (oj != 0x7fffffff) && (best_j == 0x7fffffff || ob > best || (ob == best && oj < best_j));
      if (take)
      {
        best = ob;
        best_j = oj;
      }
    }
    const int max_index = best_j;
    const float* qp = xyz + (int64_t) samples[s] * stride;
    double axis[3] = { sAxis[0], sAxis[1], sAxis[2] };
    const double nm[3] = { nx[max_index], ny[max_index], nz[max_index] };
    // normal = normalise((I - a a^T) n_max) (quadric.cpp:285-288)
    double np_[3];
    for (int r = 0; r < 3; r++)
    {
      double pr[3];
      for (int q = 0; q < 3; q++)
        pr[q] = ((r == q) ? 1.0 : 0.0) - axis[r] * axis[q];
      np_[r] = (pr[0] * nm[0] + pr[1] * nm[1]) + pr[2] * nm[2];
    }
    const double nn = sqrt((np_[0] * np_[0] + np_[1] * np_[1]) + np_[2] * np_[2]);
    double normal[3] = { np_[0] / nn, np_[1] / nn, np_[2] / nn };
    double binormal[3] = { axis[1] * normal[2] - axis[2] * normal[1], axis[2] * normal[0] - axis[0] * normal[2],
      axis[0] * normal[1] - axis[1] * normal[0] };  // quadric.cpp:291
    const int maj = (camcnt1 > n - camcnt1) ? 1 : 0;  // camera 1's count against camera 0's
    const double sample[3] = { (double) qp[0], (double) qp[1], (double) qp[2] };
    const double s2s[3] = { sample[0] - (maj ? cam1x : cam0x), sample[1] - (maj ? cam1y : cam0y),
      sample[2] - (maj ? cam1z : cam0z) };
    if ((normal[0] * s2s[0] + normal[1] * s2s[1]) + normal[2] * s2s[2] > 0)
      for (int r = 0; r < 3; r++)
        normal[r] *= -1.0;
    if ((binormal[0] * s2s[0] + binormal[1] * s2s[1]) + binormal[2] * s2s[2] > 0)
      for (int r = 0; r < 3; r++)
        binormal[r] *= -1.0;
    axis[0] = normal[1] * binormal[2] - normal[2] * binormal[1];  // quadric.cpp:304
    axis[1] = normal[2] * binormal[0] - normal[0] * binormal[2];
    axis[2] = normal[0] * binormal[1] - normal[1] * binormal[0];
    agh_frame fr;
    for (int r = 0; r < 3; r++)
    {
      fr.sample[r] = sample[r];
      fr.normal[r] = normal[r];
      fr.axis[r] = axis[r];
      fr.binormal[r] = binormal[r];
    }
    for (int k = 0; k < 10; k++)
      fr.params[k] = ev[k];
    fr.eigenvalue = ev[10];
    fr.n_nb = n;
    fr.majority_cam = maj;
    fr.max_index = max_index;
    fr.valid = 1;
    frames[s] = fr;
    if (normals_out)
    {
      double* o = normals_out + 3 * (int64_t) samples[s];  // cloud_normals_.col(indices[i]) (hand_search.cpp:102)
      o[0] = normal[0];
      o[1] = normal[1];
      o[2] = normal[2];
    }
  }
}

__global__ void k_draw_offsets(const int32_t* __restrict__ nt, int S, int32_t* __restrict__ draw_ofs,
  int32_t* __restrict__ total_io)
{
  draw_offsets_wave(nt, S, draw_ofs, total_io);
}

int taubin_frames(Ctx* c, const int32_t* d_samples, int64_t S, double radius, agh_frame* d_frames, int32_t* d_nt,
  bool write_normals, hipStream_t st)
{
  const int rc = taubin_moments_eigen(c, d_samples, S, radius, d_nt, st, true);
  if (rc != AGH_OK || c->debug_stop_moments)
    return rc;
  return taubin_frame_stage(c, d_samples, S, radius, d_frames, d_nt, write_normals, st, true);
}

// K1a + K1b for S samples (the sharded search exchanges the RAND50 draw counts between this stage and the next)
int taubin_moments_eigen(Ctx* c, const int32_t* d_samples, int64_t S, double radius, int32_t* d_nt, hipStream_t st,
  bool with_draw_offsets)
{
  if (S == 0)
    return AGH_OK;
  GridView gv{ c->d_desc, c->d_cell_start, c->d_sorted, c->d_cloud_off, c->n_clouds };
  const float r2f = static_cast<float>(radius * radius);  // pcl::KdTreeFLANN::radiusSearch squares in double, casts
  const double rpad = radius * 1.0001 + 1e-6;
  const int Si = (int) S;
  // capacity classes: smallest first; later classes only touch samples flagged kStatusOverflow
  const bool small_first = radius <= 0.015;
  long long* moments_dbg = nullptr;
#ifdef AGH_DEBUG_HOOKS
  if (const char* k = getenv("AGH_DEBUG_CLOCKS_KERNEL"))
    if (!strcmp(k, "moments") && !small_first)
      moments_dbg = c->d_dbg;
#endif
  bool first = true;
  int32_t* zf = c->zero_flags_pending ? c->d_flags : nullptr;
  c->zero_flags_pending = false;
  // (the samples the 1152 class hands on are listed for the 4096 class: see k_taubin_moments)
  int32_t* const ovf = (c->big_classes && c->d_ovf) ? c->d_ovf : nullptr;
  const int big_grid = std::min(Si, kBigListGrid);
  if (ovf && hipMemsetAsync(ovf, 0, sizeof(int32_t), st) != hipSuccess)
    return AGH_ERR_HIP;
  if (small_first)
  {
    hipLaunchKernelGGL(k_taubin_moments<256>, dim3(Si), dim3(256), 0, st, gv, c->d_xyz, c->stride_floats, d_samples, Si,
      r2f, rpad, 1, c->d_sums, d_nt, c->d_status, c->d_nbr, c->nbr_stride, c->debug_stop_moments, c->n_is_bound ? -1 : (int) c->n,
      zf, c->d_scloud, moments_dbg, (int32_t*) nullptr, (const int32_t*) nullptr);
    zf = nullptr;
    first = false;
  }
  // The classes after the first only do work for samples the first one flagged, which the host cannot know: their
  // launches (~5 us each, empty for voxelised clouds) are skipped until a call reports such a sample (AGH_ERR_RETRY, then
  // c->big_classes stays set).  Skipped classes leave kStatusOverflow behind, which k_taubin_eigen turns into the flag.
  if (first || c->big_classes)
    hipLaunchKernelGGL(k_taubin_moments<1152>, dim3(Si), dim3(256), 0, st, gv, c->d_xyz, c->stride_floats, d_samples, Si,
      r2f, rpad, first ? 1 : 0, c->d_sums, d_nt, c->d_status, c->d_nbr, c->nbr_stride, c->debug_stop_moments, c->n_is_bound ? -1 : (int) c->n,
      zf, c->d_scloud, moments_dbg, ovf, (const int32_t*) nullptr);
  if (c->big_classes)
    hipLaunchKernelGGL(k_taubin_moments<4096>, dim3(ovf ? big_grid : Si), dim3(256), 0, st, gv, c->d_xyz, c->stride_floats, d_samples, Si,
      r2f, rpad, 0, c->d_sums, d_nt, c->d_status, c->d_nbr, c->nbr_stride, c->debug_stop_moments, c->n_is_bound ? -1 : (int) c->n,
      (int32_t*) nullptr, c->d_scloud, (long long*) nullptr, (int32_t*) nullptr, (const int32_t*) ovf);
  if (c->huge_classes && c->d_huge_stage)
  {
    // (one slot of the pool per flagged sample, handed out by the kernel; the counter starts every launch at zero)
    if (hipMemsetAsync(c->d_huge_count, 0, sizeof(unsigned long long), st) != hipSuccess)
      return AGH_ERR_HIP;
    hipLaunchKernelGGL(k_taubin_moments_huge, dim3(Si), dim3(256), 0, st, gv, c->d_xyz, c->stride_floats, d_samples, Si, r2f, rpad,
      c->d_sums, d_nt, c->d_status, c->d_nbr, c->nbr_stride, c->d_huge_stage, c->d_huge_key, c->d_huge_sorted, c->d_huge_base,
      c->d_huge_count);
  }
  timing_mark(c, "taubin_moments", st);
  if (c->debug_stop_moments)
    return AGH_OK;  // phase-timing aid: the truncated kernel left no usable sums behind
  // One sample per lane, or -- for up to 4096 samples: 512 waves, every second SIMD -- eight, which share the bisection
  // (taubin_eigen.h): C2 22.9 -> 18.5 us (HIP events); with C4's 8000 samples (1000 waves) it measured 24.7 against 22.6, so
  // C4, the 16 000 samples of a batch and the 300 000 of the all-points pass stay at one lane each.
#ifndef AGH_LPS8_MAX
#define AGH_LPS8_MAX 4096
#endif
  const int lps = Si <= AGH_LPS8_MAX ? 8 : 1;
  const bool with_orders = Si <= kOrderMaxSamples;  // (see k_taubin_eigen: beyond, the sorter work-group would be the kernel)
  const int eig_groups = (Si * lps + 255) / 256 + (with_orders ? 2 : 0);  // + the two sorter work-groups
  // (production mode: one more work-group computes the RAND50 draw offsets, unless the caller exchanges the counts
  // between the ranks first -- the sharded search)
  int32_t* dofs = (with_draw_offsets && c->p.normals_mode == AGH_NORMALS_RAND50) ? c->d_draw_ofs : nullptr;
  const int eig_grid = eig_groups + (dofs ? 1 : 0);
  // the hand sweep's block-wise order rides along (hand_sweep() uses it when it was made for exactly its launch)
  const bool with_sweep_order = with_orders && radius > 0.015 && (Si + kSweepBlock - 1) / kSweepBlock <= kSweepBlocksMax && Si >= 4 * kSweepBlock;
  c->order_frame_s = with_orders ? Si : 0;
  c->order_frame_samples = with_orders ? d_samples : nullptr;
  int* sweep_order = with_sweep_order ? c->d_order_sweep : nullptr;
  c->order_sweep_s = with_sweep_order ? Si : 0;
  c->order_sweep_samples = with_sweep_order ? d_samples : nullptr;
  if (lps == 8)
    hipLaunchKernelGGL(k_taubin_eigen<8>, dim3(eig_grid), dim3(256), 0, st, c->d_sums, d_nt, c->d_status, Si, c->d_eig,
      c->d_flags, (const int*) d_nt, with_orders ? c->d_order : (int*) nullptr, dofs, c->d_flags + 2, sweep_order);
  else
    hipLaunchKernelGGL(k_taubin_eigen<1>, dim3(eig_grid), dim3(256), 0, st, c->d_sums, d_nt, c->d_status, Si, c->d_eig,
      c->d_flags, (const int*) d_nt, with_orders ? c->d_order : (int*) nullptr, dofs, c->d_flags + 2, sweep_order);
  timing_mark(c, "taubin_eigen", st);
  return hipGetLastError() == hipSuccess ? AGH_OK : AGH_ERR_HIP;
}

// K1c (and the RAND50 draw offsets, continuing from the count in d_flags[2]) for the same S samples
int taubin_frame_stage(Ctx* c, const int32_t* d_samples, int64_t S, double radius, agh_frame* d_frames, int32_t* d_nt,
  bool write_normals, hipStream_t st, bool draw_offsets_done)
{
  if (S == 0)
    return AGH_OK;
  const int Si = (int) S;
  const int rand_mode = c->p.normals_mode == AGH_NORMALS_RAND50 ? 1 : 0;
  if (rand_mode && !draw_offsets_done)
    hipLaunchKernelGGL(k_draw_offsets, dim3(1), dim3(64), 0, st, d_nt, Si, c->d_draw_ofs, c->d_flags + 2);
  const double* co = &c->p.cam_origin[0][0];
  // capacity classes (LDS = 24 B per normal) by the number of normals a sample needs: at most 128 (typical of the
  // r = 0.01 all-points pass) -> one wave per sample; voxelised clouds at r = 0.03 fit the 1280 class (4 blocks per CU); the 4096 class only does work for the
  // samples that need it
  // (measured: in the production mode the four-wave kernel is faster -- its 50 exact column sums are shared by four
  // waves -- so the one-wave class serves the all-points pass only: 5.3 -> 4.0 ms at 300k points)
  const bool small_class = radius <= 0.015;
  long long* frame_dbg = nullptr;
#ifdef AGH_DEBUG_HOOKS
  if (const char* k = getenv("AGH_DEBUG_CLOCKS_KERNEL"))
    if (!strcmp(k, "frame") && !small_class)
      frame_dbg = c->d_dbg;
#endif
  // the longest-first order k_taubin_eigen's sorter made for exactly this launch (same sample list, same count), else sample order
  const int* frame_order = (c->order_frame_s == Si && c->order_frame_samples == d_samples) ? (const int*) c->d_order : nullptr;
#define AGH_LAUNCH_FRAME(CAP, THREADS, NMIN) AGH_LAUNCH_FRAME_L(CAP, THREADS, NMIN, Si, (const int32_t*) nullptr)
#define AGH_LAUNCH_FRAME_L(CAP, THREADS, NMIN, GRID, LIST)                                                              \
  hipLaunchKernelGGL((k_taubin_frame<CAP, THREADS>), dim3(GRID), dim3(THREADS), 0, st, c->d_nbr, c->nbr_stride, d_nt,    \
    c->d_eig, c->d_status, c->d_xyz, c->stride_floats, d_samples, Si, rand_mode, c->d_draw_ofs, c->d_draws, co[0], co[1], \
    co[2], co[3], co[4], co[5], d_frames, write_normals ? c->d_normals : nullptr, NMIN, c->debug_stop_frame,           \
    frame_order, frame_dbg, (const float4*) c->d_huge_sorted,                                                          \
    (const long long*) (c->huge_classes ? c->d_huge_base : nullptr), LIST)
  if (small_class)
    AGH_LAUNCH_FRAME(128, 64, 0);
  if (rand_mode && !small_class)
  {
    // the reference's production mode: a sample needs room for min(n, 50) normals whatever its neighbourhood holds, so one
    // class serves every sample -- 1.5 KB of LDS and no estimate phase (<= 64 VGPRs): eight work-groups per CU, all 2000
    // work-groups of C2 resident in one round (C2 49 -> 47 us, the batch of eight 333 -> 277; the 1152 class ran them five
    // per CU, each mostly waiting for its one lane in the 3 x 3 eigen solve)
    AGH_LAUNCH_FRAME(64, 256, 0);
  }
  else
  {
    if (small_class)
      AGH_LAUNCH_FRAME(1152, 256, 128);
    else
      AGH_LAUNCH_FRAME(1152, 256, 0);
    if (c->big_classes && c->d_ovf)  // (n_t > 1152 needs K1a's 4096 class, which only runs with this set: the samples it listed)
      AGH_LAUNCH_FRAME_L(4096, 256, 1152, std::min(Si, kBigListGrid), (const int32_t*) c->d_ovf);
    else if (c->big_classes)
      AGH_LAUNCH_FRAME(4096, 256, 1152);
    if (c->huge_classes)  // (its 24 bytes of LDS per normal still fit: 147 KB)
    {
      AGH_LAUNCH_FRAME(kHugeCap, 256, 4096);
      // ... and what is beyond that, from the pool (all but the pooled samples return at once)
      hipLaunchKernelGGL(k_taubin_frame_huge, dim3(Si), dim3(256), 0, st, (const float4*) c->d_huge_sorted, c->d_huge_normals,
        (const long long*) c->d_huge_base, (const int32_t*) d_nt, (const double*) c->d_eig, (const int32_t*) c->d_status, c->d_xyz,
        c->stride_floats, d_samples, Si, co[0], co[1], co[2], co[3], co[4], co[5], d_frames, write_normals ? c->d_normals : nullptr);
    }
  }
#undef AGH_LAUNCH_FRAME
#undef AGH_LAUNCH_FRAME_L
  timing_mark(c, "taubin_frame", st);
  return hipGetLastError() == hipSuccess ? AGH_OK : AGH_ERR_HIP;
}

}  // namespace agh
