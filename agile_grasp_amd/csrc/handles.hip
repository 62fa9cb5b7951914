// handles.hip -- K5: handle search on the hypotheses.
//
// Stands in for HandleSearch::findHandles (reference src/agile_grasp/handle_search.cpp:4-128) and the Handle
// constructor (src/agile_grasp/handle.cpp:3-74):
//   K5a k_handle_pairs   the H x H geometric test of :21-45 as a bit matrix (one row of 64-bit words per hand); the two
//                        acos thresholds become comparisons of the dot products with host-computed constants x1, x2
//                        (acos is monotone: acos(x) < 0.34 <=> x >= x1, pi - acos(x) < 0.34 <=> x <= x2; the constants
//                        are found by bisection with the same libm the oracle uses)
//   K5b the walk (:11-19, 47-80): for every still-available seed hand in index order collect its available inliers, sort by
//                        distance along the seed's axis, cut at the first 2 cm gap (shortenHandle, :88-118, with the meaning
//                        the oracle states for its out-of-range read), accept if long enough, retire the members.
//       k_handle_batch   (round 4) sixteen open seeds per round, one wave each, everything in registers, committed as if in
//                        index order -- for pair matrices whose rows hold at most 64 hands (what the search meets on the
//                        hands Learning::classify keeps); 15 rounds of ~4.7 us for the pipeline's 499 hands
//       k_handle_greedy  the sequential walk by one wave, for longer rows (launched along only when the previous set of
//                        hands needed it)
//   K5c k_handle_build   Handle::Handle per accepted handle, one wave each.
#include "agh_internal.h"

#include <type_traits>

namespace agh
{

constexpr int kHandleListCap = 2048;  // inliers of one seed (LDS)
constexpr int kHandleMaxHands = 8192;

__device__ __forceinline__ double dot3d(const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

// d_H (all kernels of this file): the number of hands in device memory, when the host does not know it (agh_localize: the
// hands are what the SVM kept of a search still in flight); the launch is then sized for a bound and H, W are read here.
__global__ __launch_bounds__(256) void k_handle_pairs(const agh_hypothesis* __restrict__ hands, int H, double x1, double x2,
  unsigned long long* __restrict__ bits, int W, int* __restrict__ rowcnt, const int* __restrict__ d_H)
{
  if (d_H)
  {
    H = min(*d_H, kHandleMaxHands);
    W = (H + 63) >> 6;
  }
  const int i = blockIdx.x;
  if (i >= H)
    return;
  const int tid = threadIdx.x, lane = tid & 63;
  const agh_hypothesis& hi = hands[i];
  const double ia[3] = { hi.axis[0], hi.axis[1], hi.axis[2] };
  const double ip[3] = { hi.bottom[0], hi.bottom[1], hi.bottom[2] };
  const double in_[3] = { hi.approach[0], hi.approach[1], hi.approach[2] };
  int cnt = 0;
  for (int j0 = 0; j0 < W * 64; j0 += 256)
  {
    const int j = j0 + tid;
    bool inl = false;
    if (j < H)
    {
      const agh_hypothesis& hj = hands[j];
      const double d[3] = { hj.bottom[0] - ip[0], hj.bottom[1] - ip[1], hj.bottom[2] - ip[2] };
      double v[3];
      for (int r = 0; r < 3; r++)  // (I - a a^T) d, row by row, left to right (handle_search.cpp:33)
      {
        const double p0 = ((r == 0) ? 1.0 : 0.0) - ia[r] * ia[0];
        const double p1 = ((r == 1) ? 1.0 : 0.0) - ia[r] * ia[1];
        const double p2 = ((r == 2) ? 1.0 : 0.0) - ia[r] * ia[2];
        v[r] = (p0 * d[0] + p1 * d[1]) + p2 * d[2];
      }
      const double dist_from_line = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
      const double aa = dot3d(ia, hj.axis), nn = dot3d(in_, hj.approach);
      inl = dist_from_line < 0.01 && (aa >= x1 || aa <= x2) && nn >= x1;  // :39
    }
    const unsigned long long m = __ballot(inl);
    if (lane == 0 && (j0 + tid) / 64 < W)
      bits[(int64_t) i * W + (j0 + tid) / 64] = m;
    cnt += (lane == 0) ? __popcll(m) : 0;
  }
  __shared__ int sc[4];
  if (lane == 0)
    sc[tid >> 6] = cnt;
  __syncthreads();
  if (tid == 0)
    rowcnt[i] = sc[0] + sc[1] + sc[2] + sc[3];
}

struct HandleCounts
{
  int n_handles, n_idx, error;
  int sequential;  // set by k_handle_batch when it declines (a seed with more than 64 potential inliers): k_handle_greedy runs
};

constexpr int kHandleLdsHands = 640;  // hands whose axis, bottom and pair-matrix rows fit the LDS of the small variant
constexpr int kHandleLdsWords = kHandleLdsHands / 64;

// SMALL: H <= kHandleLdsHands.  The loop is a chain of dependent look-ups per seed (seed row -> inliers -> their
// positions): the small variant keeps every hand's axis and grasp bottom AND the pair matrix in LDS (106 KiB of the
// CU's 160), so a seed costs LDS latency only; the general variant prefetches the next seed's row from global memory.  The general variant reads both from global memory.
// (wave-local ordering of LDS traffic: the loop below is run by ONE wave, whose DS instructions execute in order; this only
// keeps the compiler from moving them across the point)
#define AGH_WAVE_SYNC()                                      \
  do                                                         \
  {                                                          \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
  } while (0)

template <bool SMALL>
__global__ __launch_bounds__(256) void k_handle_greedy(const agh_hypothesis* __restrict__ hands, int H,
  const unsigned long long* __restrict__ bits, int W, const int* __restrict__ rowcnt, int min_inliers, double min_length,
  int* __restrict__ h_first, int* __restrict__ h_n, int* __restrict__ inlier_idx, HandleCounts* __restrict__ counts,
  int* __restrict__ host_idx, int host_idx_cap, int* __restrict__ host_counts, const int* __restrict__ d_H)
{
  constexpr int kCap = SMALL ? 1024 : kHandleListCap;
  if (d_H)
  {
    H = *d_H;
    W = (H + 63) >> 6;
    if (H > kHandleMaxHands || (H <= kHandleLdsHands) != SMALL)  // (both variants are launched; the other one's case)
      return;
  }
  if (!counts->sequential)  // (k_handle_batch has done the search)
    return;
  __shared__ unsigned long long alive[128];  // W <= 128 (H <= 8192)
  __shared__ unsigned long long elig[128];   // rows with at least min_inliers inliers at all (no global load per seed)
  __shared__ double ld[kCap], sd[kCap];
  __shared__ int lj[kCap], sj[kCap];
  __shared__ double hpos[SMALL ? kHandleLdsHands : 1][6];  // axis, bottom
  __shared__ unsigned long long lbits[SMALL ? kHandleLdsHands * kHandleLdsWords : 1];  // the pair matrix (51 KiB)
  __shared__ int s_gap, s_nh, s_nidx;
  const int tid = threadIdx.x;
  // set-up by the whole work-group (four waves: the tables are ~100 KB of dependent global reads); the walk itself is one wave's
  for (int t = tid; t < 128; t += 256)
  {
    alive[t] = 0ull;
    elig[t] = 0ull;
  }
  if (tid == 0)
  {
    s_nh = 0;
    s_nidx = 0;
  }
  __syncthreads();
  for (int j = tid; j < H; j += 256)
  {
    if (hands[j].width != -1.0)  // handle_search.cpp:13,25: width -1 marks a retired hand
      atomicOr(&alive[j >> 6], 1ull << (j & 63));
    if (rowcnt[j] >= min_inliers)
      atomicOr(&elig[j >> 6], 1ull << (j & 63));
    if (SMALL)
      for (int r = 0; r < 3; r++)
      {
        hpos[j][r] = hands[j].axis[r];
        hpos[j][3 + r] = hands[j].bottom[r];
      }
  }
  if (SMALL)
    for (int k = tid; k < H * W; k += 256)
      lbits[k] = bits[k];
  __syncthreads();
  if (tid >= 64)
    return;
  auto next_seed = [&](int from) -> int {  // first available, eligible seed >= from (H if none)
    for (int w = from >> 6; w < W; w++)
    {
      unsigned long long m = alive[w] & elig[w];
      if (w == (from >> 6))
        m &= ~0ull << (from & 63);
      if (m)
        return w * 64 + __ffsll((long long) m) - 1;
    }
    return H;
  };
  int i = next_seed(0);
  // the seed's row of the pair matrix: lane t holds words t and 64 + t
  auto load_row = [&](int seed, unsigned long long& r0, unsigned long long& r1) {
    if (SMALL)
    {
      r0 = (tid < W && seed < H) ? lbits[seed * W + tid] : 0ull;
      r1 = 0ull;  // (W <= 10)
    }
    else
    {
      r0 = (tid < W && seed < H) ? bits[(int64_t) seed * W + tid] : 0ull;
      r1 = (64 + tid < W && seed < H) ? bits[(int64_t) seed * W + 64 + tid] : 0ull;
    }
  };
  unsigned long long row0, row1;
  load_row(i, row0, row1);
  while (i < H)
  {
    // the row of the seed that will most likely come next travels while this seed is processed
    const int guess = next_seed(i + 1);
    unsigned long long nrow0, nrow1;
    load_row(guess, nrow0, nrow1);
    // available inliers of seed i: mask, count, compact in ascending j
    // (one wave runs the whole loop: the count and the exclusive offsets of the words are a wave scan, no barrier and
    // no serial pass over the words)
    unsigned long long m0 = tid < W ? (row0 & alive[tid]) : 0ull;
    unsigned long long m1 = 64 + tid < W ? (row1 & alive[64 + tid]) : 0ull;
    const int c0 = __popcll(m0), c1 = __popcll(m1);
    int i0 = c0, i1 = c1, tot0;
    if (W <= 16)
    {
      // all counts sit in the first DPP row; an inclusive scan by four row shifts (no LDS crossbar: a
      // __shfl_up scan is twelve dependent ds_bpermute round trips, most of a rejected seed's time)
      i0 += __builtin_amdgcn_update_dpp(0, i0, 0x111, 0xf, 0xf, true);  // row_shr:1
      i0 += __builtin_amdgcn_update_dpp(0, i0, 0x112, 0xf, 0xf, true);  // row_shr:2
      i0 += __builtin_amdgcn_update_dpp(0, i0, 0x114, 0xf, 0xf, true);  // row_shr:4
      i0 += __builtin_amdgcn_update_dpp(0, i0, 0x118, 0xf, 0xf, true);  // row_shr:8
      tot0 = __builtin_amdgcn_readlane(i0, 15);
      i1 = 0;
    }
    else
    {
      for (int o = 1; o < 64; o <<= 1)
      {
        const int y0 = __shfl_up(i0, o), y1 = __shfl_up(i1, o);
        if (tid >= o)
        {
          i0 += y0;
          i1 += y1;
        }
      }
      tot0 = __shfl(i0, 63);
    }
    const int ofs0 = i0 - c0, ofs1 = tot0 + i1 - c1;
    if (tid == 0)
      s_gap = 0x7fffffff;
    const int n = W <= 16 ? tot0 : tot0 + __shfl(i1, 63);
    bool accept = n >= min_inliers;  // handle_search.cpp:47-48
    if (accept && n > kCap)
    {
      if (tid == 0)
      {
        counts->error = 1;
        if (host_counts)
          host_counts[2] = 1;
      }
      break;
    }
    int kept = 0;
    int out_j = 0;  // n <= 64: lane k holds the k-th inlier of the sorted list
    if (accept && n <= 64)
    {
      // Everything in registers (round 4): one inlier per lane, no LDS list, no barrier.  The members' indices reach the lanes by a
      // scalar walk over the set bits of the (wave-uniform) non-empty words -- n short iterations; the LDS version paid a
      // serial loop per WORD lane (most lanes idle) with three dependent LDS reads and two LDS writes per member, then three
      // work-group barriers and LDS atomics for the gap: ~1.8 us per seed, 343 us for the 184 such seeds of the pipeline's
      // 499 hands (profiles/r04_host_timeline_pipeline.txt).
      int my_j = 0, cnt = 0;
      for (int half = 0; half < (W > 64 ? 2 : 1); half++)
      {
        const unsigned long long mm = half ? m1 : m0;
        unsigned long long nz = __ballot(mm != 0ull);
        const int lo32 = (int) (unsigned) (mm & 0xffffffffull), hi32 = (int) (unsigned) (mm >> 32);
        while (nz)
        {
          const int w = __ffsll((long long) nz) - 1;
          nz &= nz - 1ull;
          unsigned long long mw = ((unsigned long long) (unsigned) __builtin_amdgcn_readlane(hi32, w) << 32) |
                                  (unsigned long long) (unsigned) __builtin_amdgcn_readlane(lo32, w);
          const int jbase = (half * 64 + w) * 64;
          while (mw)
          {
            const int b = __ffsll((long long) mw) - 1;
            mw &= mw - 1ull;
            if (tid == cnt)
              my_j = jbase + b;
            cnt++;
          }
        }
      }
      double de = 0.0;
      if (tid < n)
      {
        double ia[3], d[3];
        for (int r = 0; r < 3; r++)
        {
          ia[r] = SMALL ? hpos[i][r] : hands[i].axis[r];
          const double ibr = SMALL ? hpos[i][3 + r] : hands[i].bottom[r];
          d[r] = (SMALL ? hpos[my_j][3 + r] : hands[my_j].bottom[r]) - ibr;
        }
        de = dot3d(ia, d);  // dist_along_line (:34)
      }
      // rank by (distance, index): std::sort's order, ties by index (the oracle's stated choice); the other entries arrive as
      // wave-uniform scalars (v_readlane)
      const int dlo = __double2loint(de), dhi = __double2hiint(de);
      int rank = 0;
      for (int k = 0; k < n; k++)
      {
        const double dk = __hiloint2double(__builtin_amdgcn_readlane(dhi, k), __builtin_amdgcn_readlane(dlo, k));
        const int jk = __builtin_amdgcn_readlane(my_j, k);
        rank += (dk < de || (dk == de && jk < my_j)) ? 1 : 0;
      }
      // the sorted list, one entry per lane: lane e sends its entry to lane rank(e) (the ranks of the n active lanes are a
      // permutation of 0 .. n - 1); idle lanes park theirs on themselves
      const int dst = (tid < n ? rank : tid) * 4;
      const int slo = __builtin_amdgcn_ds_permute(dst, dlo), shi = __builtin_amdgcn_ds_permute(dst, dhi);
      out_j = __builtin_amdgcn_ds_permute(dst, my_j);
      const double sdv = __hiloint2double(shi, slo);
      const double nx_d = __hiloint2double(__shfl_down(shi, 1), __shfl_down(slo, 1));
      const unsigned long long gm = __ballot(tid + 1 < n && nx_d - sdv > 0.02);  // shortenHandle: first gap > 2 cm (:95-99)
      kept = gm ? __ffsll((long long) gm) - 1 : n;  // the elements before the gap position (:111)
      accept = kept >= min_inliers && kept > 0;
      if (accept)
      {
        const double s0 = __hiloint2double(__builtin_amdgcn_readlane(shi, 0), __builtin_amdgcn_readlane(slo, 0));
        const double s1 = __hiloint2double(__builtin_amdgcn_readlane(shi, kept - 1), __builtin_amdgcn_readlane(slo, kept - 1));
        const double mn = s0 < 10000000 ? s0 : 10000000;      // :62-72, the reference's +-1e7 start values
        const double mx = s1 > -10000000 ? s1 : -10000000;
        accept = (mx - mn > min_length);
      }
    }
    else if (accept)
    {
      {
        double ia[3], ib[3];
        for (int r = 0; r < 3; r++)
        {
          ia[r] = SMALL ? hpos[i][r] : hands[i].axis[r];
          ib[r] = SMALL ? hpos[i][3 + r] : hands[i].bottom[r];
        }
        for (int half = 0; half < 2; half++)
        {
          unsigned long long m = half ? m1 : m0;
          int pos = half ? ofs1 : ofs0;
          const int wbase = (half ? 64 + tid : tid) * 64;
          while (m)
          {
            const int j = wbase + __ffsll((long long) m) - 1;
            m &= m - 1ull;
            double d[3];
            for (int r = 0; r < 3; r++)
              d[r] = (SMALL ? hpos[j][3 + r] : hands[j].bottom[r]) - ib[r];
            ld[pos] = dot3d(ia, d);  // dist_along_line (:34)
            lj[pos] = j;
            pos++;
          }
        }
      }
      AGH_WAVE_SYNC();
      // rank sort by (distance, index)
      for (int e = tid; e < n; e += 64)
      {
        const double de = ld[e];
        const int je = lj[e];
        int rank = 0;
        for (int k = 0; k < n; k++)
          rank += (ld[k] < de || (ld[k] == de && lj[k] < je)) ? 1 : 0;
        sd[rank] = de;
        sj[rank] = je;
      }
      AGH_WAVE_SYNC();
      for (int k = tid; k + 1 < n; k += 64)  // shortenHandle: first gap > 2 cm (:95-99)
        if (sd[k + 1] - sd[k] > 0.02)
          atomicMin(&s_gap, k);
      AGH_WAVE_SYNC();
      kept = s_gap == 0x7fffffff ? n : s_gap;  // the elements before the gap position (:111)
      accept = kept >= min_inliers && kept > 0;
      if (accept)
      {
        const double mn = sd[0] < 10000000 ? sd[0] : 10000000;
        const double mx = sd[kept - 1] > -10000000 ? sd[kept - 1] : -10000000;
        accept = (mx - mn > min_length);
      }
    }
    if (accept)
    {
      const int h = s_nh, base = s_nidx;
      for (int k = tid; k < kept; k += 64)
      {
        const int j = n <= 64 ? out_j : sj[k];
        inlier_idx[base + k] = j;
        if (host_idx && base + k < host_idx_cap)
          host_idx[base + k] = j;  // (the host-buffer entry point: the list is on the host when the stream drains)
        atomicAnd(&alive[j >> 6], ~(1ull << (j & 63)));  // :75-78
      }
      if (tid == 0)
      {
        h_first[h] = base;
        h_n[h] = kept;
        s_nh = h + 1;
        s_nidx = base + kept;
      }
    }
    AGH_WAVE_SYNC();
    const int nxt = next_seed(i + 1);  // (retiring the members may have removed the guessed seed)
    if (nxt == guess)
    {
      row0 = nrow0;
      row1 = nrow1;
    }
    else
      load_row(nxt, row0, row1);
    i = nxt;
  }
  AGH_WAVE_SYNC();
  if (tid == 0)
  {
    counts->n_handles = s_nh;
    counts->n_idx = s_nidx;
    if (host_counts)
    {
      host_counts[0] = s_nh;
      host_counts[1] = s_nidx;
    }
  }
}


// K5b', the walk of handle_search.cpp:11-80 with its seeds evaluated SIXTEEN AT A TIME (round 4).  The walk is sequential in its
// seeds, but a seed only ever interacts with the hands of its own row of the pair matrix: a REJECTED seed changes nothing, an
// accepted one retires members of its row.  So the next sixteen open seeds (in index order) are evaluated in parallel against
// the availability at the start of the round, one wave each, everything in registers (one inlier per lane: this kernel serves
// pair matrices whose rows hold at most 64 hands, k_handle_greedy the others), and then committed AS IF in index order:
//   * a seed that shares a hand with an EARLIER seed of the batch which was accepted in this round (its evaluation may be
//     stale, or the seed itself is gone) or which stays open itself (whatever that one may yet retire must not be judged
//     before it) stays open: it is evaluated again next round, after everything it depends on;
//   * every other seed's evaluation is exactly what the sequential walk would have computed at its turn: it is committed
//     (rejected for good, or accepted: members retired).  "Shares a hand" is a test on the rows & availability (every row
//     holds its own seed), a 16 x 16 relation built by the waves in parallel; the open / committed decision is then a
//     sixteen-step recurrence on two scalar masks.
// Seeds committed out of index order belong to disjoint rows, so the handles they produce are the sequential walk's handles;
// only their ORDER of discovery differs, and the epilogue sorts the accepted handles by seed index and lays the inlier
// lists out in that order (handle_search.cpp:79 appends in seed order).  One work-group of 16 waves; the sequential kernel
// spent ~1.5 us per seed in one wave's dependent LDS / cross-lane chain (305 us for the 192 seeds of the pipeline's 499 hands).
template <bool SMALL>
__global__ __launch_bounds__(1024) void k_handle_batch(const agh_hypothesis* __restrict__ hands, int H,
  const unsigned long long* __restrict__ bits, int W, const int* __restrict__ rowcnt, int min_inliers, double min_length,
  int* __restrict__ h_first, int* __restrict__ h_n, int* __restrict__ inlier_idx, HandleCounts* __restrict__ counts,
  int* __restrict__ host_idx, int host_idx_cap, int* __restrict__ host_counts, int* __restrict__ tmp, const int* __restrict__ d_H)
{
  constexpr int kBatch = 16;
  if (d_H)
  {
    H = *d_H;
    W = (H + 63) >> 6;
    if (H > kHandleMaxHands)  // loud, by the general variant alone
    {
      if (!SMALL && threadIdx.x == 0)
      {
        counts->n_handles = 0;
        counts->n_idx = 0;
        counts->error = 2;
        counts->sequential = 0;
        if (host_counts)
          host_counts[2] = 2;
      }
      return;
    }
    if ((H <= kHandleLdsHands) != SMALL)  // (both variants are launched; the other one's case)
      return;
  }
  __shared__ unsigned long long alive[128], elig[128], done[128];  // W <= 128 (H <= 8192)
  __shared__ double hpos[SMALL ? kHandleLdsHands : 1][6];  // axis, bottom
  __shared__ unsigned long long lbits[SMALL ? kHandleLdsHands * kHandleLdsWords : 1];
  __shared__ int cand[kBatch], res_acc[kBatch], res_kept[kBatch];
  __shared__ unsigned long long rowm[kBatch][128];  // row & availability of the round's candidates
  __shared__ unsigned imask[kBatch];
  __shared__ int c_open[kBatch], c_h[kBatch], c_base[kBatch];
  // a candidate's members: up to kMaxRow = 128 hands, two per lane (element p of a wave's lists lives in lane p & 63, slot p >> 6).
  // Rows of more than 64 hands used to send the WHOLE search to the sequential kernel (5 ms for the 3 260-hand stress case).
  constexpr int kMaxRow = 128, kE = kMaxRow / 64;
  __shared__ unsigned short mlist[kBatch][kMaxRow];  // members in bit order; reused for the sorted member list
  __shared__ double sdist[kBatch][kMaxRow];          // the members' distances along the seed's axis, sorted
  __shared__ int n_cand, s_nh, s_nidx, s_maxrow, tot_nh, tot_nidx;
  // per accepted handle, in commit order: seed, offset of its list in tmp_idx, length; then the handles by ascending seed
  // (in LDS: the epilogue's rank and prefix loops would otherwise be chains of dependent global loads)
  // All in LDS, as 16-bit words (H <= 8192): a global store in the commit phase makes the round's closing barrier wait for
  // the memory system (~2 us per round, measured), and the epilogue's loops would be chains of dependent global loads.
  constexpr int kHandlesCap = SMALL ? kHandleLdsHands : 8192;
  __shared__ unsigned short tmp_seed[kHandlesCap], tmp_base[kHandlesCap], tmp_n[kHandlesCap], tmp_order[kHandlesCap];
  __shared__ unsigned short tmp_idx[kHandlesCap];  // inlier lists in commit order
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  (void) tmp;
  for (int t = tid; t < 128; t += 1024)
  {
    alive[t] = 0ull;
    elig[t] = 0ull;
    done[t] = 0ull;
  }
  if (tid == 0)
  {
    s_nh = 0;
    s_nidx = 0;
    s_maxrow = 0;
    counts->n_handles = 0;  // (this kernel is the first of a search to touch the counters: no memset launch in front of it)
    counts->n_idx = 0;
    counts->error = 0;
    counts->sequential = 0;
  }
  __syncthreads();
  for (int j = tid; j < H; j += 1024)
  {
    if (hands[j].width != -1.0)  // handle_search.cpp:13,25: width -1 marks a retired hand
      atomicOr(&alive[j >> 6], 1ull << (j & 63));
    const int rc = rowcnt[j];
    if (rc >= min_inliers)
    {
      atomicOr(&elig[j >> 6], 1ull << (j & 63));
      atomicMax(&s_maxrow, rc);
    }
    if (SMALL)
      for (int r = 0; r < 3; r++)
      {
        hpos[j][r] = hands[j].axis[r];
        hpos[j][3 + r] = hands[j].bottom[r];
      }
  }
  if (SMALL)
    for (int k = tid; k < H * W; k += 1024)
      lbits[k] = bits[k];
  __syncthreads();
  if (s_maxrow > kMaxRow)  // a row longer than two waves' worth: the sequential kernel (launched next) does this search
  {
    if (tid == 0)
    {
      counts->sequential = 1;
      if (host_counts)
        host_counts[3] = 1;
    }
    return;
  }
  auto load_row = [&](int seed, unsigned long long& r0, unsigned long long& r1) {  // lane t: words t and 64 + t of the row
    if (SMALL)
    {
      r0 = lane < W ? lbits[seed * W + lane] : 0ull;
      r1 = 0ull;  // (W <= 10)
    }
    else
    {
      r0 = lane < W ? bits[(int64_t) seed * W + lane] : 0ull;
      r1 = 64 + lane < W ? bits[(int64_t) seed * W + 64 + lane] : 0ull;
    }
  };
#ifdef AGH_DEBUG_HOOKS
  __shared__ long long stamps[64][6];
  int dbg_round = 0;
#define AGH_HSTAMP(i) do { if (tid == 0 && dbg_round < 64) stamps[dbg_round][i] = wall_clock64(); } while (0)
  if (tid == 0)
    stamps[63][5] = wall_clock64();
#else
#define AGH_HSTAMP(i) do { } while (0)
#endif
  for (;;)
  {
    AGH_HSTAMP(0);
    // ---- the next open seeds, in index order (wave 0) ----
    if (wave == 0)
    {
      int nc = 0, my_cand = 0;
      for (int half = 0; half < (W > 64 ? 2 : 1) && nc < kBatch; half++)
      {
        const int wi = half * 64 + lane;
        const unsigned long long todo = wi < W ? (alive[wi] & elig[wi] & ~done[wi]) : 0ull;
        unsigned long long nz = __ballot(todo != 0ull);
        const int lo32 = (int) (unsigned) (todo & 0xffffffffull), hi32 = (int) (unsigned) (todo >> 32);
        while (nz && nc < kBatch)
        {
          const int w = __ffsll((long long) nz) - 1;
          nz &= nz - 1ull;
          unsigned long long mw = ((unsigned long long) (unsigned) __builtin_amdgcn_readlane(hi32, w) << 32) |
                                  (unsigned long long) (unsigned) __builtin_amdgcn_readlane(lo32, w);
          while (mw && nc < kBatch)
          {
            const int b = __ffsll((long long) mw) - 1;
            mw &= mw - 1ull;
            if (lane == nc)
              my_cand = (half * 64 + w) * 64 + b;
            nc++;
          }
        }
      }
      if (lane < nc)
        cand[lane] = my_cand;
      if (lane == 0)
        n_cand = nc;
    }
    __syncthreads();
    const int nc = n_cand;
    if (nc == 0)
      break;
    AGH_HSTAMP(1);
    // ---- evaluation: wave k takes candidate k against the availability as it stands now ----
    unsigned long long m0 = 0ull, m1 = 0ull;
    int kept = 0;
    bool accept = false;
    if (wave < nc)
    {
      const int i = cand[wave];
      unsigned long long row0, row1;
      load_row(i, row0, row1);
      m0 = lane < W ? (row0 & alive[lane]) : 0ull;
      m1 = 64 + lane < W ? (row1 & alive[64 + lane]) : 0ull;
      // (the sharing test below assumes a row holds its own seed.  k_handle_pairs sets the diagonal only if the hand's axis and
      // approach pass the alignment test against themselves, i.e. for unit vectors; for hands a caller did not normalise a seed
      // retired by an earlier candidate of the round could share no bit with it and be committed from its stale evaluation --
      // ADVICE r4.  The seed's own bit is OR-ed into the mask the sharing test reads, not into the member set.)
      if (lane < W)
        rowm[wave][lane] = m0 | (lane == (i >> 6) ? (1ull << (i & 63)) : 0ull);
      if (64 + lane < W)
        rowm[wave][64 + lane] = m1 | (64 + lane == (i >> 6) ? (1ull << (i & 63)) : 0ull);
      const int n = wave_allsum_i32(__popcll(m0) + __popcll(m1));  // <= kMaxRow (s_maxrow)
      accept = n >= min_inliers;  // handle_search.cpp:47-48
      if (accept)
      {
        // one inlier per lane.  The holder of bit b of word w is member number base(w) + popcount(bits below b): it drops its
        // index into the wave's list at that position, and lane e picks up entry e (a scalar walk over the set bits, n steps
        // per seed, kept the CU's one scalar unit busy for all sixteen waves: the evaluation phase was 2.6 us)
        int cnt = 0;
        for (int half = 0; half < (W > 64 ? 2 : 1); half++)
        {
          const unsigned long long mm = half ? m1 : m0;
          unsigned long long nz = __ballot(mm != 0ull);
          const int lo32 = (int) (unsigned) (mm & 0xffffffffull), hi32 = (int) (unsigned) (mm >> 32);
          while (nz)
          {
            const int w = __ffsll((long long) nz) - 1;
            nz &= nz - 1ull;
            const unsigned long long mw = ((unsigned long long) (unsigned) __builtin_amdgcn_readlane(hi32, w) << 32) |
                                          (unsigned long long) (unsigned) __builtin_amdgcn_readlane(lo32, w);
            if ((mw >> lane) & 1ull)
              mlist[wave][cnt + __popcll(mw & ((1ull << lane) - 1ull))] = (unsigned short) ((half * 64 + w) * 64 + lane);
            cnt += __popcll(mw);
          }
        }
        AGH_WAVE_SYNC();
        // (E = slots per lane: 1 when no row of this search holds more than 64 hands -- the common case pays for one)
        auto evaluate = [&](auto Ec) {
          constexpr int E = decltype(Ec)::value;
          int mj[E];
          double de[E];
#pragma unroll
          for (int e = 0; e < E; e++)
          {
            const int pe = lane + 64 * e;
            mj[e] = pe < n ? (int) mlist[wave][pe] : 0;
            de[e] = 0.0;
            if (pe < n)
            {
              double ia[3], d[3];
              for (int r = 0; r < 3; r++)
              {
                ia[r] = SMALL ? hpos[i][r] : hands[i].axis[r];
                const double ibr = SMALL ? hpos[i][3 + r] : hands[i].bottom[r];
                d[r] = (SMALL ? hpos[mj[e]][3 + r] : hands[mj[e]].bottom[r]) - ibr;
              }
              de[e] = dot3d(ia, d);  // dist_along_line (:34)
            }
          }
          // rank by (distance, index): std::sort's order, ties by index (the oracle's stated choice)
          int rank[E];
#pragma unroll
          for (int e = 0; e < E; e++)
            rank[e] = 0;
#pragma unroll
          for (int f = 0; f < E; f++)  // against the elements of slot f, read lane by lane
          {
            const int dlo = __double2loint(de[f]), dhi = __double2hiint(de[f]);
            const int nf = min(64, n - 64 * f);
            for (int k = 0; k < nf; k++)
            {
              const double dk = __hiloint2double(__builtin_amdgcn_readlane(dhi, k), __builtin_amdgcn_readlane(dlo, k));
              const int jk = __builtin_amdgcn_readlane(mj[f], k);
#pragma unroll
              for (int e = 0; e < E; e++)
                rank[e] += (dk < de[e] || (dk == de[e] && jk < mj[e])) ? 1 : 0;
            }
          }
          AGH_WAVE_SYNC();  // (every lane has read its members: the list is rewritten in sorted order)
#pragma unroll
          for (int e = 0; e < E; e++)
            if (lane + 64 * e < n)
            {
              sdist[wave][rank[e]] = de[e];
              mlist[wave][rank[e]] = (unsigned short) mj[e];
            }
          AGH_WAVE_SYNC();
          // shortenHandle: the first gap of more than 2 cm between neighbours of the sorted list (:95-99); the elements before
          // the gap position stay (:111)
          kept = n;
#pragma unroll
          for (int e = E - 1; e >= 0; e--)
          {
            const int pe = lane + 64 * e;
            const unsigned long long gm = __ballot(pe + 1 < n && sdist[wave][min(pe + 1, kMaxRow - 1)] - sdist[wave][pe] > 0.02);
            kept = gm ? 64 * e + __ffsll((long long) gm) - 1 : kept;
          }
          accept = kept >= min_inliers && kept > 0;
          if (accept)
          {
            const double s0 = sdist[wave][0], s1 = sdist[wave][kept - 1];
            const double mn = s0 < 10000000 ? s0 : 10000000;  // :62-72, the reference's +-1e7 start values
            const double mx = s1 > -10000000 ? s1 : -10000000;
            accept = (mx - mn > min_length);
          }
        };
        if (s_maxrow <= 64)
          evaluate(std::integral_constant<int, 1>{});
        else
          evaluate(std::integral_constant<int, kE>{});
      }
      if (lane == 0)
      {
        res_acc[wave] = accept ? 1 : 0;
        res_kept[wave] = accept ? kept : 0;
      }
    }
    __syncthreads();
    AGH_HSTAMP(2);
    // ---- which earlier candidates of the batch does mine share a hand with?  (every row holds its own seed, so shared hands
    // cover "is retired by", "retires a member of" and "depends on the same hands as" alike) ----
    if (wave < nc)
    {
      unsigned im = 0u;
      for (int w0 = 0; w0 < W; w0 += 4)  // lane = earlier candidate (16) x word of the chunk (4)
      {
        const int kq = lane & 15, w = w0 + (lane >> 4);
        const bool hit = kq < wave && w < W && (rowm[kq][w] & rowm[wave][w]) != 0ull;
        const unsigned long long b = __ballot(hit);
        im |= (unsigned) ((b | (b >> 16) | (b >> 32) | (b >> 48)) & 0xffffull);
      }
      if (lane == 0)
        imask[wave] = im;
    }
    __syncthreads();
    AGH_HSTAMP(3);
    // ---- commit: candidate k stays open iff it shares a hand with an earlier candidate that stays open itself (whatever that
    // one may yet retire must not be judged first) or that was accepted (its evaluation may be stale, or it is gone).  Sixteen
    // steps on two scalar masks, by wave 0 alone (run by every wave, the CU's single scalar unit made this 3 us); the
    // accepted commits' handle numbers and list offsets are a 16-lane prefix sum. ----
    if (wave == 0)
    {
      const int v_acc = lane < nc ? res_acc[lane] : 0, v_kept = lane < nc ? res_kept[lane] : 0;
      const int v_im = lane < nc ? (int) imask[lane] : 0;
      unsigned pend = 0u, accm = 0u;
      for (int k = 0; k < nc; k++)
      {
        const bool a = __builtin_amdgcn_readlane(v_acc, k) != 0;
        const bool open = ((unsigned) __builtin_amdgcn_readlane(v_im, k) & (pend | accm)) != 0u;
        pend |= open ? (1u << k) : 0u;
        accm |= (a && !open) ? (1u << k) : 0u;
      }
      const int mine = ((accm >> lane) & 1u) ? v_kept : 0;
      int incl = mine;  // inclusive prefix over the first DPP row (16 lanes)
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);  // row_shr:1
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);  // row_shr:2
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);  // row_shr:4
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);  // row_shr:8
      if (lane < nc)
      {
        c_open[lane] = (pend >> lane) & 1u;
        c_h[lane] = __popc(accm & ((1u << lane) - 1u));
        c_base[lane] = incl - mine;
      }
      if (lane == 15)
      {
        tot_nh = __popc(accm);
        tot_nidx = incl;
      }
    }
    __syncthreads();
    if (wave < nc && !c_open[wave])
    {
      const int t = cand[wave];
      if (lane == 0)
        atomicOr(&done[t >> 6], 1ull << (t & 63));
      if (accept)
      {
        const int h = s_nh + c_h[wave], base = s_nidx + c_base[wave];
        for (int pe = lane; pe < kept; pe += 64)
        {
          const int oj = (int) mlist[wave][pe];  // (the wave's sorted member list)
          tmp_idx[base + pe] = (unsigned short) oj;
          atomicAnd(&alive[oj >> 6], ~(1ull << (oj & 63)));  // :75-78
        }
        if (lane == 0)
        {
          tmp_seed[h] = (unsigned short) t;
          tmp_base[h] = (unsigned short) base;
          tmp_n[h] = (unsigned short) kept;
        }
      }
    }
    __syncthreads();
    AGH_HSTAMP(4);
#ifdef AGH_DEBUG_HOOKS
    if (tid == 0 && dbg_round < 64)
      stamps[dbg_round][5] = nc;
    dbg_round++;
#endif
    if (tid == 0)
    {
      s_nh += tot_nh;
      s_nidx += tot_nidx;
    }
    // (the next round's first barrier orders this update before any reader)
  }
  // ---- epilogue: the handles in seed order, their lists laid end to end in that order ----
  __syncthreads();
  const int nh = s_nh;
  for (int h = tid; h < nh; h += 1024)
  {
    const int sh = tmp_seed[h];
    int rank = 0;
    for (int q = 0; q < nh; q++)
      rank += tmp_seed[q] < sh ? 1 : 0;
    tmp_order[rank] = (unsigned short) h;
  }
  __syncthreads();
  for (int r = wave; r < nh; r += 16)
  {
    const int h = tmp_order[r];
    int first = 0;  // lengths of the handles in front of this one
    for (int q = lane; q < r; q += 64)
      first += tmp_n[tmp_order[q]];
    first = wave_allsum_i32(first);
    const int n = tmp_n[h], base = tmp_base[h];
    if (lane == 0)
    {
      h_first[r] = first;
      h_n[r] = n;
    }
    for (int p = lane; p < n; p += 64)
    {
      const int j = tmp_idx[base + p];
      inlier_idx[first + p] = j;
      if (host_idx && first + p < host_idx_cap)
        host_idx[first + p] = j;  // (the host-buffer entry point: the list is on the host when the stream drains)
    }
  }
  if (tid == 0)
  {
    counts->n_handles = nh;
    counts->n_idx = s_nidx;
    if (host_counts)
    {
      host_counts[0] = nh;
      host_counts[1] = s_nidx;
    }
#ifdef AGH_DEBUG_HOOKS
    const long long t_end = wall_clock64(), t0 = stamps[63][5];
    printf("k_handle_batch H=%d rounds=%d: setup %lld, total %lld (10 ns ticks)\n", H, dbg_round, stamps[0][0] - t0, t_end - t0);
    for (int r = 0; r < dbg_round && r < 40; r++)
      printf("  round %d nc=%lld: select %lld eval %lld share %lld commit %lld\n", r, stamps[r][5], stamps[r][1] - stamps[r][0],
        stamps[r][2] - stamps[r][1], stamps[r][3] - stamps[r][2], stamps[r][4] - stamps[r][3]);
#endif
  }
}
#undef AGH_HSTAMP

__global__ __launch_bounds__(64) void k_handle_build(const agh_hypothesis* __restrict__ hands, const int* __restrict__ h_first,
  const int* __restrict__ h_n, const int* __restrict__ inlier_idx, const HandleCounts* __restrict__ counts,
  agh_handle* __restrict__ out, agh_handle* __restrict__ host_out, int host_cap)
{
  // (one wave per handle; a grid smaller than the bound on the handles -- agh_localize -- strides over them)
  const int lane = threadIdx.x;
  const int n_handles = counts->n_handles;
  for (int h = blockIdx.x; h < n_handles; h += gridDim.x)
  {
  const int n = h_n[h];
  const int* in = inlier_idx + h_first[h];
  // axis_mat * axis_mat^T in the oracle's LaneSum64 order (handle.cpp:14-20)
  double m[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
  for (int k = lane; k < n; k += 64)
  {
    const double* a = hands[in[k]].axis;
    m[0] += a[0] * a[0];
    m[1] += a[0] * a[1];
    m[2] += a[0] * a[2];
    m[3] += a[1] * a[1];
    m[4] += a[1] * a[2];
    m[5] += a[2] * a[2];
  }
#pragma unroll
  for (int k = 0; k < 6; k++)
    for (int o = 32; o > 0; o >>= 1)
      m[k] = m[k] + __shfl_xor(m[k], o);
  double axis[3] = { 0.0, 0.0, 0.0 };
  if (lane == 0)
  {
    double A[3][3] = { { m[0], m[1], m[2] }, { m[1], m[3], m[4] }, { m[2], m[4], m[5] } };
    double V[3][3], dd[3];
    jacobi3_serial(A, V, dd);
    int mx = 0;
    for (int r = 1; r < 3; r++)
      if (dd[r] > dd[mx])
        mx = r;  // maxCoeff: first maximum (handle.cpp:24)
    for (int r = 0; r < 3; r++)
      axis[r] = V[r][mx];
    if (dot3d(axis, hands[in[0]].axis) < 0)  // the sign the oracle fixes (EigenSolver's is arbitrary)
      for (int r = 0; r < 3; r++)
        axis[r] *= -1.0;
  }
  for (int r = 0; r < 3; r++)
    axis[r] = __shfl(axis[r], 0);
  // dist_along_handle, its extremes, the inlier nearest the middle (handle.cpp:28-56)
  double dmin = INFINITY, dmax = -INFINITY;
  for (int k = lane; k < n; k += 64)
  {
    const double al = dot3d(axis, hands[in[k]].bottom);
    dmin = fmin(dmin, al);
    dmax = fmax(dmax, al);
  }
  for (int o = 32; o > 0; o >>= 1)
  {
    dmin = fmin(dmin, __shfl_xor(dmin, o));
    dmax = fmax(dmax, __shfl_xor(dmax, o));
  }
  const double center_dist = (dmax + dmin) / 2.0;
  double best = 10000000;
  int best_k = 0x7fffffff;
  for (int k = lane; k < n; k += 64)
  {
    const double dist = fabs(dot3d(axis, hands[in[k]].bottom) - center_dist);
    if (dist < best)  // strict: the first minimum of this lane's (ascending) positions
    {
      best = dist;
      best_k = k;
    }
  }
  for (int o = 32; o > 0; o >>= 1)
  {
    const double ob = __shfl_xor(best, o);
    const int ok = __shfl_xor(best_k, o);
    if (ok != 0x7fffffff && (best_k == 0x7fffffff || ob < best || (ob == best && ok < best_k)))
    {
      best = ob;
      best_k = ok;
    }
  }
  // handle.cpp:66-73 sums the widths in list order: the loads go out together (one per lane), the additions stay sequential
  double wsum64 = 0.0;
  if (n <= 64)
  {
    const double wl = lane < n ? hands[in[lane]].width : 0.0;
    const int wlo = __double2loint(wl), whi = __double2hiint(wl);
    for (int k = 0; k < n; k++)
      wsum64 += __hiloint2double(__builtin_amdgcn_readlane(whi, k), __builtin_amdgcn_readlane(wlo, k));
  }
  if (lane == 0)
  {
    const int min_idx = best_k == 0x7fffffff ? 0 : best_k;
    const agh_hypothesis& c = hands[in[min_idx]];
    double wsum = 0.0;
    if (n > 64)
      for (int k = 0; k < n; k++)  // handle.cpp:66-73: an explicit loop, kept sequential
        wsum += hands[in[k]].width;
    else
      wsum = wsum64;
    agh_handle hd;
    for (int r = 0; r < 3; r++)
    {
      hd.axis[r] = axis[r];
      hd.center[r] = c.bottom[r];
      hd.approach[r] = c.approach[r];
      hd.hands_center[r] = c.surface[r];
    }
    hd.binormal[0] = c.approach[1] * axis[2] - c.approach[2] * axis[1];
    hd.binormal[1] = c.approach[2] * axis[0] - c.approach[0] * axis[2];
    hd.binormal[2] = c.approach[0] * axis[1] - c.approach[1] * axis[0];
    hd.width = wsum / (double) n;
    hd.n_inliers = n;
    hd.first_inlier = h_first[h];
    out[h] = hd;
    if (host_out && h < host_cap)
      host_out[h] = hd;
  }
  }  // (the handles of this wave)
}

// with_sequential: also launch k_handle_greedy, which does the search when k_handle_batch declines it on the device (a row of the
// pair matrix longer than a wave).  Without it a declined search leaves counts[3] = 1 and no result: the caller repeats the call
// with the kernel (agh_find_handles remembers what the previous set of hands needed, so a stream of similar clouds pays the
// ~5 us of an unneeded launch only when it is needed).
int handle_search(Ctx* c, int64_t H, double x1, double x2, int min_inliers, double min_length, hipStream_t st,
  const HandleMirror& hm, bool with_sequential, const int* d_H)
{
  // d_H: the count lives on the device and H is only its bound (agh_localize) -- launches sized for the bound, both LDS
  // variants of the walk queued, each returning at once when the count is the other one's case
  const int Hi = (int) H, W = (Hi + 63) / 64;
  if (Hi == 0)
    return hipMemsetAsync(c->d_h_counts, 0, sizeof(HandleCounts), st) == hipSuccess ? AGH_OK : AGH_ERR_HIP;
  hipLaunchKernelGGL(k_handle_pairs, dim3(Hi), dim3(256), 0, st, (const agh_hypothesis*) c->d_h_hands, Hi, x1, x2,
    c->d_h_bits, W, c->d_h_rowcnt, d_H);
  // the walk: sixteen seeds at a time (rows of at most 64 hands), else -- flagged on the device -- the sequential kernel
  if (Hi <= kHandleLdsHands || d_H)
    hipLaunchKernelGGL(k_handle_batch<true>, dim3(1), dim3(1024), 0, st, (const agh_hypothesis*) c->d_h_hands, Hi,
      (const unsigned long long*) c->d_h_bits, W, (const int*) c->d_h_rowcnt, min_inliers, min_length, c->d_h_first,
      c->d_h_n, c->d_h_idx, reinterpret_cast<HandleCounts*>(c->d_h_counts), hm.idx, hm.idx_cap, hm.counts, c->d_h_tmp, d_H);
  if (Hi > kHandleLdsHands)
    hipLaunchKernelGGL(k_handle_batch<false>, dim3(1), dim3(1024), 0, st, (const agh_hypothesis*) c->d_h_hands, Hi,
      (const unsigned long long*) c->d_h_bits, W, (const int*) c->d_h_rowcnt, min_inliers, min_length, c->d_h_first,
      c->d_h_n, c->d_h_idx, reinterpret_cast<HandleCounts*>(c->d_h_counts), hm.idx, hm.idx_cap, hm.counts, c->d_h_tmp, d_H);
  if (with_sequential)
  {
    if (Hi <= kHandleLdsHands || d_H)
      hipLaunchKernelGGL(k_handle_greedy<true>, dim3(1), dim3(256), 0, st, (const agh_hypothesis*) c->d_h_hands, Hi,
        (const unsigned long long*) c->d_h_bits, W, (const int*) c->d_h_rowcnt, min_inliers, min_length, c->d_h_first,
        c->d_h_n, c->d_h_idx, reinterpret_cast<HandleCounts*>(c->d_h_counts), hm.idx, hm.idx_cap, hm.counts, d_H);
    if (Hi > kHandleLdsHands)
      hipLaunchKernelGGL(k_handle_greedy<false>, dim3(1), dim3(256), 0, st, (const agh_hypothesis*) c->d_h_hands, Hi,
        (const unsigned long long*) c->d_h_bits, W, (const int*) c->d_h_rowcnt, min_inliers, min_length, c->d_h_first,
        c->d_h_n, c->d_h_idx, reinterpret_cast<HandleCounts*>(c->d_h_counts), hm.idx, hm.idx_cap, hm.counts, d_H);
  }
  // (a handle holds at least min_inliers >= 1 hands: H bounds the handles too)
  hipLaunchKernelGGL(k_handle_build, dim3(d_H ? std::min(Hi, 1024) : Hi), dim3(64), 0, st, (const agh_hypothesis*) c->d_h_hands,
    (const int*) c->d_h_first, (const int*) c->d_h_n, (const int*) c->d_h_idx,
    (const HandleCounts*) reinterpret_cast<HandleCounts*>(c->d_h_counts), c->d_h_handles, hm.handles, hm.handle_cap);
  return hipGetLastError() == hipSuccess ? AGH_OK : AGH_ERR_HIP;
}

}  // namespace agh
