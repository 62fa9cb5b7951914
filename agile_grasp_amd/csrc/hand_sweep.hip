// hand_sweep.hip -- K2: hand-frame transform + RotatingHand/FingerHand occupancy sweep, and K4: compaction.
//
// Reference path (src/agile_grasp/...): HandSearch::findHands (private) hand_search.cpp:116-206 (OMP loop B) ->
// kdtree.radiusSearch r=0.08 (147) -> RotatingHand::transformPoints rotating_hand.cpp:19-75 ->
// RotatingHand::evaluateHand 78-177 -> FingerHand::evaluateFingers/evaluateHand/deepenHand/evaluateGraspParameters
// finger_hand.cpp:20-233 -> Antipodal::evaluateGrasp antipodal.cpp:12-86; image of Learning::convertToImage
// learning.cpp:320-365.
//
// One 256-thread workgroup per sample.  The ball's grid rows are read with coalesced float4 loads, filtered with the
// FLANN float32 distance, rotated into the hand frame in fp64 and cropped to |z| < hand_height straight into an LDS
// tile (x', y' as double2 + point id).  The reference then re-scans the cropped points for each of 20 finger slots at
// each of up to 11 bite depths in each of 8 orientations.  Here every point is classified ONCE per orientation:
//   region  = its position among the <= 40 sorted slot thresholds {fs_i, fs_i + w} (exact fp64 comparisons), and
//   depth   = how many bite depths d_k are <= y,
// and one LDS atomic OR records (region, depth) in an 81 x 16 bit table.  "Some point in gap i has y < d_k" and
// "some point beside slot i has y < d_k" are then range-ORs over that table, so evaluateFingers / evaluateHand /
// deepenHand become integer logic on one lane -- bit-identical to the literal sweep (all comparisons are the
// reference's own `<` / `>` on the same doubles; see tests/test_oracle.py::test_reduction_form_equals_literal_sweep).
// A second pass over the LDS tile (only for orientations that produced a hand) yields the grasp width, the
// points-in-box count, the antipodal counts and the 80x100 occupancy image.
#include "agh_internal.h"

#include <cstddef>
#include <type_traits>

#include <hip/hip_ext.h>

namespace agh
{

#ifdef AGH_DEBUG_HOOKS
#define AGH_DBG_AND(x) &&(x)
#else
#define AGH_DBG_AND(x)
#endif

// fmin / fmax for values that are never NaN: one v_min_f64 / v_max_f64.  (The library functions quiet signalling NaNs first
// -- a `v_max_f64 x, x, x` per operand in IEEE mode -- which tripled the cost of the running minima in the point loops.)
__device__ __forceinline__ double min_f64_raw(double a, double b)
{
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double max_f64_raw(double a, double b)
{
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

struct OriState
{
  double T[3][3];  // frame_ * rot^T
  double approach[3], binormal[3];
  double cs, sn;
  double ymin;  // min of y over all cropped points (written after pass A)
  int rejected;
  int has_hand;
  int e;
  int last;
};

// Wave-wide minimum / maximum of a double, the same on every lane: row_shr 1, 2, 4, 8 inside each 16-lane row, row_bcast:15
// into rows 1 and 3, row_bcast:31 into rows 2 and 3 (a lane without a source keeps its own value: min and max are
// idempotent), then lane 63 holds the result.  Two 32-bit DPP moves and one v_min_f64 per step instead of the two
// ds_bpermute round trips of a __shfl_xor butterfly (the finger phase's longest chain).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_or_self_f64(double v)
{
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane63_f64(double v)
{
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ double wave_min_f64(double v)
{
  v = min_f64_raw(v, dpp_or_self_f64<0x111, 0xf>(v));
  v = min_f64_raw(v, dpp_or_self_f64<0x112, 0xf>(v));
  v = min_f64_raw(v, dpp_or_self_f64<0x114, 0xf>(v));
  v = min_f64_raw(v, dpp_or_self_f64<0x118, 0xf>(v));
  v = min_f64_raw(v, dpp_or_self_f64<0x142, 0xa>(v));
  v = min_f64_raw(v, dpp_or_self_f64<0x143, 0xc>(v));
  return lane63_f64(v);
}
__device__ __forceinline__ double wave_max_f64(double v)
{
  v = max_f64_raw(v, dpp_or_self_f64<0x111, 0xf>(v));
  v = max_f64_raw(v, dpp_or_self_f64<0x112, 0xf>(v));
  v = max_f64_raw(v, dpp_or_self_f64<0x114, 0xf>(v));
  v = max_f64_raw(v, dpp_or_self_f64<0x118, 0xf>(v));
  v = max_f64_raw(v, dpp_or_self_f64<0x142, 0xa>(v));
  v = max_f64_raw(v, dpp_or_self_f64<0x143, 0xc>(v));
  return lane63_f64(v);
}
// Inclusive OR scan over the 64 lanes: row_shr 1, 2, 4, 8 inside each 16-lane row, then row_bcast:15 into rows 1 and 3
// and row_bcast:31 into rows 2 and 3 (the compiler fuses each step into one v_or_b32_dpp).
__device__ __forceinline__ unsigned wave_prefix_or(unsigned v)
{
  v |= (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, false);
  v |= (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, false);
  v |= (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, false);
  v |= (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, false);
  v |= (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false);
  v |= (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false);
  return v;
}
// (int) v for any v: truncation, saturation at the int range, NaN -> 0 -- what the instruction does, where the C++ cast of
// an out-of-range value is undefined
__device__ __forceinline__ int cvt_i32_f64_sat(double v)
{
  int r;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(v));
  return r;
}

__device__ __forceinline__ int wave_sum_i32(int v)
{
  return wave_allsum_i32(v);  // (v_permlane swaps and DPP moves, no ds_bpermute)
}

// Minima of EIGHT doubles per lane over the wave in one transposed butterfly: at the strides 32, 16 and 8 a lane hands on the
// half of its values its partner keeps and keeps the other half (v_permlane32_swap / v_permlane16_swap exchange the two halves
// in place, row_ror:8 inside a row), strides 4, 2, 1 finish the one value left.  Lane l ends with the minimum of value
// (l >> 3) & 7 -- 36 instructions, where eight separate wave reductions are 8 x 20.
__device__ __forceinline__ double swap_min32(double a, double b)  // lanes < 32: min over (l, l + 32) of a; lanes >= 32: of b
{
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
  return min_f64_raw(__hiloint2double((int) hi[0], (int) lo[0]), __hiloint2double((int) hi[1], (int) lo[1]));
}
__device__ __forceinline__ double swap_min16(double a, double b)  // even rows: min over (l, l + 16) of a; odd rows: of b
{
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
  return min_f64_raw(__hiloint2double((int) hi[0], (int) lo[0]), __hiloint2double((int) hi[1], (int) lo[1]));
}
__device__ __forceinline__ double wave_min8_f64(const double (&m)[8], int lane)
{
  const double v0 = swap_min32(m[0], m[4]), v1 = swap_min32(m[1], m[5]), v2 = swap_min32(m[2], m[6]), v3 = swap_min32(m[3], m[7]);
  const double w0 = swap_min16(v0, v2), w1 = swap_min16(v1, v3);  // value (j) + 2 (l >> 4 & 1) + 4 (l >> 5)
  const bool up = (lane & 8) != 0;
  double x = min_f64_raw(up ? w1 : w0, xor_partner_f64<8>(up ? w0 : w1));
  x = min_f64_raw(x, xor_partner_f64<4>(x));
  x = min_f64_raw(x, xor_partner_f64<2>(x));
  x = min_f64_raw(x, xor_partner_f64<1>(x));
  return x;
}

// Cropped points staged in LDS at a time: 2176 x double2 (34 KiB) for the online path, 1728 x (double2 + point id)
// (34 KiB) when per-point normals are needed for the antipodal test -- both leave room for 3 blocks per CU.

__device__ __forceinline__ unsigned lowmask(int n)
{
  return n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
}

// PX / PY: look-up probes per finger-slot threshold cell / bite-depth cell (HandGeom::x_probes, y_probes): the default
// hand needs 2 and 1, any admissible geometry at most kLutProbe.
// MODE: 0 = occupancy sweep only; 1 = with the antipodal counts (cloud normals); 2 = mode 1 plus one image per camera
// for the training instances createInstance(h, cam_pos, cam = 0 / 1) (learning.cpp:389-397).
// NT: threads of a work-group -- 256 (four waves, two orientations each, three work-groups per CU) or 512 (eight waves, an
// orientation each, two work-groups per CU with a tile that holds every neighbourhood of the voxelised clouds in one piece).
// WG4: four work-groups per CU with a 1408-point tile instead of three with 2176 points -- for launches of many rounds of
// work-groups (C4, a batch of clouds), where the fourth resident work-group is worth more than the tile (round 6, after the parking
// of multi-tile neighbourhoods was gone: C4 228 -> 214 us; at C2's 2.6 rounds the two are equal, and round 5 -- with the parking --
// had measured 97 against 73 us).
template <int MODE, int PX, int PY, int NT, int WG4>
#ifndef AGH_SWEEP_WGS
#define AGH_SWEEP_WGS 3
#endif
#ifndef AGH_SWEEP_TILE
#define AGH_SWEEP_TILE 2176
#endif
__global__ __launch_bounds__(NT, NT == 512 ? 4 : (WG4 ? 4 : AGH_SWEEP_WGS)) void k_hand_sweep(GridView gv, const HandGeom* __restrict__ geom_p,
  const agh_frame* __restrict__ frames, const int32_t* __restrict__ samples, const int32_t* __restrict__ cam_source,
  int S, float r2f, double rpad, const double* __restrict__ normals, double img_cell,
  int32_t* __restrict__ status, agh_hypothesis* __restrict__ slots, uint32_t* __restrict__ images, int debug_stop, long long* __restrict__ dbg,
  const int* __restrict__ order, uint8_t* __restrict__ vmask, uint32_t* __restrict__ images_cam)
{
  constexpr bool NORMALS = MODE != 0, TRAIN = MODE == 2;
  constexpr int NW = NT / 64, kOW = 8 / NW;  // waves; orientations a wave owns in the finger logic and the results
  constexpr int kImgPlanes = TRAIN ? 16 : 8;  // TRAIN: plane o = camera 0's points, plane 8 + o = camera 1's
  // the block must stay under a third of the CU's 160 KiB (512-B granules)
  constexpr int kTile = TRAIN ? 1280 : (NORMALS ? 1728 : (NT == 512 ? 3712 : (WG4 ? 1408 : AGH_SWEEP_TILE)));
  static_assert(!WG4 || (MODE == 0 && NT == 256), "the four-per-CU form exists for the online variant only");
  __shared__ double2 pts[kTile];
  __shared__ unsigned pid[NORMALS ? kTile : 1];
  __shared__ RowTable rt;
  __shared__ HandGeom G;
  __shared__ double thr_s[64];
  __shared__ double dep_s[24];
  __shared__ OriState ori[8];
  __shared__ unsigned regmask[8][44];
  __shared__ __attribute__((aligned(16))) unsigned pre_s[NW][88];
  __shared__ unsigned suf_s[NW][88];
  __shared__ unsigned img[kImgPlanes][kImageWords + 2];
  // Pass A sets its (region, depth) bits in kRmCopies copies of the table, one per lane & 3, and the finger phase ORs them
  // together: neighbouring lanes hold neighbouring points, which mostly fall into the same table word, and an LDS atomic
  // serialises the lanes of a wave that hit one word (or one bank) -- with a single table the atomic was 45 % of pass A.
  // The copies live in the image planes, which nothing touches before pass B (the finger phase leaves them zeroed again);
  // copy c starts 8 banks after copy c - 1.
  constexpr int kRmCopies = 4, kRmOriStride = 48, kRmCopyStride = 8 * kRmOriStride + 8;
  static_assert(kRmCopies * kRmCopyStride <= kImgPlanes * (kImageWords + 2), "the table copies must fit the image planes");
  unsigned* const rmc = &img[0][0];
  __shared__ double ypart[NW][8];  // a wave's minimum of y over its quarter of the tile, per orientation (screening)
  __shared__ int cnt_crop, any_hand, pending, tile_end;

  // blockIdx -> sample: blocks of 32 samples, heaviest first (K1b's sorter work-group: sweep_order_block), or sample order
  const int s = order ? order[blockIdx.x] : blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#define AGH_STAMP(i) do { if (dbg && tid == 0) dbg[(int64_t) s * 8 + (i)] = wall_clock64(); } while (0)
  AGH_STAMP(0);
  // The work-group's first microseconds are a chain of dependent loads (frame -> grid descriptor -> cell table -> points).
  // Everything that does not depend on the frame is ISSUED before the frame is waited for: the geometry words this thread
  // stages, the sample's grid descriptor, the hand angles.
  constexpr int kGeomWords = (int) (sizeof(HandGeom) / 4), kGeomPer = (kGeomWords + NT - 1) / NT;
  unsigned gw[kGeomPer];
#pragma unroll
  for (int k = 0; k < kGeomPer; k++)
    gw[k] = (tid + NT * k) < kGeomWords ? ((const unsigned*) geom_p)[tid + NT * k] : 0u;
  gv = grid_of_cloud(gv, cloud_of_point(gv, samples[s]));  // the sample's cloud of the batch
  const GridDesc gd = *gv.desc;
  const int to = tid / 9, tij = tid - 9 * to;  // thread (o, i, j) of the 72 entries of frame_ * rot^T (below)
  const double t_cs = tid < 72 ? geom_p->cos_a[to] : 0.0, t_sn = tid < 72 ? geom_p->sin_a[to] : 0.0;
  const agh_frame F = frames[s];
  if (!F.valid)
  {
    // (no records for orientations without a hypothesis: the concatenation reads vmask, never a dead slot)
    if (tid == 0)
    {
      status[s] = kStatusDegenerate;
      vmask[s] = 0;
    }
    return;
  }
  // stage the geometry tables in LDS
#pragma unroll
  for (int k = 0; k < kGeomPer; k++)
    if ((tid + NT * k) < kGeomWords)
      ((unsigned*) &G)[tid + NT * k] = gw[k];
  if (tid == 0)
  {
    cnt_crop = 0;
    any_hand = 0;
    pending = 0;
    tile_end = 0x7fffffff;
  }
  for (int k = tid; k < 8 * 44; k += NT)
    (&regmask[0][0])[k] = 0u;
  for (int k = tid; k < kImgPlanes * (kImageWords + 2); k += NT)
    (&img[0][0])[k] = 0u;
  const float sx = (float) F.sample[0], sy = (float) F.sample[1], sz = (float) F.sample[2];  // hand_search.cpp:141-144
  // frame_ << normal, normal x axis, axis (rotating_hand.cpp:24-25); column r of fr is fr[.][r]
  double fr[3][3];
  {
    const double* nm = F.normal;
    const double* ax = F.axis;
    const double nxa[3] = { nm[1] * ax[2] - nm[2] * ax[1], nm[2] * ax[0] - nm[0] * ax[2], nm[0] * ax[1] - nm[1] * ax[0] };
    for (int r = 0; r < 3; r++)
    {
      fr[r][0] = nm[r];
      fr[r][1] = nxa[r];
      fr[r][2] = ax[r];
    }
  }
  // frame_ * rot^T of the eight orientations (rotating_hand.cpp:89-96), an entry per thread, while the row table's loads are
  // in flight (eight threads used to form all 72 entries one after the other behind it)
  if (tid < 72)
  {
    const int i = tij / 3, j = tij - 3 * i;
    const double rj0 = j == 0 ? t_cs : (j == 1 ? t_sn : 0.0);
    const double rj1 = j == 0 ? -1.0 * t_sn : (j == 1 ? t_cs : 0.0);
    const double rj2 = j == 2 ? 1.0 : 0.0;
    const double fi0 = i == 0 ? fr[0][0] : (i == 1 ? fr[1][0] : fr[2][0]);
    const double fi1 = i == 0 ? fr[0][1] : (i == 1 ? fr[1][1] : fr[2][1]);
    const double fi2 = i == 0 ? fr[0][2] : (i == 1 ? fr[1][2] : fr[2][2]);
    ori[to].T[i][j] = (fi0 * rj0 + fi1 * rj1) + fi2 * rj2;
  }
  // only the rows (and the parts of rows) that can hold a point of the hand's slab |axis . (p - sample)| < hand_height
  build_rows<true>(gv, sx, sy, sz, rpad, rt, F.axis, geom_p->hand_height, &gd);  // ends with barriers: G, ori[].T and counters are visible afterwards
  if (tid < 64)
    thr_s[tid] = tid < G.n_thr ? G.thr[tid] : INFINITY;
  if (tid < 24)
    dep_s[tid] = tid < G.n_depths ? G.depths[tid] : INFINITY;
  if (rt.bad)
  {
    if (tid == 0)
    {
      status[s] = kStatusRows;
      vmask[s] = 0;
    }
    return;
  }
  const double hh = G.hand_height;
  const int total = rt.total;
  // ---- orientation setup (rotating_hand.cpp:86-104) ----
  if (tid < 8)
  {
    const int o = tid;
    OriState& O = ori[o];
    const double cs = G.cos_a[o], sn = G.sin_a[o];  // (O.T = frame_ * rot^T was formed above, an entry per thread)
    double cams[2][3];
    for (int c = 0; c < 2; c++)
      for (int r = 0; r < 3; r++)
        cams[c][r] = G.cam_origin[c][r] - F.sample[r];
    for (int i = 0; i < 3; i++)
    {
      O.approach[i] = (O.T[i][0] * 0.0 + O.T[i][1] * 1.0) + O.T[i][2] * 0.0;
      O.binormal[i] = (O.T[i][0] * 1.0 + O.T[i][1] * 0.0) + O.T[i][2] * 0.0;
    }
    const double d0 = (O.approach[0] * cams[0][0] + O.approach[1] * cams[0][1]) + O.approach[2] * cams[0][2];
    const double d1 = (O.approach[0] * cams[1][0] + O.approach[1] * cams[1][1]) + O.approach[2] * cams[1][2];
    O.rejected = (d0 > 0 && d1 > 0) ? 1 : 0;  // rotating_hand.cpp:99-102
    O.cs = cs;
    O.sn = sn;
    O.has_hand = 0;
    O.e = -1;
    O.last = 0;
    O.ymin = 0.0;
  }
  if (debug_stop == 1)
    return;
  AGH_STAMP(1);

  // Gather: every wave owns the 128-candidate segments j = wave, wave + 4, ... of the concatenated grid rows and walks each
  // with its 64 lanes, two loads in flight per lane (a segment spans one to three rows: coalesced runs).  Each
  // candidate is filtered (FLANN float32 distance), rotated into the hand frame and cropped (rotating_hand.cpp:26,37-51)
  // and the survivors are appended to the LDS tile: one LDS atomic per wave-instruction reserves their slots; if the
  // reservation does not fit, the tile is closed and the wave pauses AT that instruction, so a neighbourhood that does
  // not fit one tile streams through it in several rounds.  The cursor (segment, first row) says where to resume.
  int cur_j = wave, cur_r0 = 0;
  auto gather_reset = [&]() {
    cur_j = wave;
    cur_r0 = 0;
  };
  // Filters two candidates per lane (a 128-candidate row segment per wave) and appends the survivors with ONE
  // reservation; returns false if the tile is full (nothing was appended, the segment must be offered again).
  // Straight-line code: the kernel is bound by the instructions a wave issues, and exec-mask juggling around the
  // ~1/8 of the lanes that survive costs more than computing the hand-frame coordinates for all of them.
  auto classify1 = [&](const float4& p, bool have, bool& inball, bool& keep, double& tx, double& ty) {
    // p - sample in float32 (hand_search.cpp:157-158) serves FLANN's distance too: q - p = -(p - q) exactly, and the squares
    // are the same bits -- three subtractions per candidate less
    const float fx = p.x - sx, fy = p.y - sy, fz = p.z - sz;
    const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(fx, fx), __fmul_rn(fy, fy)), __fmul_rn(fz, fz));  // flann_d2's order
    inball = have & (d2 < r2f);
    const double cx = (double) fx, cy = (double) fy, cz = (double) fz;
    const double tz = (fr[0][2] * cx + fr[1][2] * cy) + fr[2][2] * cz;
    keep = inball & (tz > -1.0 * hh) & (tz < hh);  // (bitwise: no exec-mask regions around three compares)
    tx = (fr[0][0] * cx + fr[1][0] * cy) + fr[2][0] * cz;
    ty = (fr[0][1] * cx + fr[1][1] * cy) + fr[2][1] * cz;
  };
  auto consume2 = [&](const float4& p0, bool h0, const float4& p1, bool h1) -> bool {
    bool in0, in1, k0, k1;
    double tx0, ty0, tx1, ty1;
    classify1(p0, h0, in0, k0, tx0, ty0);
    classify1(p1, h1, in1, k1, tx1, ty1);
    const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
    if (m0 | m1)
    {
      const int c0 = __popcll(m0), cnt = c0 + __popcll(m1);
      int base = 0;
      if (lane == 0)
      {
        // Reservations are totally ordered by the atomic: every one that fits precedes the first that does not, and the
        // counter only grows, so the valid entries are exactly [0, base of the first failed reservation).
        base = atomicAdd(&cnt_crop, cnt);
        if (base + cnt > kTile)
        {
          atomicMin(&tile_end, base);
          base = -1;
        }
      }
      base = __builtin_amdgcn_readfirstlane(base);
      if (base < 0)
        return false;
      const unsigned long long below = (1ull << lane) - 1ull;
      if (k0)
      {
        const int k = base + __popcll(m0 & below);
        pts[k] = make_double2(tx0, ty0);
        if (NORMALS)
          pid[k] = __float_as_uint(p0.w);
      }
      if (k1)
      {
        const int k = base + c0 + __popcll(m1 & below);
        pts[k] = make_double2(tx1, ty1);
        if (NORMALS)
          pid[k] = __float_as_uint(p1.w);
      }
    }
    return true;
  };
  // Fills the tile from the cursors on; returns the tile's point count and whether every wave reached the end.
  // Must be entered with cnt_crop == 0 made visible by a barrier.
  // The walk is software-pipelined: the loads of the NEXT 128 candidates are issued before the current 128 are
  // filtered, so a wave always has two row segments in flight (the gather is bound by L2 latency, not by bandwidth).
  // (the candidates are walked as flat 128-candidate segments over the concatenated rows -- FlatRows, agh_internal.h: full
  // segments and an equal share per wave, where a row per wave left a third of the lanes idle; the row table lives in lane
  // registers and is read with v_readlane: no LDS round trip between a segment and the next one's first load)
  FlatRows flat;
  flat.init(rt, lane);
  const int nseg = flat.segments();
  auto seg_load = [&](int j, int& r0, float4& p0, float4& p1, bool& h0, bool& h1) {
    int a0, a1;
    flat.locate(j, r0, lane, a0, a1, h0, h1);
    // unconditional loads (a lane without a candidate reads element 0 and ignores it): a load under a branch would
    // be waited for at the join, which is exactly the latency the pipeline is there to hide
    p0 = gv.sorted[h0 ? a0 : 0];
    p1 = gv.sorted[h1 ? a1 : 0];
  };
  auto gather_tile = [&](bool& all_done) -> int {
    bool full = false;
    float4 p0, p1;
    bool h0, h1;
    seg_load(cur_j, cur_r0, p0, p1, h0, h1);
    while (cur_j < nseg && !full)
    {
      int nr0 = cur_r0;
      float4 q0, q1;
      bool g0, g1;
      seg_load(cur_j + NW, nr0, q0, q1, g0, g1);  // in flight while the current segment is consumed
      if (!consume2(p0, h0, p1, h1))
        full = true;
      else
      {
        cur_j += NW;
        cur_r0 = nr0;
        p0 = q0;
        p1 = q1;
        h0 = g0;
        h1 = g1;
      }
    }
    if (lane == 0 && cur_j < nseg)
      pending = 1;
    __syncthreads();
    const int c = min(cnt_crop, tile_end);
    all_done = pending == 0;
    return c;
  };
  // Between two tiles of one pass: everybody has finished with the tile, then the counters are cleared.
  auto next_tile = [&]() {
    __syncthreads();
    if (tid == 0)
    {
      cnt_crop = 0;
      pending = 0;
      tile_end = 0x7fffffff;
    }
    __syncthreads();
  };

  const int K = G.n_depths;
  // ---- pass A: classify every cropped point once per orientation that can still produce a hand ----
  // SCREENING.  FingerHand::evaluateFingers returns "no finger" as soon as one point with y < bite also has y < back
  // (finger_hand.cpp:29-42); at the initial bite that is `ymin < depths[0] && ymin < backs[0]`, and an orientation without
  // a hand at the initial bite has no hypothesis (rotating_hand.cpp:111).  So the minimum of y alone -- 4 instructions per
  // point against the ~36 of the (region, depth) classification -- decides most orientations: 73 % of those the camera test
  // leaves at C2 (a table's points reach 8 cm behind a hand that approaches along it).  The minimum only falls from tile to
  // tile, so an orientation that is out stays out, and a work-group whose orientations are all out stops gathering.
  // Every wave screens a quarter of the tile for all orientations still in, the partial minima meet in LDS, and lanes
  // 0..7 of EVERY wave form the same running minima: `live` is uniform over the work-group without a second barrier.
  unsigned classified = 0;  // orientations with bits in the table copies
  unsigned live = 0;  // (set after the first tile's closing barrier: the camera test's flags are visible from there on)
  double run_ymin = INFINITY;  // lane o < 8: min of y over the tiles so far in orientation o
  auto screen = [&](int nc) {
    double m[8];
#pragma unroll
    for (int o = 0; o < 8; o++)
      m[o] = INFINITY;
    double2 pn = pts[tid < nc ? tid : 0];
    for (int b0 = 0; b0 < nc; b0 += NT)
    {
      // (a lane past the end of the tile holds point 0 once more: the minimum is idempotent)
      const double2 p = pn;
      const int tn = b0 + NT + tid;
      pn = pts[tn < nc ? tn : 0];
#pragma unroll
      for (int o = 0; o < 8; o++)
        if ((live >> o) & 1u)  // (uniform)
          m[o] = min_f64_raw(m[o], geom_p->sin_a[o] * p.x + geom_p->cos_a[o] * p.y);  // row y of rot * points_ (rotating_hand.cpp:91)
    }
    const double mine = wave_min8_f64(m, lane);  // lane l: orientation l >> 3 (+inf for one that is out)
    if ((lane & 7) == 0)
      ypart[wave][lane >> 3] = mine;
    __syncthreads();
    bool out = false;
    if (lane < 8)
    {
      for (int w = 0; w < NW; w++)
        run_ymin = min_f64_raw(run_ymin, ypart[w][lane]);
      out = (run_ymin < G.depths[0]) && (run_ymin < G.backs[0]);  // finger_hand.cpp:29-42 at the initial bite
    }
    live &= ~(unsigned) __ballot(out);
  };
  // CLASSIFICATION of the orientations still in, every wave a quarter of the tile's points (all four waves set bits in the
  // same table copies: LDS atomics; the finger phase waits for a barrier).
  // The inner loop is written in STAGES over four points per lane -- rotate, y look-up, x look-up, table update -- with no
  // control flow between the stages: each look-up is a chain of two dependent LDS reads, and with a branch per point (the
  // earlier form) the compiler waited for every read before issuing the next point's, sixteen exposed LDS round trips per
  // iteration instead of four.  Points beyond the deepest bite depth take the x look-up along and are masked at the update.
  auto classify = [&](int nc) {
    for (int o = 0; o < 8; o++)
    {
      if (!((live >> o) & 1u))
        continue;
      const double cs = ori[o].cs, ms = -1.0 * ori[o].sn, sn = ori[o].sn;
      const double ylo = G.ylut_lo, ysc = G.ylut_scale, xlo = G.xlut_lo, xsc = G.xlut_scale;
      // (the four points of the NEXT pass are read while this pass classifies: one LDS round trip less on every pass)
      double2 pn[4];
#pragma unroll
      for (int u = 0; u < 4; u++)
        pn[u] = pts[(tid + NT * u) < nc ? tid + NT * u : 0];
      // one pass over U x 256 points, U points per lane; the tile's last pass takes the U that covers what is left (a pass of
      // four for 1.1 k points classified 2 k)
      auto pass = [&](auto Uc, int b0) {
        constexpr int U = decltype(Uc)::value;
        double xr[U], yr[U];
#pragma unroll
        for (int u = 0; u < U; u++)
        {
          const double2 p = pn[u];
          xr[u] = cs * p.x + ms * p.y;  // rot * points_ (rotating_hand.cpp:91)
          yr[u] = sn * p.x + cs * p.y;
        }
        if (U == 4 && b0 + 4 * NT < nc)
        {
#pragma unroll
          for (int u = 0; u < 4; u++)
          {
            const int tn = b0 + 4 * NT + NT * u + tid;
            pn[u] = pts[tn < nc ? tn : 0];
          }
        }
        // (a lane past the end of the tile holds point 0 once more: the table bits are idempotent, so nothing below needs
        // to know -- no selects, no predicate besides the depth test)
        // depth class yk = #{k : d_k <= y} and region rank c = #{k : thr_k < x} by cell look-up + exact probes
        int ly[U], lx[U];
#pragma unroll
        for (int u = 0; u < U; u++)
        {
          // v_cvt_i32_f64 truncates and saturates (NaN -> 0), so the clamp can follow the conversion as one integer med3
          const int cy = min(max(cvt_i32_f64_sat((yr[u] - ylo) * ysc), 0), 63);
          const int cx = min(max(cvt_i32_f64_sat((xr[u] - xlo) * xsc), 0), 1023);
          ly[u] = G.ylut[cy];
          lx[u] = G.xlut[cx];
        }
        double dv[U][PY], tv[U][PX];
#pragma unroll
        for (int u = 0; u < U; u++)
        {
#pragma unroll
          for (int j = 0; j < PY; j++)
            dv[u][j] = dep_s[ly[u] + j];
#pragma unroll
          for (int j = 0; j < PX; j++)
            tv[u][j] = thr_s[lx[u] + j];
        }
#pragma unroll
        for (int u = 0; u < U; u++)
        {
          int yk = ly[u];
#pragma unroll
          for (int j = 0; j < PY; j++)
            yk += (dv[u][j] <= yr[u]) ? 1 : 0;
          int c = lx[u], e = 0;
#pragma unroll
          for (int j = 0; j < PX; j++)
          {
            c += (tv[u][j] < xr[u]) ? 1 : 0;
            e |= (tv[u][j] == xr[u]) ? 1 : 0;
          }
          const int key = 2 * c + e;
          if ((yk < K) AGH_DBG_AND(debug_stop != 11 && debug_stop != 10))
            atomicOr(&rmc[(lane & (kRmCopies - 1)) * kRmCopyStride + o * kRmOriStride + (key >> 1)], 1u << ((key & 1) * 16 + yk));
        }
      };
      for (int b0 = 0; b0 < nc; b0 += 4 * NT)
      {
        const int left = nc - b0;
        if (left > 3 * NT)
          pass(std::integral_constant<int, 4>{}, b0);
        else if (left > 2 * NT)
          pass(std::integral_constant<int, 3>{}, b0);
        else if (left > NT)
          pass(std::integral_constant<int, 2>{}, b0);
        else
          pass(std::integral_constant<int, 1>{}, b0);
      }
    }
  };
  // A neighbourhood that needs more than one tile is gathered again for pass B -- by the few work-groups that get there: an
  // orientation with a hand is the exception (a fifth of the samples).  (Until round 5 every multi-tile neighbourhood parked its
  // cropped points in global memory during pass A, 14 MB per launch at C2 written to be read back by a fifth of them.)
  int ntiles = 0, nc = 0, ncrop_all = 0;
  for (;;)
  {
    bool all_done = false;
    nc = gather_tile(all_done);
    if (ntiles == 0)
      AGH_STAMP(2);
    if (debug_stop == 2)
      return;
    if (ntiles == 0)
      for (int o = 0; o < 8; o++)
        live |= ori[o].rejected ? 0u : (1u << o);
    screen(nc);
    classified |= live;
    classify(nc);
    ntiles++;
    ncrop_all += nc;
    if (all_done || live == 0)  // (no orientation left: the rest of the neighbourhood cannot change the result)
      break;
    next_tile();
  }
  // an orientation the screening put out is treated like one the camera test rejected from here on: no finger logic, no
  // pass B, no record
  if (wave == 0 && lane < 8)
  {
    ori[lane].ymin = run_ymin;
    if (!((live >> lane) & 1u))
      ori[lane].rejected = 1;
  }
  __syncthreads();  // all four waves wrote the tables of every orientation still in
  AGH_STAMP(3);
  if (debug_stop == 3)
    return;
  // ---- finger / hand / deepen logic, wave-parallel (finger_hand.cpp:20-115,173-233) ----
  // lane = region key for the prefix / suffix ORs, lane = finger slot for the finger masks, lane = depth for the
  // back-of-hand collision test.  All integer logic on the (region, depth) bit table.
  const int R = 2 * G.n_thr + 1;
  for (int oo = 0; oo < kOW; oo++)
  {
    const int o = wave + NW * oo;
    if (ori[o].rejected)
    {
      // put out by a later tile after an earlier one had been classified: its table words go back to zero like those the
      // finger logic reads (they are image planes from pass B on)
      if (((classified & ~live) >> o) & 1u)
        for (int k = lane; k < kRmCopies * 44; k += 64)
          rmc[(k / 44) * kRmCopyStride + o * kRmOriStride + (k % 44)] = 0u;
      continue;
    }
    const double ymin = ori[o].ymin;  // (over all tiles, from the screening)
    if (lane < 44)  // this orientation's table: the OR of its copies (only this wave wrote them), which are zeroed again
    {
      unsigned m = 0;
      for (int c = 0; c < kRmCopies; c++)
      {
        m |= rmc[c * kRmCopyStride + o * kRmOriStride + lane];
        rmc[c * kRmCopyStride + o * kRmOriStride + lane] = 0u;
      }
      regmask[o][lane] = m;
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    auto half = [&](int key) -> unsigned { return (regmask[o][key >> 1] >> ((key & 1) * 16)) & 0xffffu; };
    const unsigned v0 = lane < R ? half(lane) : 0u;
    const unsigned v1 = (64 + lane) < R ? half(64 + lane) : 0u;
    // inclusive prefix / suffix ORs over the 128 keys: six v_or_b32_dpp per scan (row shifts, then row broadcasts);
    // the suffix scans run on lane-reversed data (one ds_bpermute there, one back)
    const int rev = (63 - lane) << 2;
    unsigned p0 = wave_prefix_or(v0);
    unsigned p1 = wave_prefix_or(v1) | (unsigned) __builtin_amdgcn_readlane((int) p0, 63);
    unsigned s1 = wave_prefix_or((unsigned) __builtin_amdgcn_ds_bpermute(rev, (int) v1));
    unsigned s0 = wave_prefix_or((unsigned) __builtin_amdgcn_ds_bpermute(rev, (int) v0)) |
                  (unsigned) __builtin_amdgcn_readlane((int) s1, 63);  // lane 63 of the reversed scan = OR of all of v1
    s1 = (unsigned) __builtin_amdgcn_ds_bpermute(rev, (int) s1);
    s0 = (unsigned) __builtin_amdgcn_ds_bpermute(rev, (int) s0);
    pre_s[wave][lane] = p0;
    suf_s[wave][lane] = s0;
    if (lane < 24)
    {
      pre_s[wave][64 + lane] = p1;
      suf_s[wave][64 + lane] = s1;
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    unsigned Fm = 0;
    if (lane < 20)
    {
      const int lo = G.lo_idx[lane], hi = G.hi_idx[lane];
      unsigned gap = 0;
      for (int key = 2 * lo + 2; key <= 2 * hi; key++)  // points with fs_i < x < fs_i + w
        gap |= half(key);
      unsigned side;
      if (lane <= 10)  // the reference's `i <= m / 2` (finger_hand.cpp:72): slot 10 uses the right-side rule
        side = (2 * hi + 2 < R) ? suf_s[wave][2 * hi + 2] : 0u;  // x > fs_i + w
      else
        side = pre_s[wave][2 * lo];  // x < fs_i
      const int mingap = gap ? (__ffs(gap) - 1) : 32, minside = side ? (__ffs(side) - 1) : 32;
      // finger_i at depth k  <=>  no gap point with y < d_k  and  some side point with y < d_k
      Fm = lowmask(min(mingap, K)) & ~lowmask(minside);
    }
    const bool collide = lane < K && (ymin < G.depths[lane]) && (ymin < G.backs[lane]);  // finger_hand.cpp:29-42
    const unsigned Cm = (unsigned) __ballot(collide);
    const unsigned Fpair = __shfl(Fm, (lane + 10) & 63);
    const unsigned H = (lane < 10) ? (Fm & Fpair & ~Cm & lowmask(K)) : 0u;  // hand_j over depths
    const unsigned hand0 = (unsigned) __ballot((H & 1u) != 0);
    int e = -1, last = 0;
    if (hand0)
    {
      const int cnt = __popc(hand0);
      int pick = (cnt + 1) / 2 - 1;  // ceil(cnt / 2.0) - 1 (finger_hand.cpp:190)
      unsigned hm = hand0;
      while (pick-- > 0)
        hm &= hm - 1u;
      e = __ffs(hm) - 1;
      const unsigned He = __shfl(H, e);
      last = __ffs(~(He >> 1)) - 1;  // consecutive depths 1.. at which hand_e still exists (finger_hand.cpp:204-225)
    }
    if (lane == 0)
    {
      ori[o].has_hand = (e >= 0) ? 1 : 0;
      ori[o].e = e;
      ori[o].last = last;
      if (e >= 0)
        any_hand = 1;
    }
  }
  __syncthreads();
  AGH_STAMP(4);
  if (debug_stop == 4)
    return;
  // ---- pass B: grasp parameters, box, antipodal counts, image (rotating_hand.cpp:111-170) ----
  const int cam_s = cam_source ? (cam_source[samples[s]] & 1) : 0;  // hands_cam_source(i) = pts_cam_source(indices[i])
  // Few orientations have a hand (815 of C2's 9354), and a work-group that has one used to leave its points to ONE wave while
  // three idled -- on exactly the work-groups that run longest.  Every wave now takes a quarter of the tile for each such
  // orientation; its extrema and counts meet in LDS (PassBPart, on top of the finger phase's suffix table, dead by now) and the
  // orientation's own wave combines them for the record.  ymax: the max of y over ALL cropped points, for grasp_bottom
  // (finger_hand.cpp:135) -- only an orientation with a hand needs it, so it is taken here and not in pass A.
  struct PassBPart
  {
    double wmin, wmax, ymax;
    int nbox, numl, numr, pad;
  };
  static_assert(sizeof(PassBPart) * 8 * NW <= sizeof(suf_s), "the partial results live on the suffix table");
  PassBPart* const part = reinterpret_cast<PassBPart*>(&suf_s[0][0]);  // [wave][orientation]
  if (lane < 8)
    part[wave * 8 + lane] = PassBPart{ 100000.0, -100000.0, -INFINITY, 0, 0, 0, 0 };  // (a wave reads back only what it wrote)
  // surface point and image orientation of orientation o (rotating_hand.cpp:118-121, learning.cpp:382-383)
  auto ori_consts = [&](int o, double (&surf)[3], bool& pos_x, double& hor_pos) {
    const OriState& O = ori[o];
    const int e = O.e < 0 ? 0 : O.e;
    hor_pos = (G.hand_outer_diameter / 2.0) + (G.fs[e] / 1);  // finger_hand.cpp:127-132, one-hot hand_
    double s2c[3];
    for (int i = 0; i < 3; i++)
    {
      surf[i] = (O.T[i][0] * hor_pos + O.T[i][1] * O.ymin) + O.T[i][2] * 0.0;
      s2c[i] = (surf[i] + F.sample[i]) - G.cam_origin[cam_s][i];
    }
    pos_x = ((O.binormal[0] * s2c[0] + O.binormal[1] * s2c[1]) + O.binormal[2] * s2c[2]) > 0;
  };
  // one tile's share of pass B for every orientation with a hand
  auto pass_b_tile = [&](int nc) {
    for (int o = 0; o < 8; o++)
    {
      const OriState& O = ori[o];
      if (O.rejected || !O.has_hand)
        continue;
      const int e = O.e, last = O.last;
      const double cs = O.cs, ms = -1.0 * O.sn, sn = O.sn;
      const double left = G.fs[e], right = G.fs[10 + e];
      const double box_y = G.boxy[last];
      const double bite = G.init_bite;
      double surf[3], hor_pos_;
      bool pos_x;
      ori_consts(o, surf, pos_x, hor_pos_);
      const double sfx = surf[0], sfy = surf[1];
      double wmin = 100000.0, wmax = -100000.0, ymax = -INFINITY;
      int nbox = 0, numl = 0, numr = 0;
      for (int t0 = tid; t0 < nc; t0 += 4 * NT)
      {
        // four independent points per lane in stages (straight-line code: the two exactly-rounded divisions of each point
        // overlap with those of the others; the image update is the only predicated part)
        const bool full = (t0 - tid) + 4 * NT <= nc;
        double xr[4], yr[4];
        bool act[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
          const int t = t0 + NT * u;
          act[u] = full || t < nc;
          const double2 p = pts[act[u] ? t : 0];
          xr[u] = cs * p.x + ms * p.y;
          yr[u] = sn * p.x + cs * p.y;
          ymax = max_f64_raw(ymax, yr[u]);  // (an inactive lane holds point 0 once more: idempotent)
        }
        int bit[4];
        bool inbox[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
          const bool inw = act[u] & (yr[u] < bite) & (xr[u] > left) & (xr[u] < right);  // finger_hand.cpp:158-167
          wmin = min_f64_raw(wmin, inw ? xr[u] : 100000.0);   // (the sentinels are the initial values: no-ops)
          wmax = max_f64_raw(wmax, inw ? xr[u] : -100000.0);
          const double bx = xr[u] - sfx;  // rotating_hand.cpp:138 (world-frame offset, as in the reference)
          const double by = yr[u] - sfy;
          const double hx = pos_x ? (bx - (-0.05)) / img_cell : (-bx - (-0.05)) / img_cell;  // learning.cpp:330-333
          const double vy = (by - 0.0) / img_cell;
          inbox[u] = act[u] & (yr[u] < box_y);  // rotating_hand.cpp:125-130
          int hc = (int) floor(hx), vc = (int) floor(vy);
          hc = min(99, max(0, hc));
          vc = min(79, max(0, vc));
          bit[u] = (79 - vc) * 100 + hc;
          nbox += inbox[u] ? 1 : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
          if (inbox[u])
          {
            if (TRAIN)  // pid = (index << 1) | camera
              atomicOr(&img[o + 8 * (int) (pid[t0 + NT * u] & 1u)][bit[u] >> 5], 1u << (bit[u] & 31));
            else
              atomicOr(&img[o][bit[u] >> 5], 1u << (bit[u] & 31));
            if (NORMALS)
            {
              const double* nn = normals + 3 * (int64_t) (pid[t0 + NT * u] >> 1);
              const double n0 = nn[0], n1 = nn[1], n2 = nn[2];
              const double nxp = (fr[0][0] * n0 + fr[1][0] * n1) + fr[2][0] * n2;  // frame_^T * normals (33)
              const double nyp = (fr[0][1] * n0 + fr[1][1] * n1) + fr[2][1] * n2;
              const double nxr = cs * nxp + ms * nyp;  // rot * normals_ (92)
              numl += (-1.0 * nxr > G.cos_antipodal) ? 1 : 0;  // antipodal.cpp:28,38
              numr += (nxr > G.cos_antipodal) ? 1 : 0;
            }
          }
        }
      }
      // this wave's share of the tile, folded into its slot (the tiles of a neighbourhood come one after the other)
      wmin = wave_min_f64(wmin);
      wmax = wave_max_f64(wmax);
      ymax = wave_max_f64(ymax);
      nbox = wave_sum_i32(nbox);
      if (NORMALS)
      {
        numl = wave_sum_i32(numl);
        numr = wave_sum_i32(numr);
      }
      if (lane == 0)
      {
        PassBPart& P = part[wave * 8 + o];
        P.wmin = min_f64_raw(P.wmin, wmin);
        P.wmax = max_f64_raw(P.wmax, wmax);
        P.ymax = max_f64_raw(P.ymax, ymax);
        P.nbox += nbox;
        P.numl += numl;
        P.numr += numr;
      }
    }
  };
  if (any_hand)
  {
    // A neighbourhood that fitted one tile is still in LDS; one that streamed through the tile in pass A is gathered again.
    if (ntiles == 1)
      pass_b_tile(nc);
    else
    {
      next_tile();
      gather_reset();
      for (;;)
      {
        bool all_done = true;
        nc = gather_tile(all_done);
        pass_b_tile(nc);
        if (all_done)
          break;
        next_tile();
      }
    }
  }
  __syncthreads();  // every wave's share of pass B (partials, image bits) is in LDS
  AGH_STAMP(5);
  if (debug_stop == 5)
    return;
  // ---- results ----
  for (int oo = 0; oo < kOW; oo++)
  {
    const int o = wave + NW * oo;
    const OriState& O = ori[o];
    if (O.rejected || !O.has_hand)
      continue;  // (dead slots are not written: vmask tells the concatenation which are live -- 2.4 MB of stores less at C2)
    agh_hypothesis h;
    memset(&h, 0, sizeof(h));
    h.sample = s;
    h.orientation = o;
    double wmin = 100000.0, wmax = -100000.0, ymax = -INFINITY;
    int nbox = 0, numl = 0, numr = 0;
    for (int w = 0; w < NW; w++)
    {
      const PassBPart P = part[w * 8 + o];
      wmin = min_f64_raw(wmin, P.wmin);
      wmax = max_f64_raw(wmax, P.wmax);
      ymax = max_f64_raw(ymax, P.ymax);
      nbox += P.nbox;
      numl += P.numl;
      numr += P.numr;
    }
    double surf[3], hor_pos;
    bool pos_x_;
    ori_consts(o, surf, pos_x_, hor_pos);
    for (int i = 0; i < 3; i++)
    {
      h.axis[i] = F.axis[i];
      h.approach[i] = O.approach[i];
      h.binormal[i] = O.binormal[i];
      h.bottom[i] = ((O.T[i][0] * hor_pos + O.T[i][1] * ymax) + O.T[i][2] * 0.0) + F.sample[i];  // rotating_hand.cpp:120-122,153-154
      h.surface[i] = surf[i] + F.sample[i];
    }
    h.width = wmax - wmin;
    h.cam_source = cam_s;
    h.n_in_box = nbox;
    const bool full = numl > 6 && numr > 6;
    const bool half_a = numl > 6 || numr > 6;
    h.half_antipodal = (half_a || full) ? 1 : 0;
    h.full_antipodal = full ? 1 : 0;
    h.valid = 1;
    h.finger_index = O.e;
    h.depth_index = O.last;
    // the record leaves as ONE 160-byte store of ten lanes (staged through the wave's prefix table, dead since the finger
    // phase): ten 16-byte stores of one lane were ten partial-line write transactions (WRITE_SIZE 19.8 MB per launch at C2
    // against 3.4 MB of payload, profiles/r03_pmc_traffic.json)
    static_assert(sizeof(agh_hypothesis) <= sizeof(pre_s[0]) && sizeof(pre_s[0]) % 16 == 0, "record staging");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0)
      *reinterpret_cast<agh_hypothesis*>(&pre_s[wave][0]) = h;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    if (lane < 10)
      reinterpret_cast<uint4*>(slots + ((int64_t) s * 8 + o))[lane] = reinterpret_cast<const uint4*>(&pre_s[wave][0])[lane];
    for (int k = lane; k < kImageWords; k += 64)
    {
      if (TRAIN)  // ins.pts of cam = -1 is the union of the two cameras' points
      {
        const unsigned a = img[o][k], b = img[o + 8][k];
        images[((int64_t) s * 8 + o) * kImageWords + k] = a | b;
        images_cam[(((int64_t) s * 8 + o) * 2 + 0) * kImageWords + k] = a;
        images_cam[(((int64_t) s * 8 + o) * 2 + 1) * kImageWords + k] = b;
      }
      else
        images[((int64_t) s * 8 + o) * kImageWords + k] = img[o][k];
    }
  }
  if (debug_stop == 6)
    return;
  if (tid == 0)
  {
    status[s] = kStatusOk;
    unsigned m = 0;  // orientations that produced a hypothesis: what the concatenation kernel scans
    for (int o = 0; o < 8; o++)
      m |= (!ori[o].rejected && ori[o].has_hand) ? (1u << o) : 0u;
    vmask[s] = (uint8_t) m;
    if (dbg)
    {
      dbg[(int64_t) s * 8 + 6] = wall_clock64();
      dbg[(int64_t) s * 8 + 7] = ((long long) ncrop_all << 32) | (unsigned) (ntiles << 24) | (unsigned) (total & 0xffffff);
    }
  }
#undef AGH_STAMP
}

// ---------------------------------------------------------------------------------------------------------------
// K4: concatenate the per-sample lists in sample order (hand_search.cpp:194-200): scan of the slot valid flags.
// ---------------------------------------------------------------------------------------------------------------
// (slot i is live iff bit (i & 7) of vmask[i >> 3] is set: the sweep does not write the records of dead slots)
__global__ __launch_bounds__(256) void k_compact_sums(const uint8_t* __restrict__ vmask, int n,
  int* __restrict__ block_sums)
{
  __shared__ int ws[4];
  const int i0 = blockIdx.x * 1024 + threadIdx.x * 4;
  int sum = 0;
  for (int k = 0; k < 4; k++)
    if (i0 + k < n)
      sum += (vmask[(i0 + k) >> 3] >> ((i0 + k) & 7)) & 1;
  sum = wave_sum_i32(sum);
  if ((threadIdx.x & 63) == 0)
    ws[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0)
    block_sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(256) void k_compact_top(int* __restrict__ block_sums, int nb, int64_t* __restrict__ n_out,
  const int32_t* __restrict__ flags_in, int64_t* __restrict__ hdr_flags_out, int hdr_big, HostMirror mir)
{
  // nb <= 4096: serial chunks of 256 with a wave scan
  __shared__ int ws[4];
  int carry = 0;
  for (int b0 = 0; b0 < nb; b0 += 256)
  {
    const int i = b0 + threadIdx.x;
    const int v = i < nb ? block_sums[i] : 0;
    int inc = v;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1)
    {
      const int t = __shfl_up(inc, o);
      if (lane >= o)
        inc += t;
    }
    if (lane == 63)
      ws[w] = inc;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < w; k++)
      base += ws[k];
    const int tot = ws[0] + ws[1] + ws[2] + ws[3];
    if (i < nb)
      block_sums[i] = carry + base + inc - v;
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0)
  {
    *n_out = carry;
    if (hdr_flags_out)  // sharded search: see k_compact_offsets
      *hdr_flags_out = shard_header_word(flags_in[0], hdr_big);
    if (mir.hdr)
    {
      mir.hdr[0] = carry;
      mir.hdr[1] = flags_in[0];
    }
  }
}

__global__ __launch_bounds__(256) void k_compact_write(const agh_hypothesis* __restrict__ slots,
  const uint8_t* __restrict__ vmask, int n, const int* __restrict__ block_sums, agh_hypothesis* __restrict__ out, int64_t cap, int32_t* __restrict__ slot_of_hyp,
  int32_t* __restrict__ flags, int32_t epoch, HostMirror mir)
{
  __shared__ int ws[4];
  const int i0 = blockIdx.x * 1024 + threadIdx.x * 4;
  int v[4], sum = 0;
  for (int k = 0; k < 4; k++)
  {
    v[k] = (i0 + k < n && ((vmask[(i0 + k) >> 3] >> ((i0 + k) & 7)) & 1)) ? 1 : 0;
    sum += v[k];
  }
  int inc = sum;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int o = 1; o < 64; o <<= 1)
  {
    const int t = __shfl_up(inc, o);
    if (lane >= o)
      inc += t;
  }
  if (lane == 63)
    ws[w] = inc;
  __syncthreads();
  int pos = block_sums[blockIdx.x] + inc - sum;
  for (int k = 0; k < w; k++)
    pos += ws[k];
  for (int k = 0; k < 4; k++)
    if (v[k])
    {
      if (pos < cap)
      {
        out[pos] = slots[i0 + k];
        out[pos].epoch = epoch;
        slot_of_hyp[pos] = i0 + k;
        if (mir.rec && pos < mir.cap)
        {
          mir.rec[pos] = slots[i0 + k];
          mir.rec[pos].epoch = epoch;
        }
      }
      else
        atomicOr(&flags[0], 2);
      pos++;
    }
}

// K4 in two launches for S <= 65536: the sweep leaves an 8-bit mask of the orientations with a hypothesis per sample;
// one work-group scans the S popcounts (output offset of every sample, total count), then 10 threads copy each record
// (16 bytes per thread, coalesced).
__global__ __launch_bounds__(1024) void k_compact_offsets(const uint8_t* __restrict__ vmask, int S, int* __restrict__ offs,
  int64_t* __restrict__ n_out, const int32_t* __restrict__ flags_in, int64_t* __restrict__ hdr_flags_out, int hdr_big, HostMirror mir)
{
  // thread t owns the samples [t per, (t + 1) per), per a multiple of 16: its masks arrive as 16-byte loads, all in flight
  // together (one byte per load and iteration made this kernel 15 us for the 16 000 samples of a batch; the buffer is
  // allocated 16 bytes longer than its samples, and masks at or beyond S count as zero)
  __shared__ int ws[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int per = (((S + 1023) / 1024) + 15) & ~15;  // <= 64 for S <= 65536
  const int s0 = tid * per;
  uint4 m[4];
  int cnt = 0;
#pragma unroll
  for (int c = 0; c < 4; c++)
  {
    const int base = s0 + 16 * c;
    m[c] = (16 * c < per && base < S) ? *reinterpret_cast<const uint4*>(vmask + base) : make_uint4(0u, 0u, 0u, 0u);
    unsigned* q = reinterpret_cast<unsigned*>(&m[c]);
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      const int left = S - (base + 4 * k);  // valid bytes of this word
      if (left < 4)
        q[k] &= left <= 0 ? 0u : (0xffffffffu >> (8 * (4 - left)));
      cnt += __popc(q[k]);
    }
  }
  int inc = cnt;
  for (int o = 1; o < 64; o <<= 1)
  {
    const int t = __shfl_up(inc, o);
    if (lane >= o)
      inc += t;
  }
  if (lane == 63)
    ws[w] = inc;
  __syncthreads();
  int pos = inc - cnt, total = 0;
  for (int k = 0; k < 16; k++)
  {
    pos += k < w ? ws[k] : 0;
    total += ws[k];
  }
  if (tid == 0)
  {
    *n_out = total;
    if (hdr_flags_out)  // sharded search: this rank's findings (capacity class, bad index) travel in its segment header, so
      *hdr_flags_out = shard_header_word(flags_in[0], hdr_big);  // that every rank learns them from the same all-gather
    if (mir.hdr)
    {
      mir.hdr[0] = total;
      mir.hdr[1] = flags_in[0];
    }
  }
#pragma unroll
  for (int c = 0; c < 4; c++)
  {
    const int base = s0 + 16 * c;
    if (16 * c < per && base < S)
    {
      const unsigned* q = reinterpret_cast<const unsigned*>(&m[c]);
#pragma unroll
      for (int k = 0; k < 16; k++)
      {
        if (base + k < S)
          offs[base + k] = pos;
        pos += __popc((q[k >> 2] >> (8 * (k & 3))) & 0xffu);
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_compact_copy(const agh_hypothesis* __restrict__ slots,
  const uint8_t* __restrict__ vmask, const int* __restrict__ offs, int n_slots, agh_hypothesis* __restrict__ out,
  int64_t cap, int32_t* __restrict__ slot_of_hyp, int32_t* __restrict__ flags, int32_t epoch, HostMirror mir)
{
  static_assert(sizeof(agh_hypothesis) == 160, "10 x 16 bytes per record");
  static_assert(offsetof(agh_hypothesis, epoch) == 156, "the stamp is the last word of the record");
  const int slot = blockIdx.x * 25 + threadIdx.x / 10, part = threadIdx.x % 10;
  if (threadIdx.x >= 250 || slot >= n_slots)
    return;
  const int s = slot >> 3, o = slot & 7;
  const unsigned m = vmask[s];
  if (!((m >> o) & 1u))
    return;
  const int64_t pos = offs[s] + __popc(m & ((1u << o) - 1u));
  if (pos >= cap)
  {
    if (part == 0)
      atomicOr(&flags[0], 2);
    return;
  }
  uint4 v = reinterpret_cast<const uint4*>(slots + slot)[part];
  if (part == 9)
    v.w = (unsigned) epoch;
  reinterpret_cast<uint4*>(out + pos)[part] = v;
  if (mir.rec && pos < mir.cap)
    reinterpret_cast<uint4*>(mir.rec + pos)[part] = v;
  if (part == 0)
    slot_of_hyp[pos] = slot;
}

// K4 in ONE launch for S <= 4096: every copy work-group (25 slots, i.e. at most five samples) sums the masks in front of its
// first sample itself -- at most 4 KB out of L2, one 16-byte load per thread -- instead of waiting for an offsets kernel
// (a launch costs ~4 us on this part, the redundant sums ~1).  The last work-group also knows the total.
__global__ __launch_bounds__(256) void k_compact_fused(const agh_hypothesis* __restrict__ slots,
  const uint8_t* __restrict__ vmask, int S, int n_slots, agh_hypothesis* __restrict__ out, int64_t cap,
  int32_t* __restrict__ slot_of_hyp, int32_t* __restrict__ flags, int32_t epoch, int64_t* __restrict__ n_out,
  int64_t* __restrict__ hdr_flags_out, int hdr_big, HostMirror mir)
{
  __shared__ int wsum[2][4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int s_first = (blockIdx.x * 25) >> 3;
  const bool last_group = blockIdx.x == gridDim.x - 1;
  auto popc_below = [&](const uint4& m, int base, int limit) -> int {  // masks of the samples base .. base + 15 below `limit`
    const unsigned q[4] = { m.x, m.y, m.z, m.w };
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      const int left = limit - (base + 4 * k);
      const unsigned keep = left >= 4 ? 0xffffffffu : (left <= 0 ? 0u : (0xffffffffu >> (8 * (4 - left))));
      c += __popc(q[k] & keep);
    }
    return c;
  };
  int before = 0, all = 0;
  const int s_scan = last_group ? S : s_first;
  for (int b = tid * 16; b < s_scan; b += 256 * 16)
  {
    const uint4 m = *reinterpret_cast<const uint4*>(vmask + b);  // (the buffer is 16 bytes longer than its samples)
    before += popc_below(m, b, s_first);
    if (last_group)
      all += popc_below(m, b, S);
  }
  for (int o = 32; o > 0; o >>= 1)
  {
    before += __shfl_xor(before, o);
    all += __shfl_xor(all, o);
  }
  if (lane == 0)
  {
    wsum[0][w] = before;
    wsum[1][w] = all;
  }
  __syncthreads();
  before = (wsum[0][0] + wsum[0][1]) + (wsum[0][2] + wsum[0][3]);
  if (last_group && tid == 0)
  {
    *n_out = (wsum[1][0] + wsum[1][1]) + (wsum[1][2] + wsum[1][3]);
    if (hdr_flags_out)  // sharded search: see k_compact_offsets
      *hdr_flags_out = shard_header_word(flags[0], hdr_big);
    if (mir.hdr)  // (bit 1 of the word, "list longer than cap", may still be raised by other work-groups of this launch:
    {             // the host derives it from the count)
      mir.hdr[0] = (wsum[1][0] + wsum[1][1]) + (wsum[1][2] + wsum[1][3]);
      mir.hdr[1] = flags[0];
    }
  }
  const int slot = blockIdx.x * 25 + tid / 10, part = tid % 10;
  if (tid >= 250 || slot >= n_slots)
    return;
  const int sm = slot >> 3, o = slot & 7;
  const unsigned m = vmask[sm];
  if (!((m >> o) & 1u))
    return;
  int64_t pos = before + __popc(m & ((1u << o) - 1u));
  for (int q = s_first; q < sm; q++)  // (at most four samples)
    pos += __popc((unsigned) vmask[q]);
  if (pos >= cap)
  {
    if (part == 0)
      atomicOr(&flags[0], 2);
    return;
  }
  uint4 v = reinterpret_cast<const uint4*>(slots + slot)[part];
  if (part == 9)
    v.w = (unsigned) epoch;
  reinterpret_cast<uint4*>(out + pos)[part] = v;
  if (mir.rec && pos < mir.cap)
    reinterpret_cast<uint4*>(mir.rec + pos)[part] = v;
  if (part == 0)
    slot_of_hyp[pos] = slot;
}

// The number of points radiusSearch(sample, nn_radius_hands) returns (hand_search.cpp:147), for agh_get_neighbor_counts: a
// lazy getter.  The sweep itself only visits the slab of the ball the hand can occupy, so it never sees this number.
__global__ __launch_bounds__(256) void k_ball_count(GridView gv, const agh_frame* __restrict__ frames,
  const int32_t* __restrict__ scloud, int S, float r2f, double rpad, int32_t* __restrict__ nh)
{
  __shared__ RowTable rt;
  __shared__ int cnt;
  const int s = blockIdx.x, tid = threadIdx.x;
  const agh_frame F = frames[s];
  if (!F.valid)
  {
    if (tid == 0)
      nh[s] = 0;
    return;
  }
  if (tid == 0)
    cnt = 0;
  const float sx = (float) F.sample[0], sy = (float) F.sample[1], sz = (float) F.sample[2];
  gv = grid_of_cloud(gv, scloud[s]);
  build_rows(gv, sx, sy, sz, rpad, rt);
  int n = 0;
  if (!rt.bad)
    for (int j = tid; j < rt.total; j += 256)
    {
      const float4 p = gv.sorted[row_lookup(rt, j)];
      n += flann_d2(sx, sy, sz, p.x, p.y, p.z) < r2f ? 1 : 0;
    }
  n = wave_sum_i32(n);
  if ((tid & 63) == 0 && n)
    atomicAdd(&cnt, n);
  __syncthreads();
  if (tid == 0)
    nh[s] = cnt;
}

int ball_counts(Ctx* c, int64_t S, hipStream_t st)
{
  if (S == 0)
    return AGH_OK;
  GridView gv{ c->d_desc, c->d_cell_start, c->d_sorted, c->d_cloud_off, c->n_clouds };
  const double radius = c->p.nn_radius_hands;
  hipLaunchKernelGGL(k_ball_count, dim3((int) S), dim3(256), 0, st, gv, (const agh_frame*) c->d_frames,
    (const int32_t*) c->d_scloud, (int) S, static_cast<float>(radius * radius), radius * 1.0001 + 1e-6, c->d_nh);
  return hipGetLastError() == hipSuccess ? AGH_OK : AGH_ERR_HIP;
}

int hand_sweep(Ctx* c, const int32_t* d_samples, int64_t S, bool use_normals, hipStream_t st)
{
  if (S == 0)
    return AGH_OK;
  GridView gv{ c->d_desc, c->d_cell_start, c->d_sorted, c->d_cloud_off, c->n_clouds };
  const double radius = c->p.nn_radius_hands;
  const float r2f = static_cast<float>(radius * radius);
  const double rpad = radius * 1.0001 + 1e-6;
  const double img_cell = (0.05 - (-0.05)) / (double) 100;  // learning.cpp:324
  const int Si = (int) S;
  const HandGeom* dg = c->d_geom;
  const double* nrm = use_normals ? c->d_normals : nullptr;
  // the block-wise order K1b's sorter made for exactly this launch (same sample list, same count), else sample order
  const int* order = (c->order_sweep_s == Si && c->order_sweep_samples == d_samples) ? c->d_order_sweep : nullptr;
#ifdef AGH_DEBUG_HOOKS
  {
    static int* dbg_order = nullptr;
    static int dbg_n = -1;
    if (dbg_n < 0)
    {
      dbg_n = 0;
      if (const char* f = getenv("AGH_DEBUG_SWEEP_ORDER"))
        if (FILE* fp = fopen(f, "rb"))
        {
          std::vector<int> h(1 << 20);
          dbg_n = (int) fread(h.data(), 4, h.size(), fp);
          fclose(fp);
          hipMalloc(&dbg_order, sizeof(int) * (size_t) dbg_n);
          hipMemcpy(dbg_order, h.data(), sizeof(int) * (size_t) dbg_n, hipMemcpyHostToDevice);
        }
    }
    if (dbg_n == Si)
      order = dbg_order;
  }
#endif
  long long* sweep_dbg = c->d_dbg;
#ifdef AGH_DEBUG_HOOKS
  if (getenv("AGH_DEBUG_CLOCKS_KERNEL"))
    sweep_dbg = nullptr;  // another kernel's timestamps are wanted (k_taubin_frame)
#endif
  const bool few = c->geom.x_probes <= 2 && c->geom.y_probes <= 1;
  // (profile levels 2 and 3: the launch carries its own start and stop events)
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  const bool timed = timing_launch_events(c, "hand_sweep", &ev_start, &ev_stop);
#define AGH_SWEEP_ARGS                                                                                                  \
  gv, dg, (const agh_frame*) c->d_frames, d_samples, (const int32_t*) c->d_cam, Si, r2f, rpad, nrm, img_cell, c->d_status, \
    c->d_slots, c->d_images, c->debug_stop_sweep, sweep_dbg, order, c->d_vmask, c->d_images_cam
#ifndef AGH_SWEEP_NT
#define AGH_SWEEP_NT 256  // threads of the online variant's work-groups (512: eight waves, two per CU, one 3712-point tile -- measured, not faster)
#endif
#define AGH_LAUNCH_SWEEP_NT(N, PX, PY, NT) AGH_LAUNCH_SWEEP_W(N, PX, PY, NT, 0)
#define AGH_LAUNCH_SWEEP_W(N, PX, PY, NT, W4)                                                                                    \
  do                                                                                                                             \
  {                                                                                                                              \
    if (timed)                                                                                                                   \
      hipExtLaunchKernelGGL((k_hand_sweep<N, PX, PY, NT, W4>), dim3(Si), dim3(NT), 0, st, ev_start, ev_stop, 0, AGH_SWEEP_ARGS); \
    else                                                                                                                         \
      hipLaunchKernelGGL((k_hand_sweep<N, PX, PY, NT, W4>), dim3(Si), dim3(NT), 0, st, AGH_SWEEP_ARGS);                           \
  } while (0)
  // (the variants with normals need more than the 128 registers that eight-wave work-groups, two per CU, leave a lane)
#define AGH_LAUNCH_SWEEP(N, PX, PY) AGH_LAUNCH_SWEEP_NT(N, PX, PY, ((N) == 0 ? AGH_SWEEP_NT : 256))
  const bool train = nrm && c->training_images && c->d_images_cam;
  if (train)
    AGH_LAUNCH_SWEEP(2, kLutProbe, kLutProbe);  // (offline path: one instantiation covers every geometry)
  else if (nrm && few)
    AGH_LAUNCH_SWEEP(1, 2, 1);
  else if (nrm)
    AGH_LAUNCH_SWEEP(1, kLutProbe, kLutProbe);
  else if (few && Si > kSweepWg4MinSamples && AGH_SWEEP_NT == 256)
    AGH_LAUNCH_SWEEP_W(0, 2, 1, 256, 1);  // many rounds of work-groups: four per CU (see WG4)
  else if (few)
    AGH_LAUNCH_SWEEP(0, 2, 1);
  else
    AGH_LAUNCH_SWEEP_NT(0, kLutProbe, kLutProbe, 256);  // (unusual hand geometries: the general probes spill at 128 registers)
  c->last_has_cam_images = train;
#undef AGH_LAUNCH_SWEEP
#undef AGH_LAUNCH_SWEEP_NT
#undef AGH_LAUNCH_SWEEP_W
#undef AGH_SWEEP_ARGS
  timing_mark(c, "hand_sweep", st);
  return hipGetLastError() == hipSuccess ? AGH_OK : AGH_ERR_HIP;
}

int compact_hypotheses(Ctx* c, int64_t S, agh_hypothesis* d_out, int64_t cap, int64_t* d_nout, hipStream_t st,
  int64_t* d_hdr_flags)
{
  const int n = (int) (S * 8);
  const int nb = (n + 1023) / 1024;
  if (n == 0)
  {
    hipMemsetAsync(d_nout, 0, sizeof(int64_t), st);
    return AGH_OK;
  }
  const HostMirror mir = c->mirror;
  if (S <= 4096)
    hipLaunchKernelGGL(k_compact_fused, dim3((n + 24) / 25), dim3(256), 0, st, (const agh_hypothesis*) c->d_slots,
      (const uint8_t*) c->d_vmask, (int) S, n, d_out, cap, c->d_slot_index, c->d_flags, c->epoch, d_nout, d_hdr_flags,
      class_level(c), mir);
  else if (S <= 65536)
  {
    hipLaunchKernelGGL(k_compact_offsets, dim3(1), dim3(1024), 0, st, (const uint8_t*) c->d_vmask, (int) S, c->d_scan_tmp,
      d_nout, (const int32_t*) c->d_flags, d_hdr_flags, class_level(c), mir);
    hipLaunchKernelGGL(k_compact_copy, dim3((n + 24) / 25), dim3(256), 0, st, (const agh_hypothesis*) c->d_slots,
      (const uint8_t*) c->d_vmask, (const int*) c->d_scan_tmp, n, d_out, cap, c->d_slot_index, c->d_flags, c->epoch, mir);
  }
  else
  {
    hipLaunchKernelGGL(k_compact_sums, dim3(nb), dim3(256), 0, st, (const uint8_t*) c->d_vmask, n, c->d_scan_tmp);
    hipLaunchKernelGGL(k_compact_top, dim3(1), dim3(256), 0, st, c->d_scan_tmp, nb, d_nout, (const int32_t*) c->d_flags, d_hdr_flags,
      class_level(c), mir);
    hipLaunchKernelGGL(k_compact_write, dim3(nb), dim3(256), 0, st, c->d_slots, (const uint8_t*) c->d_vmask, n, c->d_scan_tmp, d_out, cap,
      c->d_slot_index, c->d_flags, c->epoch, mir);
  }
  timing_mark(c, "compact", st);
  return hipGetLastError() == hipSuccess ? AGH_OK : AGH_ERR_HIP;
}

}  // namespace agh
