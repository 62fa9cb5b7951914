// api.hip -- host side of the C ABI declared in include/agh.h (context, buffers, launch order, error reporting).
//
// Launch order of one agh_find_hands_device call = HandSearch::findHands (reference hand_search.cpp:4-62):
//   [all-points normals pass, r = 0.01, only if calculates_antipodal (17-26)]  K1 over every cloud point
//   findQuadrics over the samples (53)                                          K1a moments, K1b eigen, K1c frame
//   findHands over the quadrics (59)                                            K2 hand sweep
//   concatenation of the per-sample lists (194-200)                             K4 compaction
// Everything is enqueued on one HIP stream with no host synchronisation in between.
#include "agh_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace agh;


namespace
{
thread_local std::string g_create_error;
std::atomic<int32_t> g_epoch{ 0 };  // stamps of agh_find_hands* calls, unique across the contexts of a process
}  // namespace

namespace agh
{
int32_t next_epoch()
{
  int32_t e;
  do
    e = (int32_t) ((uint32_t) g_epoch.fetch_add(1) + 1u) & 0x7fffffff;
  while (e == 0);
  return e;
}
}  // namespace agh

namespace
{

#define HIPCHK(ctx, expr)                                                                             \
  do                                                                                                  \
  {                                                                                                   \
    hipError_t e__ = (expr);                                                                          \
    if (e__ != hipSuccess)                                                                            \
    {                                                                                                 \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                                \
      return AGH_ERR_HIP;                                                                             \
    }                                                                                                 \
  } while (0)

// glibc rand() (TYPE_3 additive feedback) -- the generator behind quadric.cpp:184 in a default Linux build.
struct GlibcRand
{
  std::vector<uint32_t> r;
  size_t pos;
  explicit GlibcRand(uint32_t seed)
  {
    r.resize(344);
    int32_t word = (int32_t) (seed == 0 ? 1u : seed);
    r[0] = (uint32_t) word;
    for (int i = 1; i < 31; i++)
    {
      const long hi = word / 127773, lo = word % 127773;
      long w = 16807 * lo - 2836 * hi;
      if (w < 0)
        w += 2147483647;
      word = (int32_t) w;
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
    const uint32_t o = r[pos] >> 1;
    pos++;
    return (int32_t) o;
  }
};

void build_geometry(const agh_params& p, HandGeom* g, std::string* err)
{
  std::memset(g, 0, sizeof(*g));
  g->finger_width = p.finger_width;
  g->hand_outer_diameter = p.hand_outer_diameter;
  g->hand_depth = p.hand_depth;
  g->hand_height = p.hand_height;
  g->init_bite = p.init_bite;
  std::memcpy(g->cam_origin, p.cam_origin, sizeof(g->cam_origin));
  // hand angles: LinSpaced(9, -pi, pi) without the last (rotating_hand.cpp:13-15); cos/sin from the host libm
  const double step = (M_PI - (-1.0 * M_PI)) / 8.0;
  for (int o = 0; o < 8; o++)
  {
    const double ang = -1.0 * M_PI + o * step;
    g->cos_a[o] = std::cos(ang);
    g->sin_a[o] = std::sin(ang);
  }
  // finger_spacing_ (finger_hand.cpp:8-15): fs_half = LinSpaced(10, 0, od - fw) = low + i * step
  const double low = 0.0, high = p.hand_outer_diameter - p.finger_width;
  const double fstep = (high - low) / 9.0;
  for (int i = 0; i < 10; i++)
  {
    const double h = low + i * fstep;
    g->fs[i] = (h - p.hand_outer_diameter) + p.finger_width;
    g->fs[10 + i] = h;
  }
  std::vector<double> t;
  for (int i = 0; i < 20; i++)
  {
    t.push_back(g->fs[i]);
    t.push_back(g->fs[i] + p.finger_width);  // formed as fs_i + fw at use (finger_hand.cpp:63,77)
  }
  std::sort(t.begin(), t.end());
  t.erase(std::unique(t.begin(), t.end()), t.end());
  g->n_thr = (int) t.size();
  for (int k = 0; k < 40; k++)
    g->thr[k] = k < g->n_thr ? t[k] : INFINITY;  // padded: fixed-length straight-line comparisons in the kernel
  for (int i = 0; i < 20; i++)
  {
    g->lo_idx[i] = (int) (std::lower_bound(t.begin(), t.end(), g->fs[i]) - t.begin());
    g->hi_idx[i] = (int) (std::lower_bound(t.begin(), t.end(), g->fs[i] + p.finger_width) - t.begin());
  }
  // bite depths: init_bite, then repeated += 0.005 while <= hand_depth (finger_hand.cpp:199-204)
  int k = 0;
  g->depths[k++] = p.init_bite;
  for (double d = p.init_bite + 0.005; d <= p.hand_depth; d += 0.005)
  {
    if (k >= 16)
    {
      *err = "more than 15 deepening steps (hand_depth - init_bite > 0.075) are not supported";
      break;
    }
    g->depths[k++] = d;
  }
  g->n_depths = k;
  for (int i = k; i < 16; i++)
    g->depths[i] = INFINITY;  // padded likewise
  for (int i = 0; i < k; i++)
  {
    g->backs[i] = -1.0 * (p.hand_depth - g->depths[i]);  // finger_hand.cpp:22
    g->boxy[i] = g->backs[i] + p.hand_depth;             // rotating_hand.cpp:127
  }
  g->cos_antipodal = std::cos(20 * M_PI / 180.0);  // antipodal.cpp:16 with thresh 20 (rotating_hand.cpp:162)
  // look-up tables (see HandGeom): the device evaluates exactly this cell formula
  auto build_lut = [&](const double* tab, int n, int ncell, double* lo_out, double* scale_out, unsigned char* lut,
                       int* probes) -> bool {
    *probes = 0;
    const double lo = tab[0], hi = tab[n - 1];
    const double scale = (hi > lo) ? (double) ncell / (hi - lo) : 0.0;
    *lo_out = lo;
    *scale_out = scale;
    std::vector<int> cell(n);
    for (int k = 0; k < n; k++)
      cell[k] = (int) std::fmin(std::fmax((tab[k] - lo) * scale, 0.0), (double) (ncell - 1));
    for (int c = 0; c < ncell; c++)
    {
      int before = 0, inside = 0;
      for (int k = 0; k < n; k++)
      {
        before += cell[k] < c ? 1 : 0;
        inside += cell[k] == c ? 1 : 0;
      }
      if (inside > kLutProbe)
        return false;
      *probes = std::max(*probes, inside);
      lut[c] = (unsigned char) before;
    }
    return true;
  };
  if (!build_lut(g->thr, g->n_thr, 1024, &g->xlut_lo, &g->xlut_scale, g->xlut, &g->x_probes) ||
      !build_lut(g->depths, g->n_depths, 64, &g->ylut_lo, &g->ylut_scale, g->ylut, &g->y_probes))
    *err = "hand geometry packs more than 4 finger-slot thresholds (or bite depths) into one look-up cell";
}

}  // namespace

namespace agh
{
int ensure_call_buffers(Ctx* c, int64_t S)
{
  if (c->training_images && c->images_cam_cap < std::max<int64_t>(std::max<int64_t>(S, c->s_cap), 1024))
  {
    const int64_t cap = std::max<int64_t>(std::max<int64_t>(S, c->s_cap), 1024);
    int rc;
    if ((rc = dev_alloc(c, &c->d_images_cam, (size_t) cap * 16 * kImageWords)))
      return rc;
    c->images_cam_cap = cap;
  }
  const int64_t want_stride = c->huge_classes ? kHugeCap : kBigCap;
  if (c->huge_classes && !c->d_huge_stage)
  {
    int rc;
    if ((rc = dev_alloc(c, &c->d_huge_stage, (size_t) kHugeEntries)) || (rc = dev_alloc(c, &c->d_huge_key, (size_t) kHugeEntries)) ||
        (rc = dev_alloc(c, &c->d_huge_sorted, (size_t) kHugeEntries)) || (rc = dev_alloc(c, &c->d_huge_normals, (size_t) kHugeEntries * 3)) ||
        (rc = dev_alloc(c, &c->d_huge_count, 2)))
      return rc;
  }
  if (S <= c->s_cap && c->nbr_stride == want_stride && (!c->huge_classes || c->d_huge_base))
    return AGH_OK;
  const int64_t cap = std::max<int64_t>(std::max<int64_t>(S, c->s_cap), 1024);
  int rc;
  if ((rc = dev_alloc(c, &c->d_samples, cap)))
    return rc;
  if ((rc = dev_alloc(c, &c->d_sums, cap * kSumStride)))
    return rc;
  if ((rc = dev_alloc(c, &c->d_nt, cap)) || (rc = dev_alloc(c, &c->d_ovf, cap + 16)))
    return rc;
  if ((rc = dev_alloc(c, &c->d_nh, cap)) || (rc = dev_alloc(c, &c->d_scloud, cap)))
    return rc;
  if ((rc = dev_alloc(c, &c->d_status, cap)) || (rc = dev_alloc(c, &c->d_weight, cap)) || (rc = dev_alloc(c, &c->d_order, cap)) || (rc = dev_alloc(c, &c->d_order_sweep, cap)) || (rc = dev_alloc(c, &c->d_vmask, cap + 16)))
    return rc;
  c->nbr_stride = want_stride;  // (the sorted neighbour list of a sample: 4096 entries, 6144 once the 6144 class is on)
  if ((rc = dev_alloc(c, &c->d_nbr, cap * c->nbr_stride)))
    return rc;
  if ((rc = dev_alloc(c, &c->d_eig, cap * 12)))
    return rc;
  if ((rc = dev_alloc(c, &c->d_frames, cap)))
    return rc;
  if ((rc = dev_alloc(c, &c->d_slots, cap * 8)))
    return rc;
  if ((rc = dev_alloc(c, &c->d_images, cap * 8 * kImageWords)))
    return rc;
  if ((rc = dev_alloc(c, &c->d_slot_index, cap * 8)))
    return rc;
  if ((rc = dev_alloc(c, &c->d_scan_tmp, cap + 1024)  /* block sums of the 3-kernel scan or one offset per sample */))
    return rc;
  if ((rc = dev_alloc(c, &c->d_out_own, cap * 8)))
    return rc;
  if ((rc = dev_alloc(c, &c->d_draw_ofs, cap)))
    return rc;
  if (c->huge_classes && (rc = dev_alloc(c, &c->d_huge_base, cap)))
    return rc;
  c->s_cap = cap;
  return AGH_OK;
}

// pinned staging of the host-buffer entry points: [header | sample indices | records] (kPinHeaderBytes: agh_internal.h)
constexpr int64_t kMirrorMaxRecords = 1 << 16;  // 10 MB of pinned memory at most; longer lists finish from the device copy
static inline int64_t pin_round(int64_t b) { return (b + 255) & ~(int64_t) 255; }
int ensure_host_staging(Ctx* c, int64_t samples, int64_t records)
{
  if (c->h_pin && samples <= c->h_pin_samples && records <= c->h_pin_records)
    return AGH_OK;
  const int64_t ns = std::max<int64_t>(std::max<int64_t>(samples, c->h_pin_samples), 1024);
  const int64_t nr = std::max<int64_t>(std::max<int64_t>(records, c->h_pin_records), 1024);
  if (c->h_pin)
  {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void) hipHostFree(c->h_pin);
    c->h_pin = nullptr;
    c->h_pin_samples = c->h_pin_records = 0;
  }
  void* p = nullptr;
  HIPCHK(c, hipHostMalloc(&p, (size_t) (kPinHeaderBytes + pin_round(ns * 4) + nr * (int64_t) sizeof(agh_hypothesis)), hipHostMallocDefault));
  c->h_pin = static_cast<uint8_t*>(p);
  c->h_pin_samples = ns;
  c->h_pin_records = nr;
  return AGH_OK;
}

// per-cloud grid tables for a batch of C clouds
int ensure_clouds(Ctx* c, int C)
{
  if (C <= c->clouds_cap)
    return AGH_OK;
  int rc;
  if ((rc = dev_alloc(c, &c->d_desc, (size_t) C)) || (rc = dev_alloc(c, &c->d_cell_start, (size_t) C * (size_t) kCellStride)) ||
      (rc = dev_alloc(c, &c->d_cell_count, (size_t) C * kCellCap)) || (rc = dev_alloc(c, &c->d_bbox_part, (size_t) C * kBboxBlocks * 6)) ||
      (rc = dev_alloc(c, &c->d_tile_state, (size_t) C * (kCellCap / 1024))))
    return rc;
  c->clouds_cap = C;
  c->grid_clean = false;  // fresh tables: the next build resets them
  return AGH_OK;
}

int ensure_draws(Ctx* c, int64_t count, hipStream_t st)
{
  if (count <= c->draws_cap)
    return AGH_OK;
  std::vector<int32_t> h((size_t) count);
  GlibcRand g(c->p.rand_seed);
  for (int64_t i = 0; i < count; i++)
    h[i] = g.next();
  int rc;
  if ((rc = dev_alloc(c, &c->d_draws, (size_t) count)))
    return rc;
  HIPCHK(c, hipMemcpyAsync(c->d_draws, h.data(), sizeof(int32_t) * count, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipStreamSynchronize(st));
  c->draws_cap = count;
  return AGH_OK;
}

__global__ void k_iota(int32_t* out, int base, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    out[i] = base + i;
}

// hand_search.cpp:13-26: findQuadrics over the cloud points [p0, p1) with r = nn_radius_normals, writing cloud_normals_
int normals_pass(Ctx* c, int64_t p0, int64_t p1, hipStream_t st)
{
  for (int64_t b0 = p0; b0 < p1; b0 += kNormalsChunk)
  {
    const int nb = (int) std::min<int64_t>(kNormalsChunk, p1 - b0);
    hipLaunchKernelGGL(k_iota, dim3((nb + 255) / 256), dim3(256), 0, st, c->d_samples, (int) b0, nb);
    const int rc = taubin_frames(c, c->d_samples, nb, c->p.nn_radius_normals, c->d_frames, c->d_nt, true, st);
    if (rc != AGH_OK)
      return rc;
  }
  return AGH_OK;
}
}  // namespace agh

namespace agh
{

void timing_begin(Ctx* c, hipStream_t st)
{
  if (c->p.profile != 1 && c->ev_used <= 60000)
    return;
  if (c->ev_used > 60000)  // nobody is reading the timings: start over instead of growing without bound
  {
    c->ev_used = 0;
    c->ev_name.clear();
  }
  if (c->p.profile == 1)
    timing_mark(c, "start", st);
}

}  // namespace agh

namespace agh
{
void timing_mark(Ctx* c, const char* name, hipStream_t st)
{
  if (!c->p.profile)
    return;
  if (c->p.profile >= 2)  // only k_hand_sweep is timed, by the two events its own launch carries (timing_launch_events)
    return;
  hipEvent_t e = timing_next_event(c, name);
  if (e)
    (void) hipEventRecord(e, st);
}

// The next event of the pool, entered into the list under `name` (nullptr if none can be created).
hipEvent_t timing_next_event(Ctx* c, const char* name)
{
  if (c->ev_used >= (int) c->ev.size())
  {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess)
      return nullptr;
    c->ev.push_back(e);
  }
  c->ev_name.push_back(name);
  return c->ev[c->ev_used++];
}

// profile 2 / 3: the start and stop events of a k_hand_sweep launch (hipExtLaunchKernelGGL attaches them to the dispatch
// itself: no barrier packets between the dependent kernels, where two hipEventRecord calls cost ~6 us per step); profile 3
// hands them out on every fourth call only.  false: launch without events.
bool timing_launch_events(Ctx* c, const char* name, hipEvent_t* start, hipEvent_t* stop)
{
  *start = *stop = nullptr;
  if (c->p.profile < 2)
    return false;
  if (c->p.profile == 3 && (c->prof_calls++ & 3u) != 0)
    return false;
  hipEvent_t a = timing_next_event(c, "start");
  hipEvent_t b = a ? timing_next_event(c, name) : nullptr;
  if (!b)
  {
    if (a)
    {
      c->ev_used--;
      c->ev_name.pop_back();
    }
    return false;
  }
  *start = a;
  *stop = b;
  return true;
}
}  // namespace agh

extern "C" {

void agh_default_params(agh_params* p)
{
  std::memset(p, 0, sizeof(*p));
  p->finger_width = 0.01;         // find_grasps.cpp:13
  p->hand_outer_diameter = 0.09;  // find_grasps.cpp:14
  p->hand_depth = 0.06;           // find_grasps.cpp:15
  p->init_bite = 0.01;            // find_grasps.cpp:16
  p->hand_height = 0.02;          // find_grasps.cpp:17
  p->nn_radius_taubin = 0.03;     // hand_search.h:85
  p->nn_radius_hands = 0.08;      // hand_search.h:85
  p->nn_radius_normals = 0.01;    // hand_search.cpp:20
  p->normals_mode = AGH_NORMALS_DETERMINISTIC;
  p->rand_seed = 1;
  p->device = 0;
  p->profile = 0;
}

const char* agh_last_error(const agh_ctx* ctx)
{
  return ctx ? ctx->c.err.c_str() : g_create_error.c_str();
}

int agh_create(const agh_params* p, agh_ctx** out)
{
  if (!p || !out)
    return AGH_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
  {
    g_create_error = "no HIP device is visible (this library has no CPU path)";
    return AGH_ERR_NO_DEVICE;
  }
  if (p->device < 0 || p->device >= ndev)
  {
    g_create_error = "device ordinal out of range";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, p->device) != hipSuccess)
  {
    g_create_error = "hipGetDeviceProperties failed";
    return AGH_ERR_HIP;
  }
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
  {
    g_create_error = std::string("device is ") + prop.gcnArchName + ", this build targets gfx950 (MI355X) only";
    return AGH_ERR_NO_DEVICE;
  }
  if (!(p->finger_width > 0) || !(p->hand_outer_diameter > p->finger_width) || !(p->hand_depth > 0) ||
      !(p->nn_radius_taubin > 0) || !(p->nn_radius_hands > 0) || !(p->nn_radius_normals > 0))
  {
    g_create_error = "invalid hand geometry or radii";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  agh_ctx* ctx = new agh_ctx();
  Ctx* c = &ctx->c;
  c->p = *p;
  c->device = p->device;
#ifdef AGH_DEBUG_HOOKS  // phase-timing aids of scripts/phase_timing.py; not compiled into the product build
  if (const char* e = std::getenv("AGH_DEBUG_STOP_SWEEP"))
    c->debug_stop_sweep = std::atoi(e);
  if (const char* e = std::getenv("AGH_DEBUG_STOP_MOMENTS"))
    c->debug_stop_moments = std::atoi(e);
  if (const char* e = std::getenv("AGH_DEBUG_STOP_FRAME"))
    c->debug_stop_frame = std::atoi(e);
  if (const char* e = std::getenv("AGH_DEBUG_STOP_HOG"))
    c->debug_stop_hog = std::atoi(e);
#endif
  std::string gerr;
  build_geometry(*p, &c->geom, &gerr);
  if (!gerr.empty())
  {
    g_create_error = gerr;
    delete ctx;
    return AGH_ERR_INVALID_ARGUMENT;
  }
  auto fail = [&](int rc) {
    g_create_error = c->err;
    agh_destroy(ctx);
    return rc;
  };
  if (hipSetDevice(c->device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess)
  {
    c->err = "hipSetDevice/hipStreamCreate failed";
    return fail(AGH_ERR_HIP);
  }
  int rc;
  if ((rc = ensure_clouds(c, 1)) || (rc = dev_alloc(c, &c->d_cloud_off, (size_t) kMaxClouds + 1)) ||
      (rc = dev_alloc(c, &c->d_block_sums, 4096)) ||
      (rc = dev_alloc(c, &c->d_flags, 8)) || (rc = dev_alloc(c, &c->d_nout, 1)) ||
      (rc = dev_alloc(c, &c->d_geom, 1)) || (rc = dev_alloc(c, &c->d_hog, 1)) ||
      (rc = dev_alloc(c, &c->d_svm_w, 3528)))
    return fail(rc);
  HogTablesDev* ht = new HogTablesDev();
  hog_tables_host(ht);
  hipError_t e1 = hipMemcpy(c->d_hog, ht, sizeof(HogTablesDev), hipMemcpyHostToDevice);
  delete ht;
  hipError_t e2 = hipMemcpy(c->d_geom, &c->geom, sizeof(HandGeom), hipMemcpyHostToDevice);
  hipError_t e3 = hipMemset(c->d_flags, 0, 8 * sizeof(int32_t));
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess)
  {
    c->err = "uploading tables failed";
    return fail(AGH_ERR_HIP);
  }
  *out = ctx;
  return AGH_OK;
}

void agh_destroy(agh_ctx* ctx)
{
  if (!ctx)
    return;
  Ctx* c = &ctx->c;
  (void) hipSetDevice(c->device);
  if (c->stream)
    (void) hipStreamSynchronize(c->stream);
  comm_release(c);
  void* ptrs[] = { c->own_xyz, c->own_cam, c->d_desc, c->d_bbox_part, c->d_cell_start, c->d_cell_count, c->d_block_sums, c->d_cell_of,
    c->d_rank_of, c->d_sorted, c->d_samples, c->d_sums, c->d_nt, c->d_nh, c->d_status, c->d_nbr, c->d_eig, c->d_frames, c->d_slots,
    c->d_images, c->d_slot_index, c->d_scan_tmp, c->d_out_own, c->d_nout, c->d_out_images, c->d_draw_ofs, c->d_draws,
    c->d_flags, c->d_normals, c->d_svm_w, c->d_hog, c->d_geom, c->d_desc_out, c->d_svm_sums, c->d_keep, c->d_vox_desc,
    c->d_weight, c->d_order, c->d_order_sweep, c->d_vmask, c->d_cloud_off, c->d_scloud, c->d_idx_own, c->d_tile_state, c->d_h_hands, c->d_h_bits, c->d_h_rowcnt, c->d_h_first,
    c->d_h_n, c->d_h_idx, c->d_h_counts, c->d_h_handles, c->d_h_tmp, c->d_images_cam, c->d_xbuf, c->d_nbuf, c->d_xcnt, c->d_cls_images, c->d_cls_keep, c->d_cls_sums, c->d_dbg, c->d_svm_svT, c->d_svm_alpha, c->d_cls_desc, c->d_cls_kbuf, c->d_vox_code, c->d_vox_blk, c->d_vox_blk2, c->d_vox_total, c->d_vox_bitmap, c->d_vox_xyz, c->d_vox_cam, c->d_raw_xyz, c->d_huge_stage, c->d_huge_key, c->d_huge_count, c->d_huge_sorted, c->d_huge_normals, c->d_huge_base, c->d_stage_xyz, c->d_ovf };
  for (void* p : ptrs)
    if (p)
      (void) hipFree(p);
  if (c->h_pin)
    (void) hipHostFree(c->h_pin);
  if (c->h_pin_handles)
    (void) hipHostFree(c->h_pin_handles);
  if (c->h_pin_keep)
    (void) hipHostFree(c->h_pin_keep);
  if (c->h_vox_desc)
    (void) hipHostFree(c->h_vox_desc);
  for (hipEvent_t e : c->ev)
    (void) hipEventDestroy(e);
  if (c->copy_done)
    (void) hipEventDestroy(c->copy_done);
  if (c->copy_gate)
    (void) hipEventDestroy(c->copy_gate);
  if (c->xyz_copied)
    (void) hipEventDestroy(c->xyz_copied);
  if (c->copy_stream)
    (void) hipStreamDestroy(c->copy_stream);
  if (c->stage_done)
    (void) hipEventDestroy(c->stage_done);
  if (c->stage_stream)
    (void) hipStreamDestroy(c->stage_stream);
  if (c->stream)
    (void) hipStreamDestroy(c->stream);
  delete ctx;
}

namespace
{
struct CloudOffArg
{
  int32_t v[agh::kMaxClouds + 1];
};
__global__ void k_set_cloud_off(CloudOffArg a, int32_t* __restrict__ out)
{
  if (threadIdx.x <= agh::kMaxClouds)
    out[threadIdx.x] = a.v[threadIdx.x];
}
}  // namespace

int agh_set_cloud_batch_device(agh_ctx* ctx, const float* d_xyz, int64_t stride_bytes, const int32_t* d_cam_source,
  const int64_t* offsets, int32_t n_clouds, void* hip_stream)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (!offsets || n_clouds < 1 || n_clouds > kMaxClouds || offsets[0] != 0)
  {
    c->err = "agh_set_cloud_batch: need 1 <= n_clouds <= 64 and offsets[0] == 0";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  for (int k = 0; k < n_clouds; k++)
    if (offsets[k + 1] < offsets[k])
    {
      c->err = "agh_set_cloud_batch: offsets must not decrease";
      return AGH_ERR_INVALID_ARGUMENT;
    }
  const int64_t n = offsets[n_clouds];
  if (n < 0 || n >= (1ll << 30) || stride_bytes < 12 || (stride_bytes % 4) != 0 || (n > 0 && !d_xyz))
  {
    c->err = "agh_set_cloud: need 0 <= n < 2^30 points in total, stride_bytes >= 12 and a multiple of 4";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t st = hip_stream ? (hipStream_t) hip_stream : c->stream;
  HIPCHK(c, order_after_cloud(c, st));  // (a build an earlier host-buffer agh_set_cloud left running writes the same tables)
  int rc = ensure_clouds(c, n_clouds);
  if (rc != AGH_OK)
    return rc;
  c->n = n;
  c->d_xyz = d_xyz;
  c->stride_floats = stride_bytes / 4;
  c->d_cam = d_cam_source;
  c->has_normals = false;
  // agh_localize: n is only a BOUND of the cloud's size -- the voxeliser, still queued on this stream, writes the true offsets
  // {0, count} into d_cloud_off itself; every kernel of the build and of the search reads them there
  const bool bound = c->defer_cloud_count && n_clouds == 1;
  c->defer_cloud_count = false;
  c->n_is_bound = bound;
  // the offsets travel to the device only when they change (a stream of equally sized clouds re-uses them)
  bool same = c->n_clouds == n_clouds && (int) c->cloud_off.size() == n_clouds + 1 && c->cloud_off_on_device;
  for (int k = 0; same && k <= n_clouds; k++)
    same = c->cloud_off[(size_t) k] == offsets[k];
  if (bound)
  {
    c->n_clouds = 1;
    c->cloud_off.assign(offsets, offsets + 2);
    c->cloud_off_on_device = false;  // (what the device holds is the true count, not this bound)
  }
  else if (!same)
  {
    c->n_clouds = n_clouds;
    c->cloud_off.assign(offsets, offsets + n_clouds + 1);
    c->cloud_off_i32.assign((size_t) kMaxClouds + 1, (int32_t) n);
    for (int k = 0; k <= n_clouds; k++)
      c->cloud_off_i32[(size_t) k] = (int32_t) offsets[k];
    // by value through a kernel's arguments: no pageable copy, no host synchronisation on the caller's stream (a stream of
    // voxelised clouds changes its point count with every frame)
    CloudOffArg a;
    for (int k = 0; k <= kMaxClouds; k++)
      a.v[k] = c->cloud_off_i32[(size_t) k];
    hipLaunchKernelGGL(k_set_cloud_off, dim3(1), dim3(128), 0, st, a, c->d_cloud_off);
    HIPCHK(c, hipGetLastError());
    c->cloud_off_on_device = true;
  }
  if (n > c->grid_cap)
  {
    if ((rc = dev_alloc(c, &c->d_cell_of, (size_t) n)) || (rc = dev_alloc(c, &c->d_rank_of, (size_t) n)) ||
        (rc = dev_alloc(c, &c->d_sorted, (size_t) n)))
      return rc;
    c->grid_cap = n;
  }
  timing_begin(c, st);
  rc = grid_build(c, st);
  timing_mark(c, "grid_build", st);
  if (rc != AGH_OK)
  {
    c->err = "grid build launch failed";
    return rc;
  }
  c->has_cloud = true;
  c->last_nout = -1;
  return AGH_OK;
}

int agh_set_cloud_device(agh_ctx* ctx, const float* d_xyz, int64_t stride_bytes, const int32_t* d_cam_source,
  int64_t n, void* hip_stream)
{
  const int64_t offsets[2] = { 0, n };
  return agh_set_cloud_batch_device(ctx, d_xyz, stride_bytes, d_cam_source, offsets, 1, hip_stream);
}

int agh_set_cloud(agh_ctx* ctx, const float* xyz, int64_t stride_bytes, const int32_t* cam_source, int64_t n)
{
  const int64_t offsets[2] = { 0, n };
  return agh_set_cloud_batch(ctx, xyz, stride_bytes, cam_source, offsets, 1);
}

// Page-locked host memory (hipHostMalloc / hipHostRegister)?  A copy from it is truly asynchronous: it returns before the
// source has been read, where a pageable copy returns after.
static bool is_pinned_host(const void* p)
{
  if (!p)
    return false;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess)
  {
    (void) hipGetLastError();  // (an unregistered pointer is an error on older runtimes, hipMemoryTypeUnregistered on newer ones)
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

int agh_set_cloud_batch(agh_ctx* ctx, const float* xyz, int64_t stride_bytes, const int32_t* cam_source, const int64_t* offsets,
  int32_t n_clouds)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (!offsets || n_clouds < 1 || n_clouds > kMaxClouds)
  {
    c->err = "agh_set_cloud_batch: need 1 <= n_clouds <= 64 and offsets";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  const int64_t n = offsets[n_clouds];
  if (n < 0 || n >= (1ll << 30) || stride_bytes < 12 || (stride_bytes % 4) != 0 || (n > 0 && !xyz))
  {
    c->err = "agh_set_cloud: need 0 <= n < 2^30, stride_bytes >= 12 and a multiple of 4";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  HIPCHK(c, hipSetDevice(c->device));
  // The points are uploaded as they lie in host memory (one contiguous copy: a strided 2-D copy of 12 of every 32
  // bytes runs at a fraction of the PCIe rate) and the kernels read them with the caller's stride; strides above 32
  // bytes are repacked to 12.
  const bool as_is = stride_bytes <= 32;
  const int64_t dev_stride = as_is ? stride_bytes : 12;
  const int64_t need = n * (dev_stride / 4);
  if (need > c->own_cap_floats || n > c->own_cap)
  {
    int rc;
    if ((rc = dev_alloc(c, &c->own_xyz, (size_t) std::max<int64_t>(need, c->own_cap_floats))) ||
        (rc = dev_alloc(c, &c->own_cam, (size_t) std::max<int64_t>(n, c->own_cap))))
      return rc;
    c->own_cap_floats = std::max<int64_t>(need, c->own_cap_floats);
    c->own_cap = std::max<int64_t>(n, c->own_cap);
  }
  // "The caller may overwrite xyz and cam_source as soon as the call returns" (agh.h) holds by itself for pageable sources (the
  // copy calls return when the source has been read); for PINNED sources the copies are waited for below -- the copies only,
  // the grid build stays queued (a frame buffer reused at once corrupted the upload silently).
  const bool xyz_pinned = n > 0 && is_pinned_host(xyz), cam_pinned = n > 0 && is_pinned_host(cam_source);
  c->cam_copy_on_copy_stream = false;
  if (n > 0)
  {
    if (as_is)
      HIPCHK(c, hipMemcpyAsync(c->own_xyz, xyz, (size_t) (n * stride_bytes - (stride_bytes - 12)), hipMemcpyHostToDevice,
                  c->stream));  // (the last point's padding may lie outside the caller's buffer)
    else
      HIPCHK(c, hipMemcpy2DAsync(c->own_xyz, 12, xyz, (size_t) stride_bytes, 12, (size_t) n, hipMemcpyHostToDevice,
                  c->stream));
    if (xyz_pinned)
    {
      if (!c->xyz_copied)
        HIPCHK(c, hipEventCreateWithFlags(&c->xyz_copied, hipEventDisableTiming));
      HIPCHK(c, hipEventRecord(c->xyz_copied, c->stream));
    }
    if (cam_source)  // (uploaded by grid_build, overlapped with its coordinate-only kernels)
    {
      c->pending_cam_host = cam_source;
      c->pending_cam_n = n;
    }
    else
      HIPCHK(c, hipMemsetAsync(c->own_cam, 0, sizeof(int32_t) * n, c->stream));
  }
  // (The pageable copies above return when the caller's buffers have been read, so they may be reused at once; the grid
  // build is left running on the context's stream -- whatever uses the cloud next is ordered behind it there, and an
  // asynchronous failure surfaces at that call's synchronisation.  Waiting here cost every cloud ~17 us plus the build.)
  const int rc = agh_set_cloud_batch_device(ctx, c->own_xyz, dev_stride, c->own_cam, offsets, n_clouds, nullptr);
  if (c->pending_cam_host)  // the build did not get as far as the upload (an early error return)
  {
    c->pending_cam_host = nullptr;
    if (rc == AGH_OK)
      HIPCHK(c, hipMemcpyAsync(c->own_cam, cam_source, sizeof(int32_t) * n, hipMemcpyHostToDevice, c->stream));
  }
  if (xyz_pinned)
    HIPCHK(c, hipEventSynchronize(c->xyz_copied));
  if (cam_pinned)
  {
    if (c->cam_copy_on_copy_stream)
      HIPCHK(c, hipEventSynchronize(c->copy_done));
    else  // the ids went up on the context's stream (no copy stream, or the fallback above): behind the build's kernels
      HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  c->cloud_async = rc == AGH_OK;
  return rc;
}

// ---- f1: preprocessing (NaN removal, workspace box, per-camera voxelisation), then the grid build ----
int agh_preprocess_device(agh_ctx* ctx, const float* d_xyz, int64_t stride_bytes, int64_t n, int64_t size_left,
  int dense, const double workspace[6], double cell_size, int64_t* n_voxels_out, void* hip_stream)
{
  return preprocess_device_impl(ctx, d_xyz, stride_bytes, n, size_left, dense, workspace, cell_size, n_voxels_out, hip_stream, false,
    nullptr);
}

// defer_count (agh_localize): when the context already has a voxel bitmap to speculate with, nothing is waited for -- the voxel
// count stays on the device (d_cloud_off, written by the voxeliser's scan), the grid build is queued for the BOUND n, and
// *deferred tells the caller that the descriptor (error word, counts) is still to be read from the pinned mirror after its own
// synchronisation.  Without a bitmap (a context's first cloud) the call behaves as agh_preprocess_device.
extern "C++" int preprocess_device_impl(agh_ctx* ctx, const float* d_xyz, int64_t stride_bytes, int64_t n, int64_t size_left,
  int dense, const double workspace[6], double cell_size, int64_t* n_voxels_out, void* hip_stream, bool defer_count, bool* deferred)
{
  if (deferred)
    *deferred = false;
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (n < 0 || n >= (1ll << 30) || stride_bytes < 12 || (stride_bytes % 4) != 0 || (n > 0 && !d_xyz) || !workspace ||
      !(cell_size > 0.0) || size_left < 0)
  {
    c->err = "agh_preprocess: need 0 <= n < 2^30, stride_bytes >= 12 and a multiple of 4, a workspace, cell_size > 0";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t st = hip_stream ? (hipStream_t) hip_stream : c->stream;
  int rc;
  if (!c->d_vox_desc)
  {
    if ((rc = dev_alloc(c, &c->d_vox_desc, 1)) || (rc = dev_alloc(c, &c->d_vox_total, 1)) ||
        (rc = dev_alloc(c, &c->d_vox_blk2, (size_t) (kVoxMaxWords / 4096) + 1)))
      return rc;
  }
  if (n > c->vox_cap || !c->d_vox_code)
  {
    if ((rc = dev_alloc(c, &c->d_vox_code, (size_t) n)) || (rc = dev_alloc(c, &c->d_vox_blk, (size_t) n / 1024 + 2)) ||
        (rc = dev_alloc(c, &c->d_vox_xyz, (size_t) n * 3)) || (rc = dev_alloc(c, &c->d_vox_cam, (size_t) n)))
      return rc;
    c->vox_cap = n;
  }
  if (!c->h_vox_desc)
  {
    void* p = nullptr;
    HIPCHK(c, hipHostMalloc(&p, sizeof(VoxDesc), hipHostMallocDefault));
    c->h_vox_desc = static_cast<VoxDesc*>(p);
  }
  timing_begin(c, st);
  // Round trips: the lattice's size (which sizes the bitmap) is only known after stage 1, the voxel count (which sizes the
  // search grid) after stage 2.  The second one stays: every launch of the grid build and of the search is sized by it on
  // the host.  The first is avoided from the second cloud on: stage 2 runs at once for the bitmap the context already has
  // (a stream of captures of one workspace keeps its lattice within a few per cent), the blocks beyond the lattice are
  // empty, and a lattice that does not fit raises error 2 -- the bitmap is enlarged and both stages are repeated.  The
  // descriptor reaches the host through a pinned mirror the kernels write (a pageable read-back cost ~45 us each).
  VoxDesc h;
  for (int attempt = 0;; attempt++)
  {
    // (a bitmap kept from a much larger lattice is dropped: every later cloud would clear and count all of it)
    if (c->d_vox_bitmap && c->vox_last_words > 0 && c->vox_bitmap_cap > 8 * c->vox_last_words + (1 << 20))
    {
      (void) hipFree(c->d_vox_bitmap);
      c->d_vox_bitmap = nullptr;
      c->vox_bitmap_cap = 0;
    }
    const bool speculative = c->d_vox_bitmap && c->vox_bitmap_cap > 0;
    if ((rc = vox_stage1(c, d_xyz, stride_bytes / 4, n, size_left, dense, workspace, cell_size, st,
           speculative ? c->vox_bitmap_cap : (int64_t) kVoxMaxWords, c->h_vox_desc, !speculative)) != AGH_OK)
    {
      c->err = "preprocessing launch failed";
      return rc;
    }
    if (!speculative)
    {
      HIPCHK(c, hipStreamSynchronize(st));  // the first cloud of the context: the lattice size decides the bitmap allocation
      h = *c->h_vox_desc;
      if (h.error)
        break;
      // a quarter of headroom -- never beyond the lattice limit the block tables (d_vox_blk2: kVoxMaxWords / 4096 + 1 entries)
      // are sized for: the unclamped size reached 81 921 blocks for lattices above 0.8 x 2^28 words
      const int64_t want = std::min<int64_t>((((int64_t) h.n_words + (int64_t) h.n_words / 4) / 4096 + 1) * 4096, (int64_t) kVoxMaxWords);
      if ((rc = dev_alloc(c, &c->d_vox_bitmap, (size_t) want + 4096)))
        return rc;
      c->vox_bitmap_cap = want;
    }
    const bool defer = defer_count && speculative && n > 0;
    if ((rc = vox_stage2(c, d_xyz, stride_bytes / 4, n, cell_size, c->vox_bitmap_cap, st, c->h_vox_desc, speculative,
           defer ? c->d_cloud_off : nullptr)) != AGH_OK)
    {
      c->err = "preprocessing launch failed";
      return rc;
    }
    timing_mark(c, "preprocess", st);
    if (defer)
    {
      *deferred = true;
      c->defer_cloud_count = true;
      const int rc_set = agh_set_cloud_device(ctx, c->d_vox_xyz, 12, c->d_vox_cam, n, hip_stream);
      c->defer_cloud_count = false;  // (also when the call left before it consumed the flag: it must never reach the caller's NEXT cloud)
      return rc_set;
    }
    HIPCHK(c, hipStreamSynchronize(st));  // the voxel count sizes the search structure
    h = *c->h_vox_desc;
    if (h.error != 2 || attempt >= 1)
      break;
    // the lattice outgrew the bitmap: drop it, so that the next pass sizes a new one from this cloud's lattice
    (void) hipFree(c->d_vox_bitmap);
    c->d_vox_bitmap = nullptr;
    c->vox_bitmap_cap = 0;
  }
  if (h.error)
  {
    c->err = "the voxel lattice of the kept points exceeds 2^33 cells (1 GiB bitmap): set a workspace "
             "(Localization::setWorkspace) that bounds the scene";
    return AGH_ERR_CAPACITY;
  }
  c->vox_last_words = (int64_t) h.n_words;
  const int64_t nv = (int64_t) (h.n_vox[0] + h.n_vox[1]);
  if (n_voxels_out)
    *n_voxels_out = nv;
  return agh_set_cloud_device(ctx, c->d_vox_xyz, 12, c->d_vox_cam, nv, hip_stream);
}

int agh_preprocess(agh_ctx* ctx, const float* xyz, int64_t stride_bytes, int64_t n, int64_t size_left, int dense,
  const double workspace[6], double cell_size, int64_t* n_voxels_out)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (n < 0 || n >= (1ll << 30) || stride_bytes < 12 || (stride_bytes % 4) != 0 || (n > 0 && !xyz))
  {
    c->err = "agh_preprocess: need 0 <= n < 2^30, stride_bytes >= 12 and a multiple of 4";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  HIPCHK(c, hipSetDevice(c->device));
  const bool as_is = stride_bytes <= 32;  // one contiguous upload, read with the caller's stride (see agh_set_cloud)
  const int64_t dev_stride = as_is ? stride_bytes : 12;
  const int64_t need = n * (dev_stride / 4);
  if (need > c->raw_cap || !c->d_raw_xyz)
  {
    int rc;
    if ((rc = dev_alloc(c, &c->d_raw_xyz, (size_t) need)))
      return rc;
    c->raw_cap = need;
  }
  if (n > 0)
  {
    if (as_is)
      HIPCHK(c, hipMemcpyAsync(c->d_raw_xyz, xyz, (size_t) (n * stride_bytes - (stride_bytes - 12)), hipMemcpyHostToDevice,
                  c->stream));
    else
      HIPCHK(c, hipMemcpy2DAsync(c->d_raw_xyz, 12, xyz, (size_t) stride_bytes, 12, (size_t) n, hipMemcpyHostToDevice,
                  c->stream));
  }
  // (the grid build is left running on the context's stream, as after agh_set_cloud)
  const int rc = agh_preprocess_device(ctx, c->d_raw_xyz, dev_stride, n, size_left, dense, workspace, cell_size, n_voxels_out,
    nullptr);
  c->cloud_async = rc == AGH_OK;
  return rc;
}

// ---- f2: handle search ----
// acos(x) < 0.34 <=> x >= x1 and M_PI - acos(x) < 0.34 <=> x <= x2 for the host's libm (the one the reference's
// safeAcos would call): found by bisection on the doubles, monotonicity checked around the result.
extern "C++" bool handle_thresholds(double* x1, double* x2)
{
  double lo = 0.9, hi = 1.0;  // acos(lo) >= 0.34 > acos(hi)
  if (!(std::acos(lo) >= 0.34) || !(std::acos(hi) < 0.34))
    return false;
  while (std::nextafter(lo, hi) != hi)
  {
    const double mid = lo + (hi - lo) / 2.0;
    if (std::acos(mid) < 0.34)
      hi = mid;
    else
      lo = mid;
  }
  *x1 = hi;
  double a = -1.0, b = -0.9;  // M_PI - acos(a) < 0.34 <= M_PI - acos(b)
  if (!(M_PI - std::acos(a) < 0.34) || (M_PI - std::acos(b) < 0.34))
    return false;
  while (std::nextafter(a, b) != b)
  {
    const double mid = a + (b - a) / 2.0;
    if (M_PI - std::acos(mid) < 0.34)
      a = mid;
    else
      b = mid;
  }
  *x2 = a;
  double t = *x1, u = *x2;
  for (int k = 0; k < 64; k++)  // a few ulps either side behave monotonically
  {
    t = std::nextafter(t, 2.0);
    u = std::nextafter(u, -2.0);
    if (!(std::acos(t) < 0.34) || !(M_PI - std::acos(u) < 0.34))
      return false;
  }
  t = *x1;
  u = *x2;
  for (int k = 0; k < 64; k++)
  {
    t = std::nextafter(t, -2.0);
    u = std::nextafter(u, 2.0);
    if (std::acos(t) < 0.34 || M_PI - std::acos(u) < 0.34)
      return false;
  }
  return true;
}

// device buffers of the handle search for up to n_hands hands, and its pinned staging: the hands go up with an asynchronous copy
// (agh_find_handles) or are written there by the device (agh_localize); the handles, the inlier lists and the counts are written
// to host memory by the kernels themselves -- one synchronisation, no read-back copies
extern "C++" int ensure_handle_buffers(Ctx* c, int64_t n_hands)
{
  if (n_hands > c->h_cap || !c->d_h_counts)
  {
    const size_t cap = (size_t) std::max<int64_t>(n_hands, 256);
    int rc;
    if ((rc = dev_alloc(c, &c->d_h_hands, cap)) || (rc = dev_alloc(c, &c->d_h_bits, cap * ((cap + 63) / 64))) ||
        (rc = dev_alloc(c, &c->d_h_rowcnt, cap)) || (rc = dev_alloc(c, &c->d_h_first, cap)) ||
        (rc = dev_alloc(c, &c->d_h_n, cap)) || (rc = dev_alloc(c, &c->d_h_idx, cap)) ||
        (rc = dev_alloc(c, &c->d_h_counts, 8)) || (rc = dev_alloc(c, &c->d_h_handles, cap)) ||
        (rc = dev_alloc(c, &c->d_h_tmp, cap)))
      return rc;
    c->h_cap = (int64_t) cap;
  }
  if (n_hands > c->h_pin_handles_cap || !c->h_pin_handles)
  {
    const int64_t cap = std::max<int64_t>(n_hands, 1024);
    if (c->h_pin_handles)
    {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      (void) hipHostFree(c->h_pin_handles);
      c->h_pin_handles = nullptr;
      c->h_pin_handles_cap = 0;
    }
    void* p = nullptr;
    HIPCHK(c, hipHostMalloc(&p, (size_t) (256 + cap * (int64_t) (sizeof(agh_hypothesis) + sizeof(agh_handle) + sizeof(int32_t))),
                hipHostMallocDefault));
    c->h_pin_handles = static_cast<uint8_t*>(p);
    c->h_pin_handles_cap = cap;
  }
  return AGH_OK;
}

int agh_find_handles(agh_ctx* ctx, const agh_hypothesis* hands, int64_t n_hands, int32_t min_inliers, double min_length,
  agh_handle* handles_out, int64_t handle_cap, int32_t* inlier_idx_out, int64_t idx_cap, int64_t* n_handles_out)
{
  if (!ctx || !n_handles_out || n_hands < 0 || (n_hands > 0 && !hands) || handle_cap < 0 || idx_cap < 0 || min_inliers < 1)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  *n_handles_out = 0;
  if (n_hands > 8192)
  {
    c->err = "agh_find_handles: more than 8192 hands (the search runs on the hands Learning::classify kept)";
    return AGH_ERR_CAPACITY;
  }
  double x1 = 0, x2 = 0;
  if (!handle_thresholds(&x1, &x2))
  {
    c->err = "agh_find_handles: this libm's acos is not monotone around the 0.34 rad thresholds";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  HIPCHK(c, hipSetDevice(c->device));
  {
    const int rc = ensure_handle_buffers(c, n_hands);
    if (rc != AGH_OK)
      return rc;
  }
  int* h_counts = reinterpret_cast<int*>(c->h_pin_handles);
  agh_hypothesis* h_hands = reinterpret_cast<agh_hypothesis*>(c->h_pin_handles + 256);
  agh_handle* h_handles = reinterpret_cast<agh_handle*>(h_hands + c->h_pin_handles_cap);
  int32_t* h_idx = reinterpret_cast<int32_t*>(h_handles + c->h_pin_handles_cap);
  if (n_hands > 0)
  {
    std::memcpy(h_hands, hands, sizeof(agh_hypothesis) * (size_t) n_hands);
    HIPCHK(c, hipMemcpyAsync(c->d_h_hands, h_hands, sizeof(agh_hypothesis) * n_hands, hipMemcpyHostToDevice, c->stream));
  }
  const HandleMirror hm{ h_handles, (int) c->h_pin_handles_cap, h_idx, (int) c->h_pin_handles_cap, h_counts };
  for (int attempt = 0; attempt < 2; attempt++)
  {
    h_counts[0] = h_counts[1] = h_counts[2] = h_counts[3] = 0;
    timing_begin(c, c->stream);
    const bool with_sequential = c->handles_sequential;
    const int rc = handle_search(c, n_hands, x1, x2, min_inliers, min_length, c->stream, hm, with_sequential);
    timing_mark(c, "handle_search", c->stream);
    if (rc != AGH_OK)
    {
      c->err = "handle search launch failed";
      return rc;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const bool declined = h_counts[3] != 0;  // a row of the pair matrix longer than a wave
    c->handles_sequential = declined;        // what this set of hands needed is the guess for the next one
    if (!declined || with_sequential)
      break;  // (declined without the sequential kernel behind it: once more, with it)
  }
  const int* counts = h_counts;
  if (counts[2])
  {
    c->err = "agh_find_handles: a seed hand has more than 2048 inliers";
    return AGH_ERR_CAPACITY;
  }
  *n_handles_out = counts[0];
  if (counts[0] > handle_cap || counts[1] > idx_cap)
  {
    c->err = "agh_find_handles: output buffers too small";
    return AGH_ERR_CAPACITY;
  }
  if (counts[0] > 0)
  {
    std::memcpy(handles_out, h_handles, sizeof(agh_handle) * (size_t) counts[0]);
    std::memcpy(inlier_idx_out, h_idx, sizeof(int32_t) * (size_t) counts[1]);
  }
  return AGH_OK;
}

int agh_get_cloud(agh_ctx* ctx, float* xyz_out, int32_t* cam_out, int64_t cap)
{
  if (!ctx || cap < 0)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (!c->has_cloud)
  {
    c->err = "agh_get_cloud: no cloud set";
    return AGH_ERR_NO_CLOUD;
  }
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  const int64_t k = std::min<int64_t>(cap, c->n);
  if (k > 0 && xyz_out)
    HIPCHK(c, hipMemcpy2D(xyz_out, 12, c->d_xyz, (size_t) c->stride_floats * 4, 12, (size_t) k, hipMemcpyDeviceToHost));
  if (k > 0 && cam_out)
  {
    if (c->d_cam)
      HIPCHK(c, hipMemcpy(cam_out, c->d_cam, sizeof(int32_t) * k, hipMemcpyDeviceToHost));
    else
      std::memset(cam_out, 0, sizeof(int32_t) * k);
  }
  return (int) std::min<int64_t>(c->n, 0x7fffffff);
}

int agh_find_hands_device(agh_ctx* ctx, const int32_t* d_sample_idx, int64_t n_samples, int calculates_antipodal,
  agh_hypothesis* d_out, int64_t cap, int64_t* d_n_out, void* hip_stream)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (!c->has_cloud)
  {
    c->err = "agh_find_hands: no cloud set";
    return AGH_ERR_NO_CLOUD;
  }
  if (n_samples < 0 || n_samples > (1 << 24) || (n_samples > 0 && (!d_sample_idx || !d_out)) || !d_n_out || cap < 0)
  {
    c->err = "agh_find_hands: bad arguments";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t st = hip_stream ? (hipStream_t) hip_stream : c->stream;
  HIPCHK(c, order_after_cloud(c, st));
  const int64_t S = n_samples;
  const int64_t chunk = kNormalsChunk;  // all-points pass batch
  int rc = ensure_call_buffers(c, std::max<int64_t>(S, calculates_antipodal ? std::min<int64_t>(c->n, chunk) : 0));
  if (rc != AGH_OK)
    return rc;
  const bool rand_mode = c->p.normals_mode == AGH_NORMALS_RAND50;
  if (rand_mode)
  {
    rc = ensure_draws(c, 50 * (S + (calculates_antipodal ? c->n : 0)), st);
    if (rc != AGH_OK)
      return rc;
  }
  timing_begin(c, st);
  c->zero_flags_pending = true;  // the first kernel of the call clears the error flags (no memset launch)
  c->last_s = S;
  c->epoch = next_epoch();
  c->last_nout = -1;
  c->last_cap = cap;
  c->d_out_last = d_out;
  c->d_nout_last = d_n_out;
  if (S == 0 || c->n == 0)
  {
    HIPCHK(c, hipMemsetAsync(c->d_flags, 0, 8 * sizeof(int32_t), st));
    c->zero_flags_pending = false;
    HIPCHK(c, hipMemsetAsync(d_n_out, 0, sizeof(int64_t), st));
    c->last_s = 0;
    return AGH_OK;
  }
  if (calculates_antipodal)
  {
    // hand_search.cpp:13-26: cloud_normals_ zeroed, then findQuadrics over ALL points with r = 0.01
    if (c->n > c->normals_cap)
    {
      if ((rc = dev_alloc(c, &c->d_normals, (size_t) c->n * 3)))
        return rc;
      c->normals_cap = c->n;
    }
    HIPCHK(c, hipMemsetAsync(c->d_normals, 0, sizeof(double) * 3 * c->n, st));
    if ((rc = normals_pass(c, 0, c->n, st)) != AGH_OK)
    {
      c->err = "normals pass launch failed";
      return rc;
    }
    c->has_normals = true;
  }
  rc = taubin_frames(c, d_sample_idx, S, c->p.nn_radius_taubin, c->d_frames, c->d_nt, calculates_antipodal != 0, st);
  if (rc != AGH_OK)
  {
    c->err = "taubin launch failed";
    return rc;
  }
  if (c->debug_stop_moments || (c->debug_stop_frame && c->debug_stop_frame < 5))
  {
    HIPCHK(c, hipMemsetAsync(d_n_out, 0, sizeof(int64_t), st));
    return AGH_OK;
  }
#ifdef AGH_DEBUG_HOOKS
  if (std::getenv("AGH_DEBUG_CLOCKS") && !c->d_dbg && hipMalloc((void**) &c->d_dbg, sizeof(long long) * 8 * c->s_cap) != hipSuccess)
    c->d_dbg = nullptr;
#endif
  rc = hand_sweep(c, d_sample_idx, S, calculates_antipodal != 0, st);
  if (rc != AGH_OK)
  {
    c->err = "hand sweep launch failed";
    return rc;
  }
  if (c->debug_stop_sweep && c->debug_stop_sweep < 10)
  {
    HIPCHK(c, hipMemsetAsync(d_n_out, 0, sizeof(int64_t), st));
    return AGH_OK;
  }
  rc = compact_hypotheses(c, S, d_out, cap, d_n_out, st);
  if (rc != AGH_OK)
  {
    c->err = "compaction launch failed";
    return rc;
  }
  return AGH_OK;
}


static int check_flags(Ctx* c, hipStream_t st)
{
  int32_t flags[8];
  HIPCHK(c, hipMemcpyAsync(flags, c->d_flags, sizeof(flags), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return flags_to_status(c, flags);
}

extern "C++" int flags_to_status(Ctx* c, const int32_t* flags_in)
{
  // after a sharded search the capacity decision comes from the gathered segment headers (the same on every rank of the
  // communicator), not from this rank's own bit 0 and big_classes: agh_internal.h, kFlagShard*
  int32_t flags[1] = { flags_in[0] };
  if (flags[0] & kFlagSharded)
  {
    flags[0] &= ~1;
    if (flags[0] & kFlagShardHard)
    {
      c->err = "the Taubin neighbourhoods (r = nn_radius_taubin) of more than 4096 points of one launch hold more than 2^21 points in "
               "all; voxelise the cloud (localization.cpp:43) or reduce the radii";
      return AGH_ERR_CAPACITY;
    }
    if (flags[0] & (kFlagShardRetry | kFlagShardRetryHuge))
    {
      c->big_classes = true;
      if (flags[0] & kFlagShardRetryHuge)
        c->huge_classes = true;
      c->err = "a Taubin neighbourhood exceeds the capacity classes launched so far; the contexts of the communicator now launch "
               "the larger classes as well: repeat the call";
      return AGH_ERR_RETRY;
    }
  }
  if ((flags[0] & 1) && !c->big_classes)
  {
    // the launches of the larger capacity classes are skipped until a cloud needs them (~5 us each, and class membership
    // is only known on the device): this call met such a neighbourhood, so its samples beyond the first class have no
    // frame yet.  From now on the context launches every class.
    c->big_classes = true;
    c->err = "a Taubin neighbourhood exceeds the first capacity class; the context now launches the larger classes as "
             "well: repeat the call";
    return AGH_ERR_RETRY;
  }
  if ((flags[0] & 1) && !c->huge_classes)
  {
    // ... and a neighbourhood beyond 4096 points needs the 6144 class (K1a through global scratch): the same on-demand switch
    c->huge_classes = true;
    c->err = "a Taubin neighbourhood exceeds 4096 points; the context now launches the 6144 class as well: repeat the call";
    return AGH_ERR_RETRY;
  }
  if (flags[0] & 1)
  {
    c->err = "the Taubin neighbourhoods (r = nn_radius_taubin) of more than 4096 points of one launch hold more than 2^21 points in "
             "all (the pool of the classes beyond the LDS-resident ones); voxelise the cloud (localization.cpp:43) or reduce the radii";
    return AGH_ERR_CAPACITY;
  }
  if (flags[0] & kFlagShardPeerNoCloud)
  {
    c->err = "a rank of the communicator holds no cloud (agh_set_cloud* on every rank first); no list";
    return AGH_ERR_NO_CLOUD;
  }
  if (flags[0] & kFlagShardPeerFailed)
  {
    c->err = "a rank of the communicator could not do its share of the call (its own agh_last_error says why); no list";
    return AGH_ERR_STATE;
  }
  if ((flags[0] & 2) && !(flags[0] & 16))
  {
    c->err = "output buffer too small for the hypotheses found";
    return AGH_ERR_CAPACITY;
  }
  if (flags[0] & 4)
  {
    c->err = "a sample index is outside the cloud";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  if (flags[0] & 16)
  {
    // sharded search: every rank reads the same segment headers, so every rank lands here and switches together
    c->shard_full_exchange = true;
    c->err = "a rank found more hypotheses than its exchange segment holds; the context now exchanges full-size segments: "
             "repeat the call";
    return AGH_ERR_RETRY;
  }
  return AGH_OK;
}

int agh_find_hands(agh_ctx* ctx, const int32_t* sample_idx, int64_t n_samples, int calculates_antipodal,
  agh_hypothesis* out, int64_t cap, int64_t* n_out)
{
  if (!ctx || !n_out)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  *n_out = 0;
  if (!c->has_cloud)
  {
    c->err = "agh_find_hands: no cloud set";
    return AGH_ERR_NO_CLOUD;
  }
  if (n_samples < 0 || (n_samples > 0 && !sample_idx) || cap < 0 || (cap > 0 && !out))
  {
    c->err = "agh_find_hands: bad arguments";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  for (int64_t i = 0; i < n_samples; i++)
    if (sample_idx[i] < 0 || sample_idx[i] >= c->n)
    {
      c->err = "agh_find_hands: sample index out of range";
      return AGH_ERR_INVALID_ARGUMENT;
    }
  HIPCHK(c, hipSetDevice(c->device));
  // size the buffers for everything the device call will need, so that it does not reallocate d_out_own under us
  int rc = ensure_call_buffers(c, std::max<int64_t>(n_samples, calculates_antipodal ? std::min<int64_t>(c->n, kNormalsChunk) : 0));
  if (rc != AGH_OK)
    return rc;
  // own sample buffer (d_samples doubles as the iota scratch of the normals pass)
  if (n_samples > c->idx_cap || !c->d_idx_own)
  {
    if ((rc = dev_alloc(c, &c->d_idx_own, (size_t) std::max<int64_t>(n_samples, 1024))))
      return rc;
    c->idx_cap = std::max<int64_t>(n_samples, 1024);
  }
  int32_t* d_idx = c->d_idx_own;
  // Pinned staging owned by the context: the sample list goes up with an asynchronous copy (a pageable hipMemcpyAsync blocks
  // the host for ~9 us and puts a staging kernel on the stream), and the list comes back WITHOUT read-back copies: K4 writes
  // every record and the [count | error word] header a second time, into this buffer, so one stream synchronisation is all
  // the call waits for (round 3 paid three pageable read-backs behind the last kernel: ~0.1 ms of a 0.31 ms call,
  // profiles/r04_host_timeline_api.txt).
  if ((rc = ensure_host_staging(c, n_samples, std::min<int64_t>(c->s_cap * 8, kMirrorMaxRecords))) != AGH_OK)
    return rc;
  int64_t* hdr = reinterpret_cast<int64_t*>(c->h_pin);
  int32_t* h_idx = reinterpret_cast<int32_t*>(c->h_pin + kPinHeaderBytes);
  agh_hypothesis* h_rec = reinterpret_cast<agh_hypothesis*>(c->h_pin + kPinHeaderBytes + pin_round(c->h_pin_samples * 4));
  if (n_samples > 0)
  {
    std::memcpy(h_idx, sample_idx, sizeof(int32_t) * (size_t) n_samples);
    HIPCHK(c, hipMemcpyAsync(d_idx, h_idx, sizeof(int32_t) * n_samples, hipMemcpyHostToDevice, c->stream));
  }
  int64_t n = 0;
  for (int attempt = 0; attempt < 3; attempt++)  // (AGH_ERR_RETRY: the context has enabled the larger classes; then, once, the 6144 class)
  {
    // (a retry may have switched a capacity class on that wants larger per-sample scratch: sized HERE, so that the device call does
    // not reallocate d_out_own after it was handed over)
    if ((rc = ensure_call_buffers(c, std::max<int64_t>(n_samples, calculates_antipodal ? std::min<int64_t>(c->n, kNormalsChunk) : 0))) != AGH_OK)
      return rc;
    hdr[0] = -1;  // (stays -1 if no concatenation kernel ran: empty input, a debug stop)
    hdr[1] = 0;
    c->mirror = HostMirror{ h_rec, c->h_pin_records, hdr };
    rc = agh_find_hands_device(ctx, d_idx, n_samples, calculates_antipodal, c->d_out_own, c->s_cap * 8, c->d_nout,
      c->stream);
    c->mirror = HostMirror{ nullptr, 0, nullptr };
    if (rc == AGH_OK)
    {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      int32_t flags[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
      if (hdr[0] >= 0)
      {
        n = hdr[0];
        flags[0] = (int32_t) hdr[1] | (n > c->s_cap * 8 ? 2 : 0);
      }
      else  // no header was written: read the count and the error word the slow way
      {
        HIPCHK(c, hipMemcpy(flags, c->d_flags, sizeof(flags), hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(&n, c->d_nout, sizeof(int64_t), hipMemcpyDeviceToHost));
      }
      rc = flags_to_status(c, flags);
    }
    else
      (void) hipStreamSynchronize(c->stream);
    if (rc != AGH_ERR_RETRY)
      break;
  }
  if (rc != AGH_OK)
    return rc;
  c->last_nout = n;
  *n_out = n;
  if (n > cap)
  {
    c->err = "output buffer too small for the hypotheses found";
    return AGH_ERR_CAPACITY;
  }
  const int64_t from_pin = hdr[0] >= 0 ? std::min<int64_t>(n, c->h_pin_records) : 0;
  if (from_pin > 0)
    std::memcpy(out, h_rec, sizeof(agh_hypothesis) * (size_t) from_pin);
  if (n > from_pin)  // (lists beyond the mirror's room: the rest comes from the device copy)
    HIPCHK(c, hipMemcpy(out + from_pin, c->d_out_own + from_pin, sizeof(agh_hypothesis) * (size_t) (n - from_pin), hipMemcpyDeviceToHost));
  return AGH_OK;
}

int agh_synchronize(agh_ctx* ctx)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
#ifdef AGH_DEBUG_HOOKS
  if (c->d_dbg && c->last_s > 0 && std::getenv("AGH_DEBUG_CLOCKS"))  // development aid: dump the phase timestamps
  {
    std::vector<long long> h((size_t) c->last_s * 8);
    if (hipMemcpy(h.data(), c->d_dbg, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
      if (FILE* f = std::fopen(std::getenv("AGH_DEBUG_CLOCKS"), "wb"))
      {
        std::fwrite(h.data(), 8, h.size(), f);
        std::fclose(f);
      }
  }
#endif
  const int rc = check_flags(c, c->stream);
  if (rc == AGH_OK && c->last_nout < 0 && c->d_nout_last)  // the count of an asynchronous call, now known to the host
  {
    int64_t n = 0;
    if (hipMemcpy(&n, c->d_nout_last, sizeof(int64_t), hipMemcpyDeviceToHost) == hipSuccess)
      c->last_nout = std::min<int64_t>(n, c->last_cap);
    else
      (void) hipGetLastError();  // (the caller may have released its buffer: not an error of this call)
  }
  return rc;
}

int agh_get_epoch(agh_ctx* ctx, int32_t* epoch, int64_t* n_hyp)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  if (epoch)
    *epoch = ctx->c.epoch;
  if (n_hyp)
    *n_hyp = ctx->c.last_nout;
  return AGH_OK;
}

int agh_get_frames(agh_ctx* ctx, agh_frame* out, int64_t cap)
{
  if (!ctx || !out)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  const int64_t n = std::min<int64_t>(cap, c->last_s);
  HIPCHK(c, hipDeviceSynchronize());
  if (n > 0)
    HIPCHK(c, hipMemcpy(out, c->d_frames, sizeof(agh_frame) * n, hipMemcpyDeviceToHost));
  return (int) n;
}

int agh_get_neighbor_counts(agh_ctx* ctx, int32_t* n_taubin, int32_t* n_hands, int64_t cap)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  const int64_t n = std::min<int64_t>(cap, c->last_s);
  HIPCHK(c, hipDeviceSynchronize());
  if (n > 0 && n_taubin)
    HIPCHK(c, hipMemcpy(n_taubin, c->d_nt, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  if (n > 0 && n_hands)  // counted on demand: the sweep only visits the part of the ball the hand can occupy
  {
    if (ball_counts(c, n, c->stream) != AGH_OK)
    {
      c->err = "k_ball_count launch failed";
      return AGH_ERR_HIP;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  if (n > 0 && n_hands)
    HIPCHK(c, hipMemcpy(n_hands, c->d_nh, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  return (int) n;
}

int agh_get_normals(agh_ctx* ctx, double* normals, int64_t cap_points)
{
  if (!ctx || !normals)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (!c->has_normals)
  {
    c->err = "no normals: the last agh_find_hands call did not use calculates_antipodal";
    return AGH_ERR_STATE;
  }
  const int64_t n = std::min<int64_t>(cap_points, c->n);
  HIPCHK(c, hipDeviceSynchronize());
  if (n > 0)
    HIPCHK(c, hipMemcpy(normals, c->d_normals, sizeof(double) * 3 * n, hipMemcpyDeviceToHost));
  return (int) n;
}

int agh_set_profile(agh_ctx* ctx, int32_t level)
{
  if (!ctx || level < 0 || level > 3)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  c->p.profile = level;
  c->prof_calls = 0;
  c->ev_used = 0;
  c->ev_name.clear();
  return AGH_OK;
}

int agh_get_timing(agh_ctx* ctx, agh_timing* out)
{
  // Sums the kernel times of every call since the previous agh_get_timing (segments between consecutive HIP
  // events recorded on the launch stream; a "start" event opens each call), then forgets them.
  if (!ctx || !out)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  std::memset(out, 0, sizeof(*out));
  std::memset(c->timing_counts, 0, sizeof(c->timing_counts));
  if (!c->p.profile || c->ev_used < 2)
    return AGH_OK;
  HIPCHK(c, hipEventSynchronize(c->ev[c->ev_used - 1]));
  int k = 0;
  for (int i = 1; i < c->ev_used; i++)
  {
    if (std::strcmp(c->ev_name[i], "start") == 0)
      continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev[i - 1], c->ev[i]) != hipSuccess)
      ms = 0.f;
    int slot = -1;
    for (int j = 0; j < k; j++)
      if (std::strcmp(out->name[j], c->ev_name[i]) == 0)
        slot = j;
    if (slot < 0)
    {
      if (k >= AGH_TIMING_SLOTS)
        continue;
      slot = k++;
      out->name[slot] = c->ev_name[i];
    }
    out->ms[slot] += ms;
    c->timing_counts[slot]++;
    out->total_ms += ms;
  }
  out->n = k;
  c->ev_used = 0;
  c->ev_name.clear();
  return AGH_OK;
}

int agh_get_timing_counts(agh_ctx* ctx, int32_t* counts, int32_t cap)
{
  if (!ctx || !counts || cap < 0)
    return AGH_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < cap && i < AGH_TIMING_SLOTS; i++)
    counts[i] = ctx->c.timing_counts[i];
  return AGH_OK;
}

int64_t agh_selftest_math(agh_ctx* ctx, int64_t n, uint64_t seed)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  return selftest_math(&ctx->c, n, seed);
}

}  // extern "C"
