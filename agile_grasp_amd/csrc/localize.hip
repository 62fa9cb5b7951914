// localize.hip -- the online chain of the reference's only online caller as one call (or two halves of one):
// GraspLocalizer::localizeGrasps, grasp_localizer.cpp:95-103 = localizeHands -> predictAntipodalHands -> findHandles per capture.
// agh_localize / agh_localize_device / agh_localize_begin / agh_localize_stage / agh_localize_end of include/agh.h; the stages
// themselves (preprocessing, search, classification, handle search) are api.hip's, voxelize.hip's, hog_svm.hip's and handles.hip's.
#include "agh_internal.h"

#include <algorithm>
#include <cstring>

using namespace agh;

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

extern "C" {

namespace
{
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x)
{
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// One sample per stratum of the cloud (include/agh.h, agh_localize); the point count is read on the device.
__global__ void k_draw_samples(const int* __restrict__ cloud_off, int n_clouds, int S, unsigned long long seed,
  int32_t* __restrict__ out, int32_t* __restrict__ host_out)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= S)
    return;
  const long long N = cloud_off[n_clouds];
  int32_t v;
  if (N >= S)
  {
    const long long lo = ((long long) k * N) / S, hi = ((long long) (k + 1) * N) / S;
    v = (int32_t) (lo + (long long) (splitmix64(seed ^ ((unsigned long long) k * 0x9E3779B97F4A7C15ull)) % (unsigned long long) (hi - lo)));
  }
  else
    v = k < N ? k : kSampleSkip;
  out[k] = v;
  if (host_out)
    host_out[k] = v;
}
// The hands Learning::classify kept (svm_keep; all of them if !use_keep), in list order (learning.cpp:236-243), as the handle
// search's input -- and a second time into pinned host memory.  One work-group: an ordered compaction is a scan.
// host_counts: [4] hypotheses, [5] kept, [6] the search's error word.
__global__ __launch_bounds__(1024) void k_compact_kept(const agh_hypothesis* __restrict__ in, const int64_t* __restrict__ n_in,
  int64_t cap_in, int use_keep, agh_hypothesis* __restrict__ out, int out_cap, int* __restrict__ n_out,
  agh_hypothesis* __restrict__ host_out, int host_cap, int* __restrict__ host_counts, const int32_t* __restrict__ flags)
{
  constexpr int kList = 8192;  // (the handle search takes no more)
  __shared__ int src[kList];   // position in the output -> position in the input
  __shared__ int wsum[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t n = min(*n_in, cap_in);
  if (tid == 0)
    carry = 0;
  __syncthreads();
  for (int64_t b0 = 0; b0 < n; b0 += 1024)
  {
    const int64_t i = b0 + tid;
    const bool keep = i < n && (!use_keep || in[i].svm_keep != 0);
    const unsigned long long m = __ballot(keep);
    if (lane == 0)
      wsum[wave] = __popcll(m);
    __syncthreads();
    int base = carry, tot = 0;
    for (int w = 0; w < 16; w++)
    {
      base += w < wave ? wsum[w] : 0;
      tot += wsum[w];
    }
    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
    if (keep && pos < kList)
      src[pos] = (int) i;
    __syncthreads();
    if (tid == 0)
      carry += tot;
    __syncthreads();
  }
  const int K = carry, kw = min(K, min(out_cap, kList));
  // ten threads per record, sixteen bytes each: every record leaves as one 160-byte run, to the device and to the host
  for (int t = tid; t < kw * 10; t += 1024)
  {
    const int k = t / 10, part = t - 10 * k;
    const uint4 v = reinterpret_cast<const uint4*>(in + src[k])[part];
    reinterpret_cast<uint4*>(out + k)[part] = v;
    if (host_out && k < host_cap)
      reinterpret_cast<uint4*>(host_out + k)[part] = v;
  }
  if (tid == 0)
  {
    *n_out = K;
    if (host_counts)
    {
      host_counts[4] = (int) n;
      host_counts[5] = K;
      host_counts[6] = flags[0] | (*n_in > cap_in ? 2 : 0);
    }
  }
}
}  // namespace

// agh_localize = agh_localize_begin (everything queued) + agh_localize_end (the one synchronisation, the results, the rare
// repeats).  Between the two the caller may stage the NEXT capture (agh_localize_stage: upload on a second stream into a second
// raw buffer, under this cloud's kernels), which the next begin adopts instead of uploading.  One chain is in flight at a time:
// the context's device buffers, pinned mirrors and host-side cloud state are single.
static int localize_begin_impl(agh_ctx* ctx, const float* xyz, bool xyz_on_device, int64_t stride_bytes, int64_t n,
  const agh_localize_params* lp);
static int localize_end_impl(agh_ctx* ctx, agh_handle* handles_out, int64_t handle_cap, int32_t* inlier_idx_out, int64_t idx_cap,
  agh_hypothesis* hands_out, int64_t hands_cap, int32_t* samples_out, agh_localize_result* result);

static int localize_check_outputs(Ctx* c, agh_handle* handles_out, int64_t handle_cap, int32_t* inlier_idx_out, int64_t idx_cap,
  agh_hypothesis* hands_out, int64_t hands_cap)
{
  if (handle_cap < 0 || idx_cap < 0 || hands_cap < 0 || (handle_cap > 0 && !handles_out) || (idx_cap > 0 && !inlier_idx_out) ||
      (hands_cap > 0 && !hands_out))
  {
    c->err = "agh_localize: bad arguments (see include/agh.h)";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  return AGH_OK;
}

int agh_localize(agh_ctx* ctx, const float* xyz, int64_t stride_bytes, int64_t n, const agh_localize_params* lp,
  agh_handle* handles_out, int64_t handle_cap, int32_t* inlier_idx_out, int64_t idx_cap, agh_hypothesis* hands_out,
  int64_t hands_cap, int32_t* samples_out, agh_localize_result* result)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  if (result)
    *result = agh_localize_result{ 0, 0, 0, 0, 0 };
  int rc = localize_check_outputs(&ctx->c, handles_out, handle_cap, inlier_idx_out, idx_cap, hands_out, hands_cap);
  if (rc == AGH_OK)
    rc = localize_begin_impl(ctx, xyz, false, stride_bytes, n, lp);
  if (rc != AGH_OK)
    return rc;
  return localize_end_impl(ctx, handles_out, handle_cap, inlier_idx_out, idx_cap, hands_out, hands_cap, samples_out, result);
}

int agh_localize_device(agh_ctx* ctx, const float* d_xyz, int64_t stride_bytes, int64_t n, const agh_localize_params* lp,
  agh_handle* handles_out, int64_t handle_cap, int32_t* inlier_idx_out, int64_t idx_cap, agh_hypothesis* hands_out,
  int64_t hands_cap, int32_t* samples_out, agh_localize_result* result)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  if (result)
    *result = agh_localize_result{ 0, 0, 0, 0, 0 };
  int rc = localize_check_outputs(&ctx->c, handles_out, handle_cap, inlier_idx_out, idx_cap, hands_out, hands_cap);
  if (rc == AGH_OK)
    rc = localize_begin_impl(ctx, d_xyz, true, stride_bytes, n, lp);
  if (rc != AGH_OK)
    return rc;
  return localize_end_impl(ctx, handles_out, handle_cap, inlier_idx_out, idx_cap, hands_out, hands_cap, samples_out, result);
}

int agh_localize_begin(agh_ctx* ctx, const float* xyz, int64_t stride_bytes, int64_t n, const agh_localize_params* lp)
{
  return localize_begin_impl(ctx, xyz, false, stride_bytes, n, lp);
}

int agh_localize_end(agh_ctx* ctx, agh_handle* handles_out, int64_t handle_cap, int32_t* inlier_idx_out, int64_t idx_cap,
  agh_hypothesis* hands_out, int64_t hands_cap, int32_t* samples_out, agh_localize_result* result)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  if (result)
    *result = agh_localize_result{ 0, 0, 0, 0, 0 };
  Ctx* c = &ctx->c;
  if (!c->loc.active)
  {
    c->err = "agh_localize_end: no agh_localize_begin in flight";
    return AGH_ERR_STATE;
  }
  const int rc = localize_check_outputs(c, handles_out, handle_cap, inlier_idx_out, idx_cap, hands_out, hands_cap);
  if (rc != AGH_OK)
  {
    // (the chain is queued: drain it, leave the context as a failed call does)
    (void) hipStreamSynchronize(c->stream);
    c->loc.active = false;
    if (c->n_is_bound)
    {
      c->n_is_bound = false;
      c->has_cloud = false;
      c->n = 0;
      c->cloud_off_on_device = false;
    }
    return rc;
  }
  return localize_end_impl(ctx, handles_out, handle_cap, inlier_idx_out, idx_cap, hands_out, hands_cap, samples_out, result);
}

// The NEXT capture up, beside the chain in flight: into the context's second raw buffer, on a stream of its own.  A pageable
// source makes the call last as long as the copy (the kernels of the chain in flight run meanwhile: that is the overlap); a
// pinned one returns at once.  The source must stay valid until the copy is done: until the agh_localize_begin that adopts the
// capture has returned (it makes the chain wait for the copy; the copy itself is then behind a host-side event wait).
int agh_localize_stage(agh_ctx* ctx, const float* xyz, int64_t stride_bytes, int64_t n)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (n < 0 || n >= (1ll << 30) || stride_bytes < 12 || (stride_bytes % 4) != 0 || (n > 0 && !xyz))
  {
    c->err = "agh_localize_stage: bad arguments";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  HIPCHK(c, hipSetDevice(c->device));
  c->loc.staged = false;
  if (!c->stage_stream)
  {
    HIPCHK(c, hipStreamCreateWithFlags(&c->stage_stream, hipStreamNonBlocking));
    if (hipEventCreateWithFlags(&c->stage_done, hipEventDisableTiming) != hipSuccess)
    {
      (void) hipStreamDestroy(c->stage_stream);
      c->stage_stream = nullptr;
      c->err = "agh_localize_stage: no event";
      return AGH_ERR_HIP;
    }
  }
  const bool as_is = stride_bytes <= 32;
  const int64_t dev_stride = as_is ? stride_bytes : 12;
  const int64_t need = n * (dev_stride / 4);
  if (need > c->stage_cap || !c->d_stage_xyz)
  {
    // (nobody reads this buffer now: the chain in flight reads d_raw_xyz)
    int rc;
    if ((rc = dev_alloc(c, &c->d_stage_xyz, (size_t) std::max<int64_t>(need, 1))))
      return rc;
    c->stage_cap = need;
  }
  if (n > 0)
  {
    if (as_is)
      HIPCHK(c, hipMemcpyAsync(c->d_stage_xyz, xyz, (size_t) (n * stride_bytes - (stride_bytes - 12)), hipMemcpyHostToDevice, c->stage_stream));
    else
      HIPCHK(c, hipMemcpy2DAsync(c->d_stage_xyz, 12, xyz, (size_t) stride_bytes, 12, (size_t) n, hipMemcpyHostToDevice, c->stage_stream));
  }
  HIPCHK(c, hipEventRecord(c->stage_done, c->stage_stream));
  c->loc.staged = true;
  c->loc.staged_src = xyz;
  c->loc.staged_stride = stride_bytes;
  c->loc.staged_n = n;
  return AGH_OK;
}

// search -> classification -> kept hands -> handle search, queued on the context's stream (handles_only: the handle search alone,
// once more, on the hands that are already there)
static int localize_queue(agh_ctx* ctx, bool handles_only)
{
  Ctx* c = &ctx->c;
  LocalizeState& L = c->loc;
  hipStream_t st = c->stream;
  int* h_counts = reinterpret_cast<int*>(c->h_pin_handles);
  agh_hypothesis* h_hands = reinterpret_cast<agh_hypothesis*>(c->h_pin_handles + 256);
  agh_handle* h_handles = reinterpret_cast<agh_handle*>(h_hands + c->h_pin_handles_cap);
  int32_t* h_hidx = reinterpret_cast<int32_t*>(h_handles + c->h_pin_handles_cap);
  const HandleMirror hm{ h_handles, (int) c->h_pin_handles_cap, h_hidx, (int) c->h_pin_handles_cap, h_counts };
  int* d_hcount = c->d_h_counts + 4;  // (behind the HandleCounts record)
  const int64_t hand_bound = std::min<int64_t>(8 * L.S, 8192);
  int rc;
  for (int k = 0; k < (handles_only ? 4 : 8); k++)  // ([4..6], the search's counts, outlive a repeat of the handle search alone)
    h_counts[k] = 0;
  L.with_sequential = c->handles_sequential;
  if (!handles_only)
  {
    c->mirror = HostMirror{ nullptr, 0, nullptr };
    if ((rc = agh_find_hands_device(ctx, c->d_idx_own, L.S, 0, c->d_out_own, c->s_cap * 8, c->d_nout, st)) != AGH_OK)
      return rc;
    if (L.classify && (rc = agh_classify_device(ctx, c->d_keep, st)) != AGH_OK)
      return rc;
    hipLaunchKernelGGL(k_compact_kept, dim3(1), dim3(1024), 0, st, (const agh_hypothesis*) c->d_out_own, (const int64_t*) c->d_nout,
      c->s_cap * 8, L.classify ? 1 : 0, c->d_h_hands, (int) hand_bound, d_hcount, h_hands, (int) c->h_pin_handles_cap, h_counts,
      (const int32_t*) c->d_flags);
    if (hipGetLastError() != hipSuccess)
    {
      c->err = "k_compact_kept launch failed";
      return AGH_ERR_HIP;
    }
  }
  timing_begin(c, st);
  rc = handle_search(c, hand_bound, L.x1, L.x2, L.min_inliers, L.min_length, st, hm, L.with_sequential, d_hcount);
  timing_mark(c, "handle_search", st);
  if (rc != AGH_OK)
    c->err = "handle search launch failed";
  return rc;
}

static int localize_begin_impl(agh_ctx* ctx, const float* xyz, bool xyz_on_device, int64_t stride_bytes, int64_t n, const agh_localize_params* lp)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  LocalizeState& L = c->loc;
  if (L.active)
  {
    c->err = "agh_localize_begin: a chain is in flight (agh_localize_end first)";
    return AGH_ERR_STATE;
  }
  if (!lp || n < 0 || n >= (1ll << 30) || stride_bytes < 12 || (stride_bytes % 4) != 0 || (n > 0 && !xyz) || !(lp->cell_size > 0.0) ||
      lp->size_left < 0 || lp->n_samples < 0 || lp->n_samples > (1 << 24) || lp->min_inliers < 1)
  {
    c->err = "agh_localize: bad arguments (see include/agh.h)";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  if (lp->classify && !c->has_svm)
  {
    c->err = "agh_localize: classify needs a loaded SVM (agh_load_svm*)";
    return AGH_ERR_NO_SVM;
  }
  if (!handle_thresholds(&L.x1, &L.x2))
  {
    c->err = "agh_localize: this libm's acos is not monotone around the 0.34 rad thresholds";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  HIPCHK(c, hipSetDevice(c->device));
  const int64_t S = lp->n_samples;
  hipStream_t st = c->stream;
  int rc;
  // ---- 1. raw cloud up (unless it is on the device already: agh_localize_device, which reads it in place with the caller's
  // stride -- or was staged: agh_localize_stage), voxelisation and grid build queued; the voxel count stays on the device when it
  // can ----
  const bool as_is = stride_bytes <= 32;  // (as agh_preprocess)
  const int64_t dev_stride = (as_is || xyz_on_device) ? stride_bytes : 12;
  const float* d_raw = xyz;
  if (!xyz_on_device)
  {
    if (L.staged && L.staged_src == xyz && L.staged_stride == stride_bytes && L.staged_n == n && c->d_stage_xyz)
    {
      // the capture is (or is about to be) in the second raw buffer: the two buffers change places, the chain waits for the copy
      std::swap(c->d_raw_xyz, c->d_stage_xyz);
      std::swap(c->raw_cap, c->stage_cap);
      L.staged = false;
      HIPCHK(c, hipStreamWaitEvent(st, c->stage_done, 0));
    }
    else
    {
      L.staged = false;  // (a staged capture that is not this one is dropped)
      const int64_t need = n * (dev_stride / 4);
      if (need > c->raw_cap || !c->d_raw_xyz)
      {
        if ((rc = dev_alloc(c, &c->d_raw_xyz, (size_t) need)))
          return rc;
        c->raw_cap = need;
      }
      if (n > 0)
      {
        if (as_is)
          HIPCHK(c, hipMemcpyAsync(c->d_raw_xyz, xyz, (size_t) (n * stride_bytes - (stride_bytes - 12)), hipMemcpyHostToDevice, st));
        else
          HIPCHK(c, hipMemcpy2DAsync(c->d_raw_xyz, 12, xyz, (size_t) stride_bytes, 12, (size_t) n, hipMemcpyHostToDevice, st));
      }
    }
    d_raw = c->d_raw_xyz;
  }
  L.S = S;
  L.classify = lp->classify != 0;
  L.min_inliers = lp->min_inliers;
  L.min_length = lp->min_length;
  L.lp = *lp;
  L.lp.sample_idx = nullptr;  // (the list is copied below; a repeat of the whole call reads it from the pinned copy)
  L.explicit_samples = lp->sample_idx != nullptr;
  L.d_raw = d_raw;
  L.dev_stride = dev_stride;
  L.n_raw = n;
  L.deferred = false;
  L.nv = 0;
  rc = preprocess_device_impl(ctx, d_raw, dev_stride, n, lp->size_left, lp->dense, lp->workspace, lp->cell_size, &L.nv, nullptr,
    true, &L.deferred);
  if (rc != AGH_OK)
    return rc;
  c->cloud_async = false;  // (everything below is queued on the context's own stream, and the call ends with its synchronisation)
  // a failure between the launches and the synchronisation, while the host only knows a BOUND of the cloud's size: the context
  // must not be left believing the bound is the cloud
  auto drop_bound_cloud = [&]() {
    if (c->n_is_bound)
    {
      c->n_is_bound = false;
      c->has_cloud = false;
      c->n = 0;
      c->cloud_off_on_device = false;
    }
  };
  // (every error return from here on first drains the stream -- a pinned source may still be in flight, the caller may free it
  // as soon as the call returns -- and drops the bound)
  auto fail = [&](int code) {
    (void) hipStreamSynchronize(st);
    drop_bound_cloud();
    return code;
  };
#define LOC_HIPCHK(expr)                                                  \
  do                                                                      \
  {                                                                       \
    hipError_t e__ = (expr);                                              \
    if (e__ != hipSuccess)                                                \
    {                                                                     \
      c->err = std::string(#expr) + ": " + hipGetErrorString(e__);        \
      return fail(AGH_ERR_HIP);                                           \
    }                                                                     \
  } while (0)
  // ---- 2. buffers for the bounds ----
  if ((rc = ensure_call_buffers(c, std::max<int64_t>(S, 1))) != AGH_OK)  // (S = 0: the later stages still want their buffers)
    return fail(rc);
  if (S > c->idx_cap || !c->d_idx_own)
  {
    if ((rc = dev_alloc(c, &c->d_idx_own, (size_t) std::max<int64_t>(S, 1024))))
      return fail(rc);
    c->idx_cap = std::max<int64_t>(S, 1024);
  }
  if ((rc = ensure_host_staging(c, S, 1024)) != AGH_OK)
    return fail(rc);
  int32_t* h_idx = reinterpret_cast<int32_t*>(c->h_pin + kPinHeaderBytes);
  const int64_t hyp_bound = 8 * S;
  const int64_t hand_bound = std::min<int64_t>(hyp_bound, 8192);
  if ((rc = ensure_handle_buffers(c, hand_bound)) != AGH_OK)
    return fail(rc);
  if (lp->classify && c->s_cap * 8 > c->keep_cap)
  {
    if (c->d_keep)
      (void) hipFree(c->d_keep);
    if (c->d_svm_sums)
      (void) hipFree(c->d_svm_sums);
    c->d_keep = nullptr;
    c->d_svm_sums = nullptr;
    c->keep_cap = 0;
    LOC_HIPCHK(hipMalloc((void**) &c->d_keep, (size_t) (c->s_cap * 8)));
    LOC_HIPCHK(hipMalloc((void**) &c->d_svm_sums, (size_t) (c->s_cap * 8) * sizeof(double)));
    c->keep_cap = c->s_cap * 8;
  }
  // ---- 3. the sample list ----
  if (S > 0)
  {
    if (lp->sample_idx)
    {
      if (lp->sample_idx != h_idx)  // (a repeat of the whole call hands the pinned copy back in)
        std::memcpy(h_idx, lp->sample_idx, sizeof(int32_t) * (size_t) S);
      LOC_HIPCHK(hipMemcpyAsync(c->d_idx_own, h_idx, sizeof(int32_t) * S, hipMemcpyHostToDevice, st));
    }
    else
    {
      hipLaunchKernelGGL(k_draw_samples, dim3((unsigned) ((S + 255) / 256)), dim3(256), 0, st, (const int*) c->d_cloud_off, 1, (int) S,
        (unsigned long long) lp->sample_seed, c->d_idx_own, h_idx);
      LOC_HIPCHK(hipGetLastError());
    }
  }
  // ---- 4. search -> classification -> kept hands -> handle search: queued; agh_localize_end waits ----
  if ((rc = localize_queue(ctx, false)) != AGH_OK)
    return fail(rc);
  L.active = true;
  return AGH_OK;
}
#undef LOC_HIPCHK

static int localize_end_impl(agh_ctx* ctx, agh_handle* handles_out, int64_t handle_cap, int32_t* inlier_idx_out, int64_t idx_cap,
  agh_hypothesis* hands_out, int64_t hands_cap, int32_t* samples_out, agh_localize_result* result)
{
  Ctx* c = &ctx->c;
  LocalizeState& L = c->loc;
  hipStream_t st = c->stream;
  L.active = false;
  const int64_t S = L.S;
  int* h_counts = reinterpret_cast<int*>(c->h_pin_handles);
  agh_hypothesis* h_hands = reinterpret_cast<agh_hypothesis*>(c->h_pin_handles + 256);
  agh_handle* h_handles = reinterpret_cast<agh_handle*>(h_hands + c->h_pin_handles_cap);
  int32_t* h_hidx = reinterpret_cast<int32_t*>(h_handles + c->h_pin_handles_cap);
  int32_t* h_idx = reinterpret_cast<int32_t*>(c->h_pin + kPinHeaderBytes);
  auto drop_bound_cloud = [&]() {
    if (c->n_is_bound)
    {
      c->n_is_bound = false;
      c->has_cloud = false;
      c->n = 0;
      c->cloud_off_on_device = false;
    }
  };
  int rc;
  bool handles_only = false;
  for (int attempt = 0;; attempt++)
  {
    if (attempt > 0)  // (attempt 0 was queued by agh_localize_begin)
    {
      if ((rc = localize_queue(ctx, handles_only)) != AGH_OK)
      {
        (void) hipStreamSynchronize(st);
        drop_bound_cloud();
        return rc;
      }
    }
    const bool with_sequential = L.with_sequential;
    if (hipStreamSynchronize(st) != hipSuccess)
    {
      drop_bound_cloud();
      c->err = "agh_localize: hipStreamSynchronize failed";
      return AGH_ERR_HIP;
    }
    if (L.deferred)  // the descriptor of the speculative voxelisation, now on the host
    {
      L.deferred = false;
      const VoxDesc h = *c->h_vox_desc;
      L.nv = (int64_t) (h.n_vox[0] + h.n_vox[1]);
      c->n_is_bound = false;
      if (h.error)
      {
        // error 2: the lattice outgrew the bitmap kept from the previous cloud -- the whole call once more, sized from this
        // cloud's lattice (the context then has no bitmap to speculate with: the preprocessing takes its own round trips).  The
        // raw capture is still where the chain read it: in the context's raw buffer, or in the caller's device memory.
        c->has_cloud = false;
        c->n = 0;
        c->cloud_off_on_device = false;
        if (h.error == 2 && !L.repeated)
        {
          (void) hipFree(c->d_vox_bitmap);
          c->d_vox_bitmap = nullptr;
          c->vox_bitmap_cap = 0;
          agh_localize_params lp = L.lp;
          lp.sample_idx = L.explicit_samples ? h_idx : nullptr;
          L.repeated = true;
          rc = localize_begin_impl(ctx, L.d_raw, true, L.dev_stride, L.n_raw, &lp);
          if (rc == AGH_OK)
            rc = localize_end_impl(ctx, handles_out, handle_cap, inlier_idx_out, idx_cap, hands_out, hands_cap, samples_out, result);
          c->loc.repeated = false;
          return rc;
        }
        c->err = "the voxel lattice of the kept points exceeds 2^33 cells (1 GiB bitmap): set a workspace "
                 "(Localization::setWorkspace) that bounds the scene";
        return AGH_ERR_CAPACITY;
      }
      c->vox_last_words = (int64_t) h.n_words;
      c->n = L.nv;
      c->cloud_off.assign({ (int64_t) 0, L.nv });
      c->cloud_off_on_device = true;  // ({0, nv}: what the voxeliser wrote)
      c->n_clouds = 1;
    }
    if (!handles_only)
    {
      int32_t flags[1] = { h_counts[6] };
      rc = flags_to_status(c, flags);
      if (rc == AGH_ERR_RETRY && attempt < 3)
      {
        // (the larger capacity classes are on now: the search once more, on the cloud that is already there)
        if ((rc = ensure_call_buffers(c, std::max<int64_t>(S, 1))) != AGH_OK)
          return rc;
        continue;
      }
      if (rc != AGH_OK)
        return rc;
    }
    if (h_counts[2] == 2 || h_counts[5] > 8192)
    {
      c->err = "agh_localize: more than 8192 hands for the handle search (classify first, or search fewer samples)";
      return AGH_ERR_CAPACITY;
    }
    const bool declined = h_counts[3] != 0;  // a row of the pair matrix longer than a wave (see agh_find_handles)
    c->handles_sequential = declined;
    if (declined && !with_sequential && attempt < 3)
    {
      handles_only = true;
      continue;
    }
    break;
  }
  if (h_counts[2])
  {
    c->err = "agh_localize: a seed hand has more than 2048 inliers";
    return AGH_ERR_CAPACITY;
  }
  const int64_t n_hyp = h_counts[4], n_kept = h_counts[5];
  c->last_nout = std::min<int64_t>(n_hyp, c->s_cap * 8);
  if (result)
    *result = agh_localize_result{ L.nv, n_hyp, n_kept, h_counts[0], h_counts[1] };
  if (samples_out && S > 0)
    std::memcpy(samples_out, h_idx, sizeof(int32_t) * (size_t) S);
  if (h_counts[0] > handle_cap || h_counts[1] > idx_cap || (hands_out && n_kept > hands_cap))
  {
    c->err = "agh_localize: output buffers too small (the counts are in *result)";
    return AGH_ERR_CAPACITY;
  }
  if (h_counts[0] > 0)
  {
    std::memcpy(handles_out, h_handles, sizeof(agh_handle) * (size_t) h_counts[0]);
    std::memcpy(inlier_idx_out, h_hidx, sizeof(int32_t) * (size_t) h_counts[1]);
  }
  if (hands_out && n_kept > 0)
    std::memcpy(hands_out, h_hands, sizeof(agh_hypothesis) * (size_t) n_kept);
  return AGH_OK;
}

}  // extern "C"
