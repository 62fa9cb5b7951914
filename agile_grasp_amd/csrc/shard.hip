// shard.hip -- one cloud's sample set sharded over the GPUs of a node (north star: "the sample set shards
// across the 8 GPUs of one node with an RCCL all-gather of hypotheses over xGMI").
//
// The reference's two OpenMP loops (hand_search.cpp:77-80 findQuadrics, 135-138 findHands) run over independent samples
// that read a shared immutable cloud and write their own result slots.  Rank g of G takes the contiguous slice
// [g*S/G, (g+1)*S/G) of the sample list, runs the unchanged single-GPU chain (K1a..K2, K4) on it, and the ranks' compacted
// lists are concatenated in rank order -- which is the reference's sample-major order.  The exchange is ONE in-place
// ncclAllGather of fixed-size segments [160-byte header (count) | seg_records x agh_hypothesis], issued from here (host
// code stays C++; no Python in the data path) on the search's stream, followed by one merge kernel.
//
// Where the reference's loops share state, the shards exchange it first:
//  * calculates_antipodal: cloud_normals_ (hand_search.cpp:13-26) is filled by an all-points pass; each rank does a point
//    range of it and the 3 x N doubles are all-gathered in place.  findQuadrics over the samples then overwrites the
//    samples' own columns (hand_search.cpp:102): those normals are all-gathered too and scattered by every rank.
//  * AGH_NORMALS_RAND50: the rand() stream is consumed in sample order (quadric.cpp:184), so a rank's first draw offset is
//    the number of draws the earlier slices consume: one int per rank, all-gathered between K1b and K1c.
//
// RCCL is bound at run time (dlopen of librccl.so.1): a single-GPU deployment does not need it installed.
// agh_comm_init_local replaces the all-gather by device copies between the contexts of one process (one host thread per
// rank): RCCL refuses two ranks on one GPU, and this is how the schedule is validated on a single-GPU machine -- every
// kernel and every offset is the same, only the transport differs.
#include "agh_internal.h"

#include <dlfcn.h>

// RCCL is opened at run time (dlopen) and NOT needed to build this library: the handful of types and constants of
// <rccl/rccl.h> (the NCCL 2 ABI: an opaque communicator pointer, the 128-byte unique id, result 0 = success, data
// type 0 = char) are declared here instead of including the header.
extern "C" {
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct
{
  char internal[NCCL_UNIQUE_ID_BYTES];
} ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
}
static constexpr ncclResult_t ncclSuccess = 0;
static constexpr ncclDataType_t ncclChar = 0;

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>

namespace agh
{

namespace
{
struct Rccl
{
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
  std::string origin;  // which image the entry points come from (agh_comm_rccl_origin)
};

Rccl* rccl()
{
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // A process that has PyTorch loaded already has an RCCL image mapped (torch/lib/librccl.so); loading a second copy
    // beside it would give the process two collective runtimes with two sets of proxy threads and IPC state.  So: an image
    // that is already mapped wins (RTLD_NOLOAD only succeeds for those), then AGH_RCCL_LIB, then the system library.
    const char* env = std::getenv("AGH_RCCL_LIB");
    const char* mapped[] = { "librccl.so", "librccl.so.1" };
    for (const char* n : mapped)
      if ((r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD)))
      {
        r.origin = std::string("already mapped: ") + n;
        break;
      }
    if (!r.lib)
      if (FILE* f = std::fopen("/proc/self/maps", "r"))  // (a copy mapped under a path dlopen's search does not know)
      {
        char line[4096];
        while (!r.lib && std::fgets(line, sizeof(line), f))
          if (const char* hit = std::strstr(line, "librccl.so"))
          {
            const char* path = std::strchr(line, '/');
            if (!path || path > hit)
              continue;
            std::string pth(path);
            while (!pth.empty() && (pth.back() == '\n' || pth.back() == ' '))
              pth.pop_back();
            if ((r.lib = dlopen(pth.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD)))
              r.origin = "already mapped: " + pth;
          }
        std::fclose(f);
      }
    const char* names[] = { env, "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so" };
    for (const char* n : names)
      if (!r.lib && n && (r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)))
        r.origin = std::string("loaded: ") + n;
    if (!r.lib)
    {
      r.err = "RCCL not found (librccl.so.1; set AGH_RCCL_LIB)";
      return;
    }
    if (std::getenv("AGH_VERBOSE"))
      std::fprintf(stderr, "agile_grasp_amd: RCCL %s\n", r.origin.c_str());
    r.GetUniqueId = (decltype(r.GetUniqueId)) dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank)) dlsym(r.lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy)) dlsym(r.lib, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather)) dlsym(r.lib, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString)) dlsym(r.lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString)
      r.err = "RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather";
  });
  return &r;
}

// n contexts of one process exchanging through device copies (agh_comm_init_local)
// How long a rank of the in-process communicator waits for its peers before it declares the group dead: 600 s, or
// AGH_LOCAL_BARRIER_TIMEOUT_S (a legitimately slow rank -- the first call's 134 MB pool allocation, an all-points antipodal pass on a
// big cloud, a debugger -- must not break the group; 0 = wait for ever).  A group that timed out or was aborted stays dead: every
// later collective returns AGH_ERR_STATE at once ("communicator unusable", distinct from a HIP failure).
inline long local_barrier_timeout_s()
{
  static const long t = [] {
    const char* e = std::getenv("AGH_LOCAL_BARRIER_TIMEOUT_S");
    char* end = nullptr;
    const long v = e ? std::strtol(e, &end, 10) : 600;
    return (e && end != e && v >= 0) ? v : 600;
  }();
  return t;
}
struct LocalGroup
{
  int n = 0;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  bool failed = false;  // a rank left a collective call with an error: the others must not wait for it
  std::vector<const void*> send;
  // false: the group is broken (a rank failed between two collectives); the caller reports an error instead of waiting
  bool barrier()
  {
    std::unique_lock<std::mutex> lk(m);
    if (failed)
      return false;
    const uint64_t g = generation;
    if (++arrived == n)
    {
      arrived = 0;
      generation++;
      cv.notify_all();
    }
    else
    {
      const long tmo = local_barrier_timeout_s();
      auto done = [&] { return generation != g || failed; };
      bool came = true;
      if (tmo == 0)
        cv.wait(lk, done);
      else
        came = cv.wait_for(lk, std::chrono::seconds(tmo), done);
      if (!came)
      {
        // a rank never came (it returned before the collective on an error its peers did not share -- a broken precondition):
        // fail every waiter instead of hanging the process; the group is unusable afterwards, whoever arrives later
        // (`arrived` is cleared so that a late rank cannot complete a generation of a dead group)
        failed = true;
        arrived = 0;
        cv.notify_all();
      }
    }
    return !failed;
  }
  void abort()
  {
    std::lock_guard<std::mutex> lk(m);
    failed = true;
    arrived = 0;
    cv.notify_all();
  }
};
}  // namespace

struct Comm
{
  int rank = 0, n_ranks = 1;
  ncclComm_t nccl = nullptr;
  std::shared_ptr<LocalGroup> local;
};

void comm_release(Ctx* c)
{
  if (!c->comm)
    return;
  if (c->comm->nccl && rccl()->CommDestroy)
    (void) rccl()->CommDestroy(c->comm->nccl);
  delete c->comm;
  c->comm = nullptr;
}

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

// In-place all-gather: rank r's contribution already sits at buf + r * bytes.
int all_gather(Ctx* c, void* buf, size_t bytes, hipStream_t st)
{
  Comm* cm = c->comm;
  if (bytes == 0 || (cm->n_ranks == 1 && !cm->nccl))
    return AGH_OK;
  if (cm->nccl)
  {
    const ncclResult_t r = rccl()->AllGather((const char*) buf + (size_t) cm->rank * bytes, buf, bytes, ncclChar, cm->nccl, st);
    if (r != ncclSuccess)
    {
      c->err = std::string("ncclAllGather: ") + rccl()->GetErrorString(r);
      return AGH_ERR_HIP;
    }
    return AGH_OK;
  }
  LocalGroup* g = cm->local.get();
  HIPCHK(c, hipStreamSynchronize(st));  // my segment is complete
  {
    std::lock_guard<std::mutex> lk(g->m);
    g->send[(size_t) cm->rank] = (const char*) buf + (size_t) cm->rank * bytes;
  }
  const char* broken = "a rank of the in-process communicator left the collective with an error; the communicator is unusable";
  if (!g->barrier())
  {
    c->err = broken;
    return AGH_ERR_STATE;
  }
  for (int q = 0; q < cm->n_ranks; q++)
    if (q != cm->rank)
      HIPCHK(c, hipMemcpyAsync((char*) buf + (size_t) q * bytes, g->send[(size_t) q], bytes, hipMemcpyDeviceToDevice, st));
  HIPCHK(c, hipStreamSynchronize(st));
  if (!g->barrier())  // nobody reuses a segment before everyone has copied it
  {
    c->err = broken;
    return AGH_ERR_STATE;
  }
  return AGH_OK;
}

constexpr int kHeaderBytes = 160;  // one record's size: keeps the records of a segment 16-byte aligned

__host__ __device__ inline int64_t shard_lo(int64_t n, int64_t r, int64_t G)
{
  return (n * r) / G;
}

// header of an empty slice: no hypotheses, and the rank's findings (shard_header_word) as k_compact_* would write them
// (extra: kHdrRankFailed / kHdrRankNoCloud -- this rank took part in the call's collectives without searching)
__global__ void k_shard_empty_header(int64_t* __restrict__ hdr, const int32_t* __restrict__ flags, int big, int extra)
{
  hdr[0] = 0;
  hdr[1] = shard_header_word(flags[0], big) | extra;
}
// a rank whose classification failed on its own: its records travel as they are, its header says so
__global__ void k_shard_flag_header(int64_t* __restrict__ hdr, int extra)
{
  hdr[1] |= extra;
}

// RAND50: draws my slice consumes (50 per neighbourhood of more than 50 points, quadric.cpp:177-193)
__global__ void k_shard_draw_count(const int32_t* __restrict__ nt, int S, int64_t* __restrict__ out)
{
  int cnt = 0;
  for (int i = threadIdx.x; i < S; i += 64)
    cnt += nt[i] > 50 ? 50 : 0;
  for (int o = 32; o > 0; o >>= 1)
    cnt += __shfl_xor(cnt, o);
  if (threadIdx.x == 0)
    *out = cnt;
}
__global__ void k_shard_draw_base(const int64_t* __restrict__ counts, int rank, int32_t* __restrict__ total_io)
{
  int64_t b = 0;
  for (int q = 0; q < rank; q++)
    b += counts[q];
  *total_io += (int32_t) b;
}

// the slice's sample normals (valid frames only) into my segment of the normals exchange buffer: 4 doubles per sample
__global__ void k_shard_pack_normals(const agh_frame* __restrict__ frames, int S, double* __restrict__ seg)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S)
    return;
  const agh_frame& f = frames[i];
  seg[4 * i + 0] = f.normal[0];
  seg[4 * i + 1] = f.normal[1];
  seg[4 * i + 2] = f.normal[2];
  seg[4 * i + 3] = f.valid ? 1.0 : 0.0;
}
// cloud_normals_.col(indices[i]) = normal (hand_search.cpp:102) for every sample of every rank
__global__ void k_shard_scatter_normals(const double* __restrict__ nbuf, int64_t seg_doubles, const int32_t* __restrict__ samples,
  int64_t S, int G, double* __restrict__ normals, int n_points)
{
  const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S)
    return;
  int q = (int) ((i * G) / S);  // owner of sample i: the q with lo(q) <= i < lo(q + 1)
  while (q + 1 < G && shard_lo(S, q + 1, G) <= i)
    q++;
  while (q > 0 && shard_lo(S, q, G) > i)
    q--;
  const double* v = nbuf + (int64_t) q * seg_doubles + 4 * (i - shard_lo(S, q, G));
  const int p = samples[i];
  if (v[3] != 0.0 && p >= 0 && p < n_points)
  {
    normals[3 * (int64_t) p + 0] = v[0];
    normals[3 * (int64_t) p + 1] = v[1];
    normals[3 * (int64_t) p + 2] = v[2];
  }
}

// Concatenate the ranks' segments in rank order: sample positions become positions in the full list, every record gets
// this call's stamp, *n_out the total.  flags: 2 = the caller's buffer is too small, 16 = a rank found more hypotheses
// than its segment holds (the segment header carries the true count).
__global__ __launch_bounds__(256) void k_shard_merge(const uint8_t* __restrict__ xbuf, int64_t seg_bytes, int64_t seg_records,
  int G, int64_t S, agh_hypothesis* __restrict__ out, int64_t cap, int64_t* __restrict__ n_out, uint8_t* __restrict__ keep,
  int32_t* __restrict__ flags, int32_t epoch)
{
  __shared__ int64_t off[65];
  __shared__ int over;
  if (threadIdx.x == 0)
  {
    int64_t o = 0;
    int ov = 0, need = 0;
    for (int q = 0; q < G; q++)
    {
      int64_t cnt = *reinterpret_cast<const int64_t*>(xbuf + (int64_t) q * seg_bytes);
      need |= (int) (reinterpret_cast<const int64_t*>(xbuf + (int64_t) q * seg_bytes)[1] & (29 | kHdrRankFailed | kHdrRankNoCloud));  // shard_header_word
      if (cnt > seg_records)
      {
        ov = 1;
        cnt = seg_records;
      }
      off[q] = o;
      o += cnt;
    }
    off[G] = o;
    over = ov;
    if (blockIdx.x == 0)
    {
      *n_out = o;
      if (ov)
        atomicOr(&flags[0], 16);
      // what the ranks found, the same word on every rank (agh_internal.h: kFlagShard*)
      atomicOr(&flags[0], kFlagSharded | ((need & 1) ? kFlagShardRetry : 0) | ((need & 8) ? kFlagShardRetryHuge : 0) |
                            ((need & 16) ? kFlagShardHard : 0) | (need & 4) | ((need & kHdrRankFailed) ? kFlagShardPeerFailed : 0) |
                            ((need & kHdrRankNoCloud) ? kFlagShardPeerNoCloud : 0));
      if (o > cap)
        atomicOr(&flags[0], 2);
    }
  }
  __syncthreads();
  const int64_t total = off[G];
  // 10 threads per record, 16 bytes each
  for (int64_t t = (int64_t) blockIdx.x * 250 + threadIdx.x; threadIdx.x < 250 && t < total * 10; t += (int64_t) gridDim.x * 250)
  {
    const int64_t h = t / 10;
    const int part = (int) (t % 10);
    if (h >= cap)
      break;
    int q = 0;
    while (q + 1 < G && off[q + 1] <= h)
      q++;
    const uint8_t* src = xbuf + (int64_t) q * seg_bytes + kHeaderBytes + (h - off[q]) * (int64_t) sizeof(agh_hypothesis);
    uint4 v = reinterpret_cast<const uint4*>(src)[part];
    if (part == 8)  // bytes 128..143: sample, orientation, cam_source, n_in_box
      v.x = (unsigned) ((int) v.x + (int) shard_lo(S, q, G));
    if (part == 9)  // bytes 144..159: flags, finger_index, depth_index, epoch
    {
      if (keep)
        keep[h] = (uint8_t) ((v.x >> 16) & 0xffu);  // svm_keep
      v.w = (unsigned) epoch;
    }
    reinterpret_cast<uint4*>(out + h)[part] = v;
  }
}

template <typename T>
int grow(Ctx* c, T** p, int64_t* have, int64_t want)
{
  if (want <= *have && *p)
    return AGH_OK;
  if (*p)
    (void) hipFree(*p);
  *p = nullptr;
  *have = 0;
  if (hipMalloc((void**) p, (size_t) std::max<int64_t>(want, 1) * sizeof(T)) != hipSuccess)
  {
    c->err = "out of device memory for the shard exchange buffers";
    return AGH_ERR_HIP;
  }
  *have = want;
  return AGH_OK;
}

int exchange_and_merge(Ctx* c, uint8_t* d_keep, hipStream_t st)
{
  Comm* cm = c->comm;
  int rc = all_gather(c, c->d_xbuf, (size_t) c->shard_seg_bytes, st);
  if (rc != AGH_OK)
    return rc;
  timing_mark(c, "shard_allgather", st);
  hipLaunchKernelGGL(k_shard_merge, dim3(256), dim3(256), 0, st, (const uint8_t*) c->d_xbuf, c->shard_seg_bytes, c->shard_seg_records,
    cm->n_ranks, c->shard_S, c->shard_out, c->shard_cap, c->shard_nout, d_keep, c->d_flags, c->epoch);
  return hipGetLastError() == hipSuccess ? AGH_OK : AGH_ERR_HIP;
}

// One word per rank, all-gathered and read back: does EVERY rank say yes?  For decisions that must be the same on every rank and
// that only some ranks can make (an allocation that failed here): 8 bytes per rank and a host round trip, so only on the calls
// that grow an exchange buffer (the first of a size) and in the offline all-points pass.  `word` travels too (the cloud's size).
int shard_agree(Ctx* c, hipStream_t st, bool mine_ok, int64_t word, bool* all_ok, int64_t* words_out)
{
  Comm* cm = c->comm;
  const int G = cm->n_ranks, r = cm->rank;
  const int64_t mine = (word << 1) | (mine_ok ? 1 : 0);
  int64_t all[64];
  HIPCHK(c, hipMemcpyAsync(c->d_xcnt + 64 + r, &mine, sizeof(int64_t), hipMemcpyHostToDevice, st));
  const int rc = all_gather(c, c->d_xcnt + 64, sizeof(int64_t), st);
  if (rc != AGH_OK)
    return rc;
  HIPCHK(c, hipMemcpyAsync(all, c->d_xcnt + 64, sizeof(int64_t) * (size_t) G, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  *all_ok = true;
  for (int q = 0; q < G; q++)
  {
    *all_ok = *all_ok && (all[q] & 1);
    if (words_out)
      words_out[q] = all[q] >> 1;
  }
  return AGH_OK;
}

// Every rank of a NEW communicator starts without exchange buffers, so that "does this call grow one" -- which decides whether the
// ranks meet in shard_agree first -- is the same question on every rank, whatever the contexts did before.
void drop_exchange_buffers(Ctx* c)
{
  if (c->d_xbuf)
    (void) hipFree(c->d_xbuf);
  if (c->d_nbuf)
    (void) hipFree(c->d_nbuf);
  c->d_xbuf = nullptr;
  c->d_nbuf = nullptr;
  c->xbuf_bytes = 0;
  c->nbuf_doubles = 0;
  c->shard_out = nullptr;
}

// agh_comm_inject_fault (testing aid): does the site fail now?  One shot per set bit.
bool injected(Ctx* c, int site)
{
  if (!(c->shard_inject & site))
    return false;
  c->shard_inject &= ~site;
  return true;
}
}  // namespace

}  // namespace agh

using namespace agh;

extern "C" {

void agh_shard_slice(int64_t n, int32_t rank, int32_t n_ranks, int64_t* lo, int64_t* hi)
{
  if (n_ranks < 1)
    n_ranks = 1;
  if (lo)
    *lo = shard_lo(n, rank, n_ranks);
  if (hi)
    *hi = shard_lo(n, (int64_t) rank + 1, n_ranks);
}

const char* agh_comm_rccl_origin(void)
{
  Rccl* r = rccl();
  return r->lib ? r->origin.c_str() : r->err.c_str();
}

int agh_comm_unique_id(uint8_t id[AGH_COMM_ID_BYTES])
{
  static_assert(AGH_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
  if (!id)
    return AGH_ERR_INVALID_ARGUMENT;
  Rccl* r = rccl();
  if (!r->err.empty())
    return AGH_ERR_STATE;
  ncclUniqueId u;
  if (r->GetUniqueId(&u) != ncclSuccess)
    return AGH_ERR_HIP;
  std::memcpy(id, u.internal, AGH_COMM_ID_BYTES);
  return AGH_OK;
}

// What every rank of a communicator must agree on: the parameters that decide WHICH collectives a call issues
// (normals_mode: the RAND50 draw-count exchange) and what the shared result means (geometry, radii, camera origins, seed).
// device and profile are per rank.  FNV-1a over the fields' bytes.
static uint64_t params_fingerprint(const agh_params& p)
{
  uint64_t h = 1469598103934665603ull;
  auto eat = [&](const void* v, size_t n) {
    const unsigned char* b = (const unsigned char*) v;
    for (size_t i = 0; i < n; i++)
      h = (h ^ b[i]) * 1099511628211ull;
  };
  eat(&p.finger_width, sizeof(double));
  eat(&p.hand_outer_diameter, sizeof(double));
  eat(&p.hand_depth, sizeof(double));
  eat(&p.hand_height, sizeof(double));
  eat(&p.init_bite, sizeof(double));
  eat(&p.nn_radius_taubin, sizeof(double));
  eat(&p.nn_radius_hands, sizeof(double));
  eat(&p.nn_radius_normals, sizeof(double));
  eat(&p.cam_origin[0][0], sizeof(double) * 6);
  eat(&p.normals_mode, sizeof(p.normals_mode));
  eat(&p.rand_seed, sizeof(p.rand_seed));
  return h;
}
static const char* kParamsDiffer = "the contexts of a communicator must be created with the same agh_params (hand geometry, radii, "
                                   "camera origins, normals_mode, rand_seed): ranks that disagree issue different collectives";

int agh_comm_init(agh_ctx* ctx, int32_t rank, int32_t n_ranks, const uint8_t id[AGH_COMM_ID_BYTES])
{
  if (!ctx || !id)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (n_ranks < 1 || n_ranks > 64 || rank < 0 || rank >= n_ranks)
  {
    c->err = "agh_comm_init: need 0 <= rank < n_ranks <= 64";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  if (c->comm)
  {
    c->err = "agh_comm_init: the context already belongs to a communicator";
    return AGH_ERR_STATE;
  }
  Rccl* r = rccl();
  if (!r->err.empty())
  {
    c->err = r->err;
    return AGH_ERR_STATE;
  }
  HIPCHK(c, hipSetDevice(c->device));
  ncclUniqueId u;
  std::memcpy(u.internal, id, AGH_COMM_ID_BYTES);
  ncclComm_t comm = nullptr;
  const ncclResult_t res = r->CommInitRank(&comm, n_ranks, u, rank);
  if (res != ncclSuccess)
  {
    c->err = std::string("ncclCommInitRank: ") + r->GetErrorString(res);
    return AGH_ERR_HIP;
  }
  drop_exchange_buffers(c);
  c->comm = new Comm();
  c->comm->rank = rank;
  c->comm->n_ranks = n_ranks;
  c->comm->nccl = comm;
  // the first collective of the communicator: every rank's parameter fingerprint (every rank calls agh_comm_init, so nobody
  // waits alone; all ranks see the same words and accept or refuse together)
  {
    int rc = AGH_OK;
    if (!c->d_xcnt)
    {
      int64_t have = 0;
      rc = grow(c, &c->d_xcnt, &have, 128);
    }
    uint64_t mine = params_fingerprint(c->p), all[64];
    if (rc == AGH_OK && hipMemcpyAsync(c->d_xcnt + 64 + rank, &mine, sizeof(mine), hipMemcpyHostToDevice, c->stream) != hipSuccess)
      rc = AGH_ERR_HIP;
    if (rc == AGH_OK)
      rc = all_gather(c, c->d_xcnt + 64, sizeof(int64_t), c->stream);
    if (rc == AGH_OK && (hipMemcpyAsync(all, c->d_xcnt + 64, sizeof(uint64_t) * (size_t) n_ranks, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                         hipStreamSynchronize(c->stream) != hipSuccess))
      rc = AGH_ERR_HIP;
    if (rc == AGH_OK)
      for (int q = 0; q < n_ranks; q++)
        if (all[q] != all[0])
        {
          c->err = kParamsDiffer;
          rc = AGH_ERR_INVALID_ARGUMENT;
          break;
        }
    if (rc != AGH_OK)
    {
      if (rc == AGH_ERR_HIP)
        c->err = "agh_comm_init: the parameter check (first all-gather of the communicator) failed";
      comm_release(c);
      return rc;
    }
  }
  return AGH_OK;
}

int agh_comm_init_local(agh_ctx* const* ctxs, int32_t n_ranks)
{
  if (!ctxs || n_ranks < 1 || n_ranks > 64)
    return AGH_ERR_INVALID_ARGUMENT;
  for (int q = 0; q < n_ranks; q++)
    if (!ctxs[q] || ctxs[q]->c.comm)
      return AGH_ERR_STATE;
  for (int q = 1; q < n_ranks; q++)
    if (params_fingerprint(ctxs[q]->c.p) != params_fingerprint(ctxs[0]->c.p))
    {
      for (int k = 0; k < n_ranks; k++)
        ctxs[k]->c.err = kParamsDiffer;
      return AGH_ERR_INVALID_ARGUMENT;
    }
  for (int q = 0; q < n_ranks; q++)  // (the words the ranks agree through: allocated here, where a failure fails the init alike)
    if (!ctxs[q]->c.d_xcnt)
    {
      int64_t have = 0;
      if (hipSetDevice(ctxs[q]->c.device) != hipSuccess || grow(&ctxs[q]->c, &ctxs[q]->c.d_xcnt, &have, 128) != AGH_OK)
        return AGH_ERR_HIP;
    }
  std::shared_ptr<LocalGroup> g(new LocalGroup());
  g->n = n_ranks;
  g->send.assign((size_t) n_ranks, nullptr);
  for (int q = 0; q < n_ranks; q++)
  {
    Comm* cm = new Comm();
    cm->rank = q;
    cm->n_ranks = n_ranks;
    cm->local = g;
    drop_exchange_buffers(&ctxs[q]->c);
    ctxs[q]->c.comm = cm;
  }
  return AGH_OK;
}

// Testing aid: the next time the context passes one of these sites of a sharded call it fails there ON ITS OWN, as an
// out-of-memory or a launch error would (one shot per bit): 1 = the per-call buffers (device variant), 2 = the Taubin launch,
// 4 = the exchange buffer's growth, 8 = the HOG / SVM launch of agh_classify_sharded, 16 = the per-call buffers (host variant).
int agh_comm_inject_fault(agh_ctx* ctx, int32_t sites)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  ctx->c.shard_inject = sites;
  return AGH_OK;
}

int agh_comm_destroy(agh_ctx* ctx)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  (void) hipSetDevice(ctx->c.device);
  (void) hipDeviceSynchronize();
  comm_release(&ctx->c);
  return AGH_OK;
}

int agh_comm_set_segment_records(agh_ctx* ctx, int64_t records)
{
  if (!ctx || records < 0)
    return AGH_ERR_INVALID_ARGUMENT;
  ctx->c.shard_seg_override = records;
  ctx->c.shard_full_exchange = false;
  return AGH_OK;
}

int agh_comm_last_count(const agh_ctx* ctx, int64_t* n_hyp)
{
  if (!ctx || !n_hyp)
    return AGH_ERR_INVALID_ARGUMENT;
  if (!ctx->c.comm || !ctx->c.shard_out)
    return AGH_ERR_STATE;
  *n_hyp = ctx->c.shard_last_n;
  return AGH_OK;
}

int agh_comm_last_exchange(const agh_ctx* ctx, int64_t* segment_bytes, int32_t* n_ranks, int32_t* via_rccl)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  if (!ctx->c.comm || !ctx->c.shard_out)
    return AGH_ERR_STATE;
  if (segment_bytes)
    *segment_bytes = ctx->c.shard_seg_bytes;
  if (n_ranks)
    *n_ranks = ctx->c.comm->n_ranks;
  if (via_rccl)
    *via_rccl = ctx->c.comm->nccl ? 1 : 0;
  return AGH_OK;
}

int agh_comm_rank(const agh_ctx* ctx, int32_t* rank, int32_t* n_ranks)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  if (rank)
    *rank = ctx->c.comm ? ctx->c.comm->rank : 0;
  if (n_ranks)
    *n_ranks = ctx->c.comm ? ctx->c.comm->n_ranks : 1;
  return AGH_OK;
}

// A rank that fails ON ITS OWN (out of memory, a launch error) after the argument checks has left -- or will leave -- the others
// inside a collective: release them with an error instead of letting them wait for ever.  (In-process communicator; across
// processes RCCL's own watchdog applies.)  Errors every rank returns alike BEFORE any collective (bad arguments, no cloud, no
// communicator, an unsupported mode: everything the *_impl functions return while *entered is still false) leave the
// communicator usable.  After an abort the communicator stays unusable, like an aborted ncclComm: agh_comm_destroy and a
// new agh_comm_init_local.
static void shard_release_peers(agh_ctx* ctx, int rc, bool entered)
{
  if (rc != AGH_OK && entered && ctx && ctx->c.comm && ctx->c.comm->local)
    ctx->c.comm->local->abort();
}

// `pre_rc`: a failure of THIS rank that the caller (the host variant) already knows -- it could not size its buffers, it holds no
// cloud.  The rank still takes part: see "degraded" below.
static int find_hands_sharded_device_impl(agh_ctx* ctx, const int32_t* d_sample_idx, int64_t n_samples, int calculates_antipodal,
  agh_hypothesis* d_out, int64_t cap, int64_t* d_n_out, void* hip_stream, bool* entered, int pre_rc, const char* pre_err);

int agh_find_hands_sharded_device(agh_ctx* ctx, const int32_t* d_sample_idx, int64_t n_samples, int calculates_antipodal,
  agh_hypothesis* d_out, int64_t cap, int64_t* d_n_out, void* hip_stream)
{
  bool entered = false;
  const int rc = find_hands_sharded_device_impl(ctx, d_sample_idx, n_samples, calculates_antipodal, d_out, cap, d_n_out, hip_stream,
    &entered, AGH_OK, nullptr);
  shard_release_peers(ctx, rc, entered);
  return rc;
}

// NO RANK LEAVES ALONE (round 6).  Once the argument checks (the same on every rank) are passed, a rank reaches every collective
// of the call whatever happens to it:
//  * a rank that cannot do its share -- no cloud (a caller's bug), its per-call buffers could not be allocated, a kernel launch
//    failed -- is DEGRADED: it skips its kernels, contributes an empty segment whose header says so (kHdrRankFailed /
//    kHdrRankNoCloud) and issues every all-gather of the call with the agreed byte counts; the merge kernel turns the header bit
//    into a flag every rank reads, so every rank returns AGH_ERR_STATE / AGH_ERR_NO_CLOUD for this call -- the failing rank at
//    once, with its own error text -- and the communicator stays usable;
//  * the buffers the collectives themselves need (the exchange buffer, and in the offline all-points pass the normals buffers) are
//    grown FIRST and the outcome is agreed on (shard_agree: 8 bytes per rank and a host round trip -- only on a call that grows
//    one, which the ranks decide alike because they made the same sharded calls before, and in the all-points pass, which
//    exchanges the cloud sizes that way anyway): a rank that is out of memory makes every rank return AGH_ERR_HIP together.
// What is left outside: a HIP runtime or RCCL call that fails INSIDE a collective (the device or the fabric is gone).
static int find_hands_sharded_device_impl(agh_ctx* ctx, const int32_t* d_sample_idx, int64_t n_samples, int calculates_antipodal,
  agh_hypothesis* d_out, int64_t cap, int64_t* d_n_out, void* hip_stream, bool* entered, int pre_rc, const char* pre_err)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (!c->comm)
  {
    c->err = "agh_find_hands_sharded: no communicator (agh_comm_init)";
    return AGH_ERR_STATE;
  }
  if (n_samples < 0 || n_samples > (1 << 24) || (n_samples > 0 && (!d_sample_idx || !d_out)) || !d_n_out || cap < 0)
  {
    c->err = "agh_find_hands_sharded: bad arguments";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  const bool rand_mode = c->p.normals_mode == AGH_NORMALS_RAND50;
  if (rand_mode && calculates_antipodal && c->comm->n_ranks > 1)
  {
    // the all-points pass would need the draw counts of every earlier point range before its first frame kernel
    c->err = "agh_find_hands_sharded: calculates_antipodal with AGH_NORMALS_RAND50 is not sharded (the rand() stream runs "
             "through all N points in order); use deterministic normals or one GPU for this offline pass";
    return AGH_ERR_STATE;
  }
  *entered = true;  // from here on a failure is this rank's own
  c->shard_symmetric_error = false;
  // the first failure of this rank's own: from then on it is degraded (no kernels of its own, every collective still issued)
  int local_rc = pre_rc;
  std::string local_err = pre_err ? pre_err : "";
  int hdr_extra = pre_rc != AGH_OK ? kHdrRankFailed : 0;
  auto degrade = [&](int rc_, const std::string& what, int extra) {
    if (local_rc == AGH_OK)
    {
      local_rc = rc_;
      local_err = what;
      hdr_extra = extra;
    }
  };
  if (!c->has_cloud)  // a caller's bug on this rank: it takes part like a rank with an empty cloud, and every rank hears of it
    degrade(AGH_ERR_NO_CLOUD, "agh_find_hands_sharded: no cloud set", kHdrRankNoCloud);
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t st = hip_stream ? (hipStream_t) hip_stream : c->stream;
  HIPCHK(c, order_after_cloud(c, st));
  const int G = c->comm->n_ranks, r = c->comm->rank;
  const int64_t S = n_samples;
  const int64_t n_cloud = c->has_cloud ? c->n : 0;
  // (a rank whose cloud is EMPTY -- possible when every rank searches a cloud of its own -- takes part with an empty slice: it
  // must still join every collective of the call, or the others wait for it for ever)
  const int64_t lo = shard_lo(S, r, G), hi = shard_lo(S, r + 1, G);
  int64_t Sr = n_cloud == 0 ? 0 : hi - lo;
  const int64_t Smax = (S + G - 1) / G;
  if (Smax > 65536)  // the same on every rank: slices this long launch every capacity class from the start
    c->big_classes = true;
  int64_t seg_records = std::min<int64_t>(8 * Smax, std::max<int64_t>(2 * Smax, 1024));
  if (c->shard_seg_override > 0)
    seg_records = std::min<int64_t>(8 * Smax, c->shard_seg_override);
  if (c->shard_full_exchange)
    seg_records = 8 * Smax;
  const int64_t seg_bytes = kHeaderBytes + seg_records * (int64_t) sizeof(agh_hypothesis);
  const int64_t pcnt = (n_cloud + G - 1) / G;  // points per rank of the all-points pass
  int rc;
  // ---- the rank's own per-call buffers: a failure degrades it ----
  if (local_rc == AGH_OK)
  {
    rc = injected(c, 1) ? AGH_ERR_HIP : ensure_call_buffers(c, std::max<int64_t>(Smax, calculates_antipodal ? std::min<int64_t>(n_cloud, kNormalsChunk) : 0));
    if (rc != AGH_OK)
      degrade(rc, rc == AGH_ERR_HIP && c->err.empty() ? "out of device memory for the per-call buffers" : c->err, kHdrRankFailed);
  }
  // (a communicator of one may run the all-points pass in the production mode: its rand() stream runs through all N
  // points before the samples, exactly as in agh_find_hands_device)
  if (local_rc == AGH_OK && rand_mode && (rc = ensure_draws(c, 50 * (S + (calculates_antipodal ? n_cloud : 0)), st)) != AGH_OK)
    degrade(rc, c->err, kHdrRankFailed);
  // ---- the buffers the collectives need: grown first, the outcome agreed on ----
  const bool grow_x = (int64_t) G * seg_bytes > c->xbuf_bytes || !c->d_xbuf;  // (the same on every rank: same calls before)
  bool grown = true;
  if (grow_x)
    grown = !injected(c, 4) && grow(c, &c->d_xbuf, &c->xbuf_bytes, (int64_t) G * seg_bytes) == AGH_OK;
  const int64_t seg_d = 4 * Smax;  // doubles per rank of the sample normals' exchange
  if (calculates_antipodal && G > 1)
  {
    // hand_search.cpp:13-26 sharded by point range; the buffer is padded to G equal ranges for the in-place all-gather
    if (grown && ((int64_t) G * pcnt > c->normals_cap || !c->d_normals))  // (normals_cap counts points)
    {
      if (c->d_normals)
        (void) hipFree(c->d_normals);
      c->d_normals = nullptr;
      c->normals_cap = 0;
      if (hipMalloc((void**) &c->d_normals, sizeof(double) * 3 * (size_t) std::max<int64_t>(G * pcnt, 1)) == hipSuccess)
        c->normals_cap = (int64_t) G * pcnt;
      else
        grown = false;
    }
    if (grown)
      grown = grow(c, &c->d_nbuf, &c->nbuf_doubles, (int64_t) G * seg_d) == AGH_OK;
  }
  if (G > 1 && (grow_x || calculates_antipodal))
  {
    // The all-points pass is sharded by POINT range, so every rank must hold the same cloud here -- unlike the plain search,
    // where a cloud per rank is a supported mode.  A rank cannot see another rank's cloud: with different point counts the
    // byte counts of the normals' all-gather below would differ from rank to rank (a hang or silent garbage under RCCL).
    // The counts travel with the outcome of the allocations and every rank refuses alike.
    bool all_ok = false;
    int64_t sizes[64];
    if ((rc = shard_agree(c, st, grown, n_cloud, &all_ok, sizes)) != AGH_OK)
      return rc;
    *entered = false;  // (every rank saw the same words and returns here or nobody does: nobody is left in a collective)
    if (!all_ok)
    {
      // (every rank drops its exchange buffer: the next call's "does it grow" is then the same question on every rank again)
      if (c->d_xbuf)
        (void) hipFree(c->d_xbuf);
      c->d_xbuf = nullptr;
      c->xbuf_bytes = 0;
      c->err = grown ? "agh_find_hands_sharded: another rank of the communicator could not allocate its exchange buffers"
                     : "agh_find_hands_sharded: out of device memory for the exchange buffers (every rank returns with this call)";
      c->shard_symmetric_error = true;
      return AGH_ERR_HIP;
    }
    if (calculates_antipodal)
      for (int q = 0; q < G; q++)
        if (sizes[q] != sizes[0])
        {
          c->err = "agh_find_hands_sharded: calculates_antipodal shards the all-points pass by point range and needs the SAME "
                   "cloud on every rank (the ranks hold clouds of different sizes)";
          c->shard_symmetric_error = true;
          return AGH_ERR_STATE;
        }
    *entered = true;
  }
  else if (!grown)  // a communicator of one, or a rank alone with its memory: nobody is waiting
  {
    c->err = "out of device memory for the shard exchange buffers";
    *entered = false;
    return AGH_ERR_HIP;
  }
  if (calculates_antipodal && G == 1 && (pcnt > c->normals_cap || !c->d_normals))
  {
    if (c->d_normals)
      (void) hipFree(c->d_normals);
    c->d_normals = nullptr;
    c->normals_cap = 0;
    HIPCHK(c, hipMalloc((void**) &c->d_normals, sizeof(double) * 3 * (size_t) std::max<int64_t>(pcnt, 1)));
    c->normals_cap = pcnt;
  }
  timing_begin(c, st);
  c->zero_flags_pending = true;
  c->epoch = next_epoch();
  c->last_s = local_rc == AGH_OK ? Sr : 0;
  c->last_nout = -1;
  c->shard_seg_records = seg_records;
  c->shard_seg_bytes = seg_bytes;
  c->shard_S = S;
  c->shard_out = d_out;
  c->shard_cap = cap;
  c->shard_nout = d_n_out;
  c->shard_last_n = -1;
  uint8_t* my_seg = c->d_xbuf + (int64_t) r * seg_bytes;
  agh_hypothesis* my_out = reinterpret_cast<agh_hypothesis*>(my_seg + kHeaderBytes);
  int64_t* my_count = reinterpret_cast<int64_t*>(my_seg);
  // what agh_classify* / the introspection getters see afterwards is this rank's own part
  c->last_cap = seg_records;
  c->d_out_last = my_out;
  c->d_nout_last = my_count;
  if (S == 0)  // (the same on every rank: nobody starts a collective)
  {
    HIPCHK(c, hipMemsetAsync(c->d_flags, 0, 8 * sizeof(int32_t), st));
    c->zero_flags_pending = false;
    HIPCHK(c, hipMemsetAsync(d_n_out, 0, sizeof(int64_t), st));
    HIPCHK(c, hipMemsetAsync(c->d_xbuf, 0, (size_t) G * seg_bytes, st));
    c->last_s = 0;
    if (local_rc != AGH_OK)
    {
      c->err = local_err;
      *entered = false;
      return local_rc;
    }
    return AGH_OK;
  }
  const int32_t* my_idx = d_sample_idx + lo;
  auto ok = [&]() { return local_rc == AGH_OK; };
  if (calculates_antipodal)
  {
    HIPCHK(c, hipMemsetAsync(c->d_normals, 0, sizeof(double) * 3 * (size_t) (G * pcnt), st));
    const int64_t p0 = std::min<int64_t>(n_cloud, (int64_t) r * pcnt), p1 = std::min<int64_t>(n_cloud, (int64_t) (r + 1) * pcnt);
    if (ok() && (rc = normals_pass(c, p0, p1, st)) != AGH_OK)
      degrade(rc, "normals pass launch failed", kHdrRankFailed);
    if ((rc = all_gather(c, c->d_normals, sizeof(double) * 3 * (size_t) pcnt, st)) != AGH_OK)
      return rc;
    c->has_normals = true;
  }
  if (ok() && Sr > 0)
  {
    if ((rc = injected(c, 2) ? AGH_ERR_HIP : taubin_moments_eigen(c, my_idx, Sr, c->p.nn_radius_taubin, c->d_nt, st)) != AGH_OK)
      degrade(rc, "taubin launch failed", kHdrRankFailed);
  }
  if (c->zero_flags_pending)  // (no Taubin launch cleared them: an empty or a degraded slice)
  {
    HIPCHK(c, hipMemsetAsync(c->d_flags, 0, 8 * sizeof(int32_t), st));
    c->zero_flags_pending = false;
  }
  if (rand_mode && G > 1)
  {
    hipLaunchKernelGGL(k_shard_draw_count, dim3(1), dim3(64), 0, st, (const int32_t*) c->d_nt, ok() ? (int) Sr : 0, c->d_xcnt + r);
    if ((rc = all_gather(c, c->d_xcnt, sizeof(int64_t), st)) != AGH_OK)
      return rc;
    hipLaunchKernelGGL(k_shard_draw_base, dim3(1), dim3(1), 0, st, (const int64_t*) c->d_xcnt, r, c->d_flags + 2);
  }
  if (ok() && Sr > 0 && (rc = taubin_frame_stage(c, my_idx, Sr, c->p.nn_radius_taubin, c->d_frames, c->d_nt, calculates_antipodal != 0, st)) != AGH_OK)
    degrade(rc, "taubin launch failed", kHdrRankFailed);
  if (calculates_antipodal && G > 1)
  {
    if (ok() && Sr > 0)
      hipLaunchKernelGGL(k_shard_pack_normals, dim3((unsigned) ((Sr + 255) / 256)), dim3(256), 0, st, (const agh_frame*) c->d_frames,
        (int) Sr, c->d_nbuf + (int64_t) r * seg_d);
    else  // (nothing of this rank's: every "valid" word zero)
      HIPCHK(c, hipMemsetAsync(c->d_nbuf + (int64_t) r * seg_d, 0, sizeof(double) * (size_t) seg_d, st));
    if ((rc = all_gather(c, c->d_nbuf, sizeof(double) * (size_t) seg_d, st)) != AGH_OK)
      return rc;
    if (ok())  // (a degraded rank searches nothing: it needs nobody's normals -- and its sample list may not exist)
      hipLaunchKernelGGL(k_shard_scatter_normals, dim3((unsigned) ((S + 255) / 256)), dim3(256), 0, st, (const double*) c->d_nbuf, seg_d,
        d_sample_idx, S, G, c->d_normals, (int) n_cloud);
  }
  if (ok() && Sr > 0)
  {
    if ((rc = hand_sweep(c, my_idx, Sr, calculates_antipodal != 0, st)) != AGH_OK)
      degrade(rc, "hand sweep launch failed", kHdrRankFailed);
    // K4 straight into my segment of the exchange buffer; an overflow of the segment shows as count > seg_records
    else if ((rc = compact_hypotheses(c, Sr, my_out, seg_records, my_count, st, my_count + 1)) != AGH_OK)
      degrade(rc, "compaction launch failed", kHdrRankFailed);
  }
  if (!ok() || Sr == 0)
  {
    // an empty slice: count 0 -- and still the flag its share of the all-points pass may have raised (the other ranks must learn
    // of a capacity-class retry from EVERY rank, or this one would repeat the collective alone); a degraded rank: count 0 and
    // the word that makes the call fail on every rank
    hipLaunchKernelGGL(k_shard_empty_header, dim3(1), dim3(1), 0, st, my_count, (const int32_t*) c->d_flags, class_level(c), hdr_extra);
    HIPCHK(c, hipGetLastError());
  }
  if (!ok())
    c->shard_cap = 0;  // (the merged list is not this rank's to keep: its output buffer may not exist)
  if ((rc = exchange_and_merge(c, nullptr, st)) != AGH_OK)
    return rc;
  timing_mark(c, "shard_merge", st);
  if (!ok())
  {
    // this rank's own failure, reported at once; its peers read it from the merged flags -- every collective has been issued
    c->err = local_err;
    c->last_s = 0;
    *entered = false;
    c->shard_symmetric_error = true;
    return local_rc;
  }
  return AGH_OK;
}

static int classify_sharded_device_impl(agh_ctx* ctx, uint8_t* d_keep, void* hip_stream, bool* entered, int pre_rc, const char* pre_err);

int agh_classify_sharded_device(agh_ctx* ctx, uint8_t* d_keep, void* hip_stream)
{
  bool entered = false;
  const int rc = classify_sharded_device_impl(ctx, d_keep, hip_stream, &entered, AGH_OK, nullptr);
  shard_release_peers(ctx, rc, entered);
  return rc;
}

// (as the search: a rank that cannot classify -- no SVM loaded on it, a launch failure, no room for the labels -- still takes part
// in the exchange, with a header that makes the call fail on every rank)
static int classify_sharded_device_impl(agh_ctx* ctx, uint8_t* d_keep, void* hip_stream, bool* entered, int pre_rc, const char* pre_err)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (!c->comm || !c->shard_out || !c->d_xbuf)
  {
    c->err = "agh_classify_sharded: needs a preceding agh_find_hands_sharded";
    return AGH_ERR_STATE;
  }
  *entered = true;
  int local_rc = pre_rc;
  std::string local_err = pre_err ? pre_err : "";
  if (local_rc == AGH_OK && !c->has_svm)
  {
    local_rc = AGH_ERR_NO_SVM;
    local_err = "agh_classify_sharded: no SVM loaded";
  }
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t st = hip_stream ? (hipStream_t) hip_stream : c->stream;
  // K3 on my own hypotheses: their images are here, svm_keep lands in my segment's records
  int rc;
  if (local_rc == AGH_OK && (rc = injected(c, 8) ? AGH_ERR_HIP : hog_svm(c, std::min<int64_t>(c->last_s * 8, c->shard_seg_records), nullptr, st)) != AGH_OK)
  {
    local_rc = rc;
    local_err = "HOG / SVM launch failed";
  }
  if (local_rc != AGH_OK)
  {
    hipLaunchKernelGGL(k_shard_flag_header, dim3(1), dim3(1), 0, st, c->d_nout_last, kHdrRankFailed);
    HIPCHK(c, hipGetLastError());
  }
  if ((rc = exchange_and_merge(c, local_rc == AGH_OK ? d_keep : nullptr, st)) != AGH_OK)
    return rc;
  timing_mark(c, "shard_merge", st);
  if (local_rc != AGH_OK)
  {
    c->err = local_err;
    *entered = false;
    c->shard_symmetric_error = true;
    return local_rc;
  }
  return AGH_OK;
}

static int shard_flags(Ctx* c, hipStream_t st, int64_t* n)
{
  int32_t flags[8];
  HIPCHK(c, hipMemcpyAsync(flags, c->d_flags, sizeof(flags), hipMemcpyDeviceToHost, st));
  if (n)
    HIPCHK(c, hipMemcpyAsync(n, c->shard_nout, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  if (flags[0] & 16)
    return 16;
  // (every rank reads the same segment headers, so every rank takes the same branch here -- whatever its own big_classes was:
  // a rank that had the larger classes on reports a hard overflow, one that had not asks for the retry, and a retry anybody
  // asks for is taken by all)
  if (flags[0] & kFlagShardHard)
  {
    c->err = "the Taubin neighbourhoods of more than 4096 points of one launch hold more than 2^21 points in all";
    return AGH_ERR_CAPACITY;
  }
  if (flags[0] & (kFlagShardRetry | kFlagShardRetryHuge))
  {
    c->big_classes = true;
    if (flags[0] & kFlagShardRetryHuge)
      c->huge_classes = true;
    c->err = "a Taubin neighbourhood exceeds the capacity classes launched so far; the contexts of the communicator now launch the "
             "larger classes as well: repeat the call";
    return AGH_ERR_RETRY;
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
  if (flags[0] & 2)
  {
    c->err = "output buffer too small for the hypotheses found";
    return AGH_ERR_CAPACITY;
  }
  if (flags[0] & 4)
  {
    c->err = "a sample index is outside the cloud";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  return AGH_OK;
}

static int find_hands_sharded_host_impl(agh_ctx* ctx, const int32_t* sample_idx, int64_t n_samples, int calculates_antipodal,
  agh_hypothesis* out, int64_t cap, int64_t* n_out, bool* entered);

int agh_find_hands_sharded(agh_ctx* ctx, const int32_t* sample_idx, int64_t n_samples, int calculates_antipodal,
  agh_hypothesis* out, int64_t cap, int64_t* n_out)
{
  bool entered = false;
  const int rc = find_hands_sharded_host_impl(ctx, sample_idx, n_samples, calculates_antipodal, out, cap, n_out, &entered);
  shard_release_peers(ctx, rc, entered);  // (the buffer set-up below can fail on one rank alone, before the device variant runs)
  return rc;
}

static int find_hands_sharded_host_impl(agh_ctx* ctx, const int32_t* sample_idx, int64_t n_samples, int calculates_antipodal,
  agh_hypothesis* out, int64_t cap, int64_t* n_out, bool* entered)
{
  if (!ctx || !n_out)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  *n_out = 0;
  if (!c->comm)
  {
    c->err = "agh_find_hands_sharded: no communicator (agh_comm_init)";
    return AGH_ERR_STATE;
  }
  if (c->p.normals_mode == AGH_NORMALS_RAND50 && calculates_antipodal && c->comm->n_ranks > 1)
  {
    c->err = "agh_find_hands_sharded: calculates_antipodal with AGH_NORMALS_RAND50 is not sharded (the rand() stream runs "
             "through all N points in order); use deterministic normals or one GPU for this offline pass";
    return AGH_ERR_STATE;
  }
  if (n_samples < 0 || (n_samples > 0 && !sample_idx) || cap < 0 || (cap > 0 && !out))
  {
    c->err = "agh_find_hands_sharded: bad arguments";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  *entered = true;  // every rank passed the same checks on the same arguments: what fails from here on fails on this rank alone
  // ... and a rank it fails on does not leave: it hands its failure to the device variant (pre_rc), which takes part in every
  // collective with an empty, flagged segment -- so does a rank without a cloud (checked there)
  int pre_rc = AGH_OK;
  std::string pre_err;
  auto pre_fail = [&](int rc_, const std::string& what) {
    if (pre_rc == AGH_OK)
    {
      pre_rc = rc_;
      pre_err = what;
    }
  };
  // (The sample indices are NOT range-checked here.  A rank reads only its slice of the list -- when every rank searches a cloud
  // of its own the other slices index other clouds -- so a host-side check could only ever fail on ONE rank, which would then
  // leave before the collectives the others wait in.  The kernels validate every index they read
  // (kStatusBadIndex), the finding travels in the rank's segment header, and every rank returns AGH_ERR_INVALID_ARGUMENT
  // together after the exchange.)
  HIPCHK(c, hipSetDevice(c->device));
  const int64_t n_cloud = c->has_cloud ? c->n : 0;
  int rc = injected(c, 16) ? AGH_ERR_HIP : ensure_call_buffers(c, std::max<int64_t>(n_samples, calculates_antipodal ? std::min<int64_t>(n_cloud, kNormalsChunk) : 0));
  if (rc != AGH_OK)
    pre_fail(rc, c->err.empty() ? "out of device memory for the per-call buffers" : c->err);
  if (pre_rc == AGH_OK && (n_samples > c->idx_cap || !c->d_idx_own))
  {
    if (c->d_idx_own)
      (void) hipFree(c->d_idx_own);
    c->d_idx_own = nullptr;
    c->idx_cap = 0;
    if (hipMalloc((void**) &c->d_idx_own, sizeof(int32_t) * (size_t) std::max<int64_t>(n_samples, 1024)) == hipSuccess)
      c->idx_cap = std::max<int64_t>(n_samples, 1024);
    else
      pre_fail(AGH_ERR_HIP, "out of device memory for the sample list");
  }
  if (pre_rc == AGH_OK && n_samples > 0 &&
      hipMemcpyAsync(c->d_idx_own, sample_idx, sizeof(int32_t) * n_samples, hipMemcpyHostToDevice, c->stream) != hipSuccess)
    pre_fail(AGH_ERR_HIP, "the sample list could not be copied to the device");
  int64_t n = 0;
  for (int attempt = 0; attempt < 4; attempt++)  // (at most two repeats for the capacity classes, one for the segment size)
  {
    // (a retry may have switched a capacity class on that wants larger per-sample scratch: sized here, not under the device call)
    if (pre_rc == AGH_OK &&
        (rc = ensure_call_buffers(c, std::max<int64_t>(n_samples, calculates_antipodal ? std::min<int64_t>(n_cloud, kNormalsChunk) : 0))) != AGH_OK)
      pre_fail(rc, c->err);
    // (s_cap >= n_samples, so d_out_own holds the complete list: 8 slots per sample.  A rank that failed above hands over
    // pointers nobody reads: a degraded rank launches nothing on the sample list and keeps no list)
    const bool usable = pre_rc == AGH_OK;
    bool in_coll = false;
    rc = find_hands_sharded_device_impl(ctx, usable ? c->d_idx_own : reinterpret_cast<const int32_t*>(c->d_flags), n_samples,
      calculates_antipodal, usable ? c->d_out_own : reinterpret_cast<agh_hypothesis*>(c->d_flags), usable ? c->s_cap * 8 : 0,
      c->d_nout, c->stream, &in_coll, pre_rc, pre_err.c_str());
    if (rc != AGH_OK)
    {
      (void) hipStreamSynchronize(c->stream);
      *entered = in_coll && !c->shard_symmetric_error;
      return rc;
    }
    rc = shard_flags(c, c->stream, &n);
    // The collectives of this attempt have completed on this rank, and what shard_flags reports (other than a HIP failure of its
    // own) was derived from the gathered headers: the same on every rank.  Such errors -- a retry, a capacity limit, a bad
    // index, a caller's buffer that is too small -- leave the communicator usable (a caller that came back with
    // a larger buffer found the in-process communicator aborted).
    *entered = rc == AGH_ERR_HIP;
    if (rc == AGH_ERR_RETRY)
    {
      *entered = true;  // (the next attempt's collectives)
      continue;  // the larger capacity classes are on now, on every rank
    }
    if (rc != 16)
      break;
    *entered = true;
    // a rank found more than its segment holds: every rank sees the same headers, so every rank repeats with 8 per sample
    c->shard_full_exchange = true;
    rc = AGH_ERR_RETRY;
    c->err = "a rank found more hypotheses than its exchange segment holds";
  }
  if (rc != AGH_OK)
    return rc;
  *n_out = n;
  c->shard_last_n = n;
  if (n > cap)
  {
    c->err = "output buffer too small for the hypotheses found";
    return AGH_ERR_CAPACITY;
  }
  if (n > 0)
    HIPCHK(c, hipMemcpy(out, c->d_out_own, sizeof(agh_hypothesis) * n, hipMemcpyDeviceToHost));
  // (last_nout stays -1 for the merged list: the per-rank state the getters see is this rank's own part)
  int64_t mine = 0;
  HIPCHK(c, hipMemcpy(&mine, c->d_nout_last, sizeof(int64_t), hipMemcpyDeviceToHost));
  c->last_nout = std::min<int64_t>(mine, c->shard_seg_records);
  return AGH_OK;
}

static int classify_sharded_host_impl(agh_ctx* ctx, agh_hypothesis* out, uint8_t* keep, int64_t cap, int64_t* n_kept, bool* entered);

int agh_classify_sharded(agh_ctx* ctx, agh_hypothesis* out, uint8_t* keep, int64_t cap, int64_t* n_kept)
{
  bool entered = false;
  const int rc = classify_sharded_host_impl(ctx, out, keep, cap, n_kept, &entered);
  shard_release_peers(ctx, rc, entered);
  return rc;
}

static int classify_sharded_host_impl(agh_ctx* ctx, agh_hypothesis* out, uint8_t* keep, int64_t cap, int64_t* n_kept, bool* entered)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (n_kept)
    *n_kept = 0;
  if (!c->comm || !c->shard_out || c->shard_out != c->d_out_own)
  {
    c->err = "agh_classify_sharded: needs a preceding agh_find_hands_sharded (host variant)";
    return AGH_ERR_STATE;
  }
  *entered = true;
  HIPCHK(c, hipSetDevice(c->device));
  int pre_rc = AGH_OK;
  const char* pre_err = nullptr;
  const int64_t room = c->s_cap * 8;
  if (room > c->keep_cap)
  {
    if (c->d_keep)
      (void) hipFree(c->d_keep);
    if (c->d_svm_sums)
      (void) hipFree(c->d_svm_sums);
    c->d_keep = nullptr;
    c->d_svm_sums = nullptr;
    c->keep_cap = 0;
    if (hipMalloc((void**) &c->d_keep, (size_t) room) == hipSuccess &&
        hipMalloc((void**) &c->d_svm_sums, (size_t) room * sizeof(double)) == hipSuccess)
      c->keep_cap = room;
    else
    {
      pre_rc = AGH_ERR_HIP;  // (this rank still takes part in the exchange: agh_classify_sharded_device's comment)
      pre_err = "out of device memory for the labels";
    }
  }
  bool in_coll = false;
  int rc = classify_sharded_device_impl(ctx, c->d_keep, c->stream, &in_coll, pre_rc, pre_err);
  if (rc != AGH_OK)
  {
    (void) hipStreamSynchronize(c->stream);
    *entered = in_coll && !c->shard_symmetric_error;
    return rc;
  }
  int64_t n = 0;
  rc = shard_flags(c, c->stream, &n);
  *entered = rc == AGH_ERR_HIP;  // (the collective is over; what is reported from here on is the same on every rank)
  if (rc != AGH_OK)
    return rc == 16 ? AGH_ERR_CAPACITY : rc;
  if (n > cap)
  {
    c->err = "agh_classify_sharded: buffers too small";
    return AGH_ERR_CAPACITY;
  }
  if (n > 0 && out)
    HIPCHK(c, hipMemcpy(out, c->d_out_own, sizeof(agh_hypothesis) * n, hipMemcpyDeviceToHost));
  if (n > 0 && keep)
  {
    HIPCHK(c, hipMemcpy(keep, c->d_keep, (size_t) n, hipMemcpyDeviceToHost));
    int64_t k = 0;
    for (int64_t i = 0; i < n; i++)
      k += keep[i] ? 1 : 0;
    if (n_kept)
      *n_kept = k;
  }
  return AGH_OK;
}

}  // extern "C"
