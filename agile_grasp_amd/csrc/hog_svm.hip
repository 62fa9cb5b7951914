// hog_svm.hip -- K3: HOG-style descriptor of the 80x100 grasp image + linear SVM score, per hypothesis.
//
// Reference path: Learning::classify src/agile_grasp/learning.cpp:165-247 (OMP loop C) ->
// cv::HOGDescriptor(winSize 64x64).compute(image, winStride 32x32) (194-195, 220; OpenCV 2.4
// modules/objdetect/src/hog.cpp) -> CvSVM::predict (225; OpenCV 2.4 modules/ml/src/svm.cpp, linear kernel, one
// support vector).  The image itself (Learning::convertToImage, 320-365) is rasterised by the hand-sweep kernel.
//
// One 256-thread workgroup per hypothesis; the descriptor never leaves the CU.
//  * The image is binary, so after the sqrt gamma LUT a centred difference takes 3 values per axis: each pixel's
//    <magnitude*(1-a), magnitude*a, bin0, bin1> comes from a 9-entry table built on the host with OpenCV's own
//    float formulas (fastAtan2 polynomial included).
//  * Only the 77 distinct 16x16 blocks of the two windows are built (the windows share 3 block columns).  Each
//    (block, cell) pair owns its 9 bins and adds its pixels' votes in HOGCache's pixData order, skipping pixels
//    with zero gradient (adding +0.0f is exact), so every float sum has the reference's order.
//  * L2-Hys per block with the reference's sequential loops; the SVM dot product keeps OpenCV's shape: float
//    products summed four at a time in float, accumulated in double in index order by one lane.
#include <cstdlib>

#include "agh_internal.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

namespace agh
{

// ---- host: tables of HOGCache::init / computeGradient for this geometry ------------------------------------
static float fast_atan2_deg_host(float y, float x)
{
  const float sc = (float) (180 / M_PI);
  const float p1 = 0.9997878412794807f * sc, p3 = -0.3258083974640975f * sc;
  const float p5 = 0.1555786518463281f * sc, p7 = -0.04432655554792128f * sc;
  const float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay)
  {
    c = ay / (ax + (float) 2.2204460492503131e-16);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  else
  {
    c = ax / (ay + (float) 2.2204460492503131e-16);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0)
    a = 180.f - a;
  if (y < 0)
    a = 360.f - a;
  return a;
}

void hog_tables_host(HogTablesDev* t)
{
  std::memset(t, 0, sizeof(*t));
  const int nbins = 9;
  const float l255 = std::sqrt((float) 255);  // gamma LUT entry of a set pixel; a clear pixel maps to 0
  const float angleScale = (float) (nbins / M_PI);
  const float rad = (float) (M_PI / 180);
  for (int sx = -1; sx <= 1; sx++)
    for (int sy = -1; sy <= 1; sy++)
    {
      const float dx = sx * l255, dy = sy * l255;
      const float mag = std::sqrt(dx * dx + dy * dy);
      float angle = (float) (fast_atan2_deg_host(dy, dx) * rad);
      angle = angle * angleScale - 0.5f;
      int hidx = (int) std::floor(angle);
      angle -= hidx;
      const int id = (sx + 1) * 3 + (sy + 1);
      t->mag0[id] = mag * (1.f - angle);
      t->mag1[id] = mag * angle;
      if (hidx < 0)
        hidx += nbins;
      else if (hidx >= nbins)
        hidx -= nbins;
      t->bin0[id] = hidx;
      hidx++;
      if (hidx >= nbins)
        hidx = 0;
      t->bin1[id] = hidx;
      for (int k = 0; k < nbins; k++)
        t->coef[id][k] = (k == t->bin0[id]) ? t->mag0[id] : ((k == t->bin1[id]) ? t->mag1[id] : 0.f);
    }
  // pixData: Gaussian window (sigma = 4) x bilinear cell weights, grouped 1-cell | 2-cell | 4-cell pixels
  const int bs = 16, csz = 8, nc = 2;
  const float sigma = 4.0f, scale = 1.f / (sigma * sigma * 2);
  struct Px
  {
    int x, y, n, cell[4];
    float w[4];
  };
  std::vector<Px> g1, g2, g4;
  for (int j = 0; j < bs; j++)
    for (int i = 0; i < bs; i++)
    {
      const float di = i - bs * 0.5f, dj = j - bs * 0.5f;
      const float gw = std::exp(-(di * di + dj * dj) * scale);
      float cellX = (j + 0.5f) / csz - 0.5f, cellY = (i + 0.5f) / csz - 0.5f;
      int x0 = (int) std::floor(cellX), y0 = (int) std::floor(cellY);
      int x1 = x0 + 1, y1 = y0 + 1;
      cellX -= x0;
      cellY -= y0;
      const bool xin0 = (unsigned) x0 < (unsigned) nc, xin1 = (unsigned) x1 < (unsigned) nc;
      const bool yin0 = (unsigned) y0 < (unsigned) nc, yin1 = (unsigned) y1 < (unsigned) nc;
      Px p;
      p.x = j;
      p.y = i;
      std::memset(p.cell, 0, sizeof(p.cell));
      std::memset(p.w, 0, sizeof(p.w));
      if (xin0 && xin1)
      {
        if (yin0 && yin1)
        {
          p.n = 4;
          p.cell[0] = x0 * nc + y0;
          p.w[0] = gw * ((1.f - cellX) * (1.f - cellY));
          p.cell[1] = x1 * nc + y0;
          p.w[1] = gw * (cellX * (1.f - cellY));
          p.cell[2] = x0 * nc + y1;
          p.w[2] = gw * ((1.f - cellX) * cellY);
          p.cell[3] = x1 * nc + y1;
          p.w[3] = gw * (cellX * cellY);
          g4.push_back(p);
        }
        else
        {
          if (yin0)
          {
            y1 = y0;
            cellY = 1.f - cellY;
          }
          p.n = 2;
          p.cell[0] = x0 * nc + y1;
          p.w[0] = gw * ((1.f - cellX) * cellY);
          p.cell[1] = x1 * nc + y1;
          p.w[1] = gw * (cellX * cellY);
          g2.push_back(p);
        }
      }
      else
      {
        if (xin0)
        {
          x1 = x0;
          cellX = 1.f - cellX;
        }
        if (yin0 && yin1)
        {
          p.n = 2;
          p.cell[0] = x1 * nc + y0;
          p.w[0] = gw * (cellX * (1.f - cellY));
          p.cell[1] = x1 * nc + y1;
          p.w[1] = gw * (cellX * cellY);
          g2.push_back(p);
        }
        else
        {
          if (yin0)
          {
            y1 = y0;
            cellY = 1.f - cellY;
          }
          p.n = 1;
          p.cell[0] = x1 * nc + y1;
          p.w[0] = gw * (cellX * cellY);
          g1.push_back(p);
        }
      }
    }
  int k = 0;
  for (const std::vector<Px>* grp : { &g1, &g2, &g4 })
    for (const Px& p : *grp)
    {
      t->pix_x[k] = p.x;
      t->pix_y[k] = p.y;
      for (int c = 0; c < 4; c++)
        t->pix_wcell[k][c] = 0.f;
      for (int c = 0; c < p.n; c++)
        t->pix_wcell[k][p.cell[c]] = p.w[c];
      k++;
    }
}

// ---- device --------------------------------------------------------------------------------------------------
constexpr int kNBlocks = 77;  // 11 block columns (x = 0..80 step 8) x 7 block rows
constexpr int kCodeW = 96, kCodeH = 64;

__global__ __launch_bounds__(256) void k_hog_svm(const uint32_t* __restrict__ images,
  const int32_t* __restrict__ slot_of_hyp, const int64_t* __restrict__ n_hyp, const HogTablesDev* __restrict__ Tg,
  const float* __restrict__ svm_w, double rho, agh_hypothesis* __restrict__ out, uint8_t* __restrict__ keep,
  double* __restrict__ sums, float* __restrict__ desc_out, int debug_stop)
{
  __shared__ uint32_t bm[kImageWords + 2];
  __shared__ __attribute__((aligned(4))) uint8_t code[kCodeH * kCodeW];
  __shared__ uint8_t nzk[kNBlocks][256];  // per block: pixData entries with a non-zero gradient, in order
  __shared__ int nzc[kNBlocks];
  __shared__ uint8_t border[kNBlocks + 3];  // blocks by descending list length
  __shared__ float hist[kNBlocks][36];
  __shared__ float grp[882];
  __shared__ __attribute__((aligned(16))) HogTablesDev Ts;  // the 11 KiB of tables are hit on every vote: keep them in LDS

  const int h = blockIdx.x;
  if (n_hyp && (int64_t) h >= *n_hyp)  // (n_hyp == nullptr: the grid is exact -- the training path's descriptor pass)
    return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < (int) (sizeof(HogTablesDev) / 4); k += 256)
    ((unsigned*) &Ts)[k] = ((const unsigned*) Tg)[k];
  const HogTablesDev* T = &Ts;
  const uint32_t* im = images + (int64_t) (slot_of_hyp ? slot_of_hyp[h] : h) * kImageWords;
  for (int k = tid; k < kImageWords; k += 256)
    bm[k] = im[k];
  __syncthreads();
  // gradient code per pixel: centred differences, BORDER_REFLECT_101 at x = 0 and y = 0 (hog.cpp computeGradient).
  // One thread per (row, 32-pixel word): the four neighbour rows come as 32-bit vectors (funnel shifts of the packed
  // image), then code = 4 + 3 (right - left) + (down - up) per pixel, four codes per LDS word.
  auto row_bits = [&](int y, int x) -> unsigned {  // pixels (y, x) .. (y, x + 31)
    const int b = y * 100 + x;
    return __funnelshift_r(bm[b >> 5], bm[(b >> 5) + 1], b & 31);
  };
  for (int unit = tid; unit < kCodeH * (kCodeW / 32); unit += 256)
  {
    const int y = unit / (kCodeW / 32), x0 = (unit % (kCodeW / 32)) * 32;
    const unsigned C = row_bits(y, x0);
    const unsigned R = row_bits(y, x0 + 1);
    const unsigned L = x0 == 0 ? ((C << 1) | ((C >> 1) & 1u)) : row_bits(y, x0 - 1);  // x = 0 reflects to x = 1
    const unsigned D = row_bits(y + 1, x0);
    const unsigned U = row_bits(y == 0 ? 1 : y - 1, x0);                               // y = 0 reflects to y = 1
    unsigned* dst = reinterpret_cast<unsigned*>(&code[y * kCodeW + x0]);
#pragma unroll
    for (int i = 0; i < 32; i += 4)
    {
      unsigned w = 0;
#pragma unroll
      for (int j = 0; j < 4; j++)
      {
        const int sx = (int) ((R >> (i + j)) & 1u) - (int) ((L >> (i + j)) & 1u);
        const int sy = (int) ((D >> (i + j)) & 1u) - (int) ((U >> (i + j)) & 1u);
        w |= (unsigned) ((sx + 1) * 3 + (sy + 1)) << (8 * j);
      }
      dst[i >> 2] = w;
    }
  }
  __syncthreads();
  if (debug_stop == 1)  // (AGH_DEBUG_STOP_HOG: phase-timing aid, like the other kernels' debug stops)
    return;
  // per block: ordered list of the pixData entries with a non-zero gradient (all other pixels vote +0.0f: nothing)
  // (a lane's four pixData entries are the same for every block: their in-block offsets are read once, not once per block --
  // the stores to nzk keep the compiler from hoisting the table reads itself)
  int pofs[4];
#pragma unroll
  for (int c = 0; c < 4; c++)
    pofs[c] = T->pix_y[c * 64 + lane] * kCodeW + T->pix_x[c * 64 + lane];
  for (int b = wave; b < kNBlocks; b += 4)
  {
    const int x0 = (b / 7) * 8, y0 = (b % 7) * 8;
    int cnt = 0;
    bool nzq[4];
#pragma unroll
    for (int c = 0; c < 4; c++)  // the four look-ups (pixel position -> gradient code) are issued together
      nzq[c] = code[y0 * kCodeW + x0 + pofs[c]] != 4;
#pragma unroll
    for (int c = 0; c < 4; c++)
    {
      const unsigned long long m = __ballot(nzq[c]);
      if (nzq[c])
        nzk[b][cnt + __popcll(m & ((1ull << lane) - 1ull))] = (uint8_t) (c * 64 + lane);
      cnt += __popcll(m);
    }
    if (lane == 0)
      nzc[b] = cnt;
  }
  __syncthreads();
  if (debug_stop == 2)
    return;
  // block histograms (HOGCache::getBlock): one (block, cell) per work item, votes in pixData order.  The nine bins live
  // in registers; every listed pixel adds mag * weight to its two bins and +0.0f (exact) to the others, and a pixel
  // that does not vote for this cell has weight 0 -- so there is no data-dependent branch and no LDS read-modify-write.
  // 308 (block, cell) items on 256 threads: a thread's time is the number of listed pixels of its items, and the work-group
  // waits for its slowest thread.  Items in plain order gave the threads 0..51 two items each, whatever their length; now the
  // blocks are ranked by their list length (border: longest first) and the 52 second items are the SHORTEST lists, handed to the
  // threads whose first item is the shortest of the first 256.
  if (tid < kNBlocks)
  {
    const int mine = nzc[tid];
    int r = 0;
    for (int q = 0; q < kNBlocks; q++)
      r += (nzc[q] > mine || (nzc[q] == mine && q < tid)) ? 1 : 0;
    border[r] = (uint8_t) tid;
  }
  __syncthreads();
  for (int pass = 0; pass < 2; pass++)
  {
    // second items: ranks 256 .. 307; the longest of them (256) goes to the thread with the shortest first item (255)
    const int wi = pass == 0 ? tid : 256 + (255 - tid);
    if (pass == 1 && wi >= kNBlocks * 4)
      break;
    const int b = border[wi >> 2], cell = wi & 3;
    const int x0 = (b / 7) * 8, y0 = (b % 7) * 8;
    float hh[9];
#pragma unroll
    for (int k = 0; k < 9; k++)
      hh[k] = 0.f;
    const int cnt = nzc[b];
    for (int q = 0; q < cnt; q++)
    {
      const int k = nzk[b][q];
      const float w = T->pix_wcell[k][cell];
      const int cd = code[(y0 + T->pix_y[k]) * kCodeW + x0 + T->pix_x[k]];
      const float4 c0 = *reinterpret_cast<const float4*>(&T->coef[cd][0]);
      const float4 c1 = *reinterpret_cast<const float4*>(&T->coef[cd][4]);
      const float c8 = T->coef[cd][8];
      // mag * w for the pixel's two bins, +0.0f for the other seven (w >= 0, mag >= 0: the zero products are +0.0f)
      hh[0] = hh[0] + c0.x * w;
      hh[1] = hh[1] + c0.y * w;
      hh[2] = hh[2] + c0.z * w;
      hh[3] = hh[3] + c0.w * w;
      hh[4] = hh[4] + c1.x * w;
      hh[5] = hh[5] + c1.y * w;
      hh[6] = hh[6] + c1.z * w;
      hh[7] = hh[7] + c1.w * w;
      hh[8] = hh[8] + c8 * w;
    }
#pragma unroll
    for (int k = 0; k < 9; k++)
      hist[b][cell * 9 + k] = hh[k];
  }
  __syncthreads();
  if (debug_stop == 3)
    return;
  // L2-Hys (HOGCache::normalizeBlockHistogram), sequential per block
  if (tid < kNBlocks)
  {
    float* hh = hist[tid];
    float sum = 0;
    for (int k = 0; k < 36; k++)
      sum += hh[k] * hh[k];
    float scale = 1.f / (sqrtf(sum) + 36 * 0.1f);
    const float thresh = 0.2f;
    sum = 0;
    for (int k = 0; k < 36; k++)
    {
      hh[k] = fminf(hh[k] * scale, thresh);
      sum += hh[k] * hh[k];
    }
    scale = 1.f / (sqrtf(sum) + 1e-3f);
    for (int k = 0; k < 36; k++)
      hh[k] *= scale;
  }
  __syncthreads();
  if (debug_stop == 4)
    return;
  // descriptor layout: window-major, block bx*7+by, cell cx*2+cy, 9 bins; window 1 starts at block column 4
  for (int m = tid; m < 882; m += 256)
  {
    const int win = m / 441, idx = (m % 441) * 4;
    const int bw = idx / 36, j = idx % 36;
    const int b = (win * 4 + bw / 7) * 7 + bw % 7;
    const float* d = &hist[b][j];
    if (svm_w)
    {
      const float* w = svm_w + m * 4;
      grp[m] = w[0] * d[0] + w[1] * d[1] + w[2] * d[2] + w[3] * d[3];  // CvSVMKernel::calc_non_rbf_base
    }
    if (desc_out)
      for (int q = 0; q < 4; q++)
        desc_out[(int64_t) h * 3528 + m * 4 + q] = d[q];
  }
  __syncthreads();
  if (debug_stop == 5 || !svm_w)  // (no model: descriptors only)
    return;
  if (wave == 0)
  {
    // CvSVM::predict's accumulation: the 882 four-product partials (floats) are added to a double IN INDEX ORDER.  One lane
    // doing it alone issued 882 x (LDS read, conversion, add): 11.6 us.  Now lane l holds entries 14 l .. 14 l + 13 as doubles
    // (fetched and converted side by side), and the RUNNING SUM travels: in step l every lane adds its own fourteen to the sum
    // that came in -- only lane l's result is the real one -- and that one is handed on as a scalar (two v_readlane per
    // fourteen adds).  Same additions in the same order; the chain is the 882 adds and nothing else.
    double gd[14];
#pragma unroll
    for (int u = 0; u < 14; u++)
      gd[u] = lane < 63 ? (double) grp[lane * 14 + u] : 0.0;
    double s = 0;
    for (int l = 0; l < 63; l++)
    {
      double t = s;
#pragma unroll
      for (int u = 0; u < 14; u++)
        t += gd[u];
      s = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t), l), __builtin_amdgcn_readlane(__double2loint(t), l));
    }
    if (lane == 0)
    {
      const float res = (float) (s * 1.0 + 0.0);
      const double sum = -rho + 1.0 * res;  // CvSVM::predict: class_labels[sum > 0 ? 0 : 1] = {-1, +1}
      const uint8_t k = (sum > 0) ? 0 : 1;  // the reference keeps prediction == 1 (learning.cpp:225-227)
      if (keep)
        keep[h] = k;
      if (sums)
        sums[h] = sum;
      if (out)
        out[h].svm_keep = k;
    }
  }
}

int hog_svm(Ctx* c, int64_t n_hyp_cap, uint8_t* d_keep, hipStream_t st)
{
  if (n_hyp_cap <= 0)
    return AGH_OK;
  timing_mark(c, "start", st);
  if (c->svm_general)
  {
    // several support vectors and/or the quadratic kernel: descriptors first, then CvSVM::predict's kernel row per
    // hypothesis and the alpha-weighted sum (train.hip)
    float* desc = c->d_desc_out;
    if (!desc)
    {
      if (n_hyp_cap > c->cls_desc_cap)
      {
        if (c->d_cls_desc)
          (void) hipFree(c->d_cls_desc);
        c->d_cls_desc = nullptr;
        c->cls_desc_cap = 0;
        if (hipMalloc((void**) &c->d_cls_desc, (size_t) n_hyp_cap * 3528 * sizeof(float)) != hipSuccess)
        {
          c->err = "agh_classify: out of device memory for the descriptors";
          return AGH_ERR_HIP;
        }
        c->cls_desc_cap = n_hyp_cap;
      }
      desc = c->d_cls_desc;
    }
    hipLaunchKernelGGL(k_hog_svm, dim3((unsigned) n_hyp_cap), dim3(256), 0, st, c->d_images, c->d_slot_index, c->d_nout_last,
      c->d_hog, (const float*) nullptr, 0.0, (agh_hypothesis*) nullptr, (uint8_t*) nullptr, (double*) nullptr, desc, 0);
    const int rc = svm_predict_general(c, desc, n_hyp_cap, d_keep, st);
    timing_mark(c, "hog_svm", st);
    return rc;
  }
  hipLaunchKernelGGL(k_hog_svm, dim3((unsigned) n_hyp_cap), dim3(256), 0, st, c->d_images, c->d_slot_index, c->d_nout_last,
    c->d_hog, c->d_svm_w, c->svm_rho, c->d_out_last, d_keep, c->d_svm_sums, c->d_desc_out, c->debug_stop_hog);
  timing_mark(c, "hog_svm", st);
  return hipGetLastError() == hipSuccess ? AGH_OK : AGH_ERR_HIP;
}

// Descriptors of `n` packed images (image index order[h] -> descriptor row h; order may be null): the feature pass of
// Learning::convertData (learning.cpp:253-288).
int hog_images(Ctx* c, const uint32_t* d_images, const int32_t* d_order, int64_t n, float* d_desc, hipStream_t st)
{
  if (n <= 0)
    return AGH_OK;
  hipLaunchKernelGGL(k_hog_svm, dim3((unsigned) n), dim3(256), 0, st, d_images, d_order, (const int64_t*) nullptr, c->d_hog,
    (const float*) nullptr, 0.0, (agh_hypothesis*) nullptr, (uint8_t*) nullptr, (double*) nullptr, d_desc, 0);
  return hipGetLastError() == hipSuccess ? AGH_OK : AGH_ERR_HIP;
}

// ---- IEEE self-test ------------------------------------------------------------------------------------------
__global__ void k_selftest(const double* __restrict__ a, const double* __restrict__ b, int64_t n, double* __restrict__ o)
{
  const int64_t i = blockIdx.x * (int64_t) blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const double x = a[i], y = b[i];
  double* r = o + i * 8;
  r[0] = x / y;
  r[1] = sqrt(fabs(x));
  r[2] = x * y + x;        // must NOT be fused
  r[3] = (x * x + y * y) + x * y;
  const float xf = (float) x, yf = (float) y;
  r[4] = (double) (xf / yf);
  r[5] = (double) sqrtf(fabsf(xf));
  r[6] = (double) (xf * yf + xf);
  r[7] = 1.0 / sqrt(x * x + 1.0);
}

int64_t selftest_math(Ctx* c, int64_t n, uint64_t seed)
{
  if (n <= 0)
    return 0;
  std::vector<double> a(n), b(n), o(n * 8);
  uint64_t s = seed ? seed : 88172645463325252ull;
  auto next = [&]() {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
  };
  for (int64_t i = 0; i < n; i++)
  {
    // mantissas uniform, exponents over a range that includes what the kernels see
    const double ma = (double) (next() >> 11) / 9007199254740992.0 + 0.5, mb = (double) (next() >> 11) / 9007199254740992.0 + 0.5;
    a[i] = std::ldexp(ma, (int) (next() % 40) - 30) * ((next() & 1) ? 1 : -1);
    b[i] = std::ldexp(mb, (int) (next() % 40) - 30) * ((next() & 1) ? 1 : -1);
  }
  double *da = nullptr, *db = nullptr, *dout = nullptr;
  if (hipMalloc((void**) &da, n * 8) != hipSuccess || hipMalloc((void**) &db, n * 8) != hipSuccess ||
      hipMalloc((void**) &dout, n * 64) != hipSuccess)
    return -1;
  (void) hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
  (void) hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_selftest, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, c->stream, da, db, n, dout);
  (void) hipStreamSynchronize(c->stream);
  (void) hipMemcpy(o.data(), dout, n * 64, hipMemcpyDeviceToHost);
  (void) hipFree(da);
  (void) hipFree(db);
  (void) hipFree(dout);
  int64_t bad = 0;
  for (int64_t i = 0; i < n; i++)
  {
    const double x = a[i], y = b[i];
    volatile double p = x * y;  // volatile: keep the host from fusing, whatever the flags
    volatile double xx = x * x, yy = y * y;
    volatile double q = xx + yy;
    const float xf = (float) x, yf = (float) y;
    volatile float pf = xf * yf;
    volatile double x1 = xx + 1.0;
    const double exp[8] = { x / y, std::sqrt(std::fabs(x)), p + x, q + p, (double) (xf / yf),
      (double) std::sqrt(std::fabs(xf)), (double) (float) (pf + xf), 1.0 / std::sqrt(x1) };
    for (int k = 0; k < 8; k++)
      if (std::memcmp(&exp[k], &o[i * 8 + k], 8) != 0)
        bad++;
  }
  return bad;
}

}  // namespace agh

// ---- C ABI: SVM + classify + introspection -------------------------------------------------------------------
using namespace agh;

#define HIPCHK2(ctx, expr)                                                                            \
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

int agh_load_svm(agh_ctx* ctx, const float* weights, int32_t n_weights, double rho)
{
  if (!ctx || !weights)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (n_weights != 3528)
  {
    c->err = "agh_load_svm: the HOG descriptor has 3528 entries (2 windows x 49 blocks x 36)";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  HIPCHK2(c, hipSetDevice(c->device));
  HIPCHK2(c, hipMemcpy(c->d_svm_w, weights, sizeof(float) * 3528, hipMemcpyHostToDevice));
  c->svm_rho = rho;
  c->svm_general = false;
  c->svm_kernel = AGH_SVM_LINEAR;
  c->svm_n_sv = 1;
  c->has_svm = true;
  return AGH_OK;
}

// Reads the OpenCV "!!opencv-ml-svm" YAML written by CvSVM::save for the two model shapes Learning::convertData
// produces: a linear C_SVC compacted to one support vector -- the format of the model shipped with the reference
// (svm_032015_linear_20_20_same) -- or a POLY degree-2 C_SVC with its support vectors (learning.cpp:296-312).
int agh_load_svm_file(agh_ctx* ctx, const char* path)
{
  if (!ctx || !path)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  FILE* f = std::fopen(path, "rb");
  if (!f)
  {
    c->err = std::string("File ") + path + " does not exist!";  // learning.cpp:176
    return AGH_ERR_IO;
  }
  std::string txt;
  char buf[8192];
  size_t r;
  while ((r = std::fread(buf, 1, sizeof(buf), f)) > 0)
    txt.append(buf, r);
  std::fclose(f);
  const size_t kp = txt.find("kernel:"), sv = txt.find("support_vectors:"), df = txt.find("decision_functions:");
  if (kp == std::string::npos || sv == std::string::npos || df == std::string::npos || txt.find("C_SVC") == std::string::npos)
  {
    c->err = "not an OpenCV C_SVC model file";
    return AGH_ERR_IO;
  }
  const std::string kline = txt.substr(kp, txt.find('\n', kp) - kp);
  int kernel_type;
  if (kline.find("LINEAR") != std::string::npos)
    kernel_type = AGH_SVM_LINEAR;
  else if (kline.find("POLY") != std::string::npos)
  {
    auto field = [&](const char* name) {
      const size_t p = kline.find(name);
      return p == std::string::npos ? NAN : std::strtod(kline.c_str() + p + std::strlen(name), nullptr);
    };
    if (field("degree:") != 2.0 || field("gamma:") != 1.0 || field("coef0:") != 0.0)
    {
      c->err = "POLY models are supported with degree 2, gamma 1, coef0 0 (what Learning::convertData trains)";
      return AGH_ERR_IO;
    }
    kernel_type = AGH_SVM_POLY2;
  }
  else
  {
    c->err = "only LINEAR and POLY kernels are supported";
    return AGH_ERR_IO;
  }
  auto parse_reals = [&](size_t lb, size_t rb, auto&& sink) {  // OpenCV parses reals as double, then stores them
    const char* s = txt.c_str() + lb + 1;
    const char* end = txt.c_str() + rb;
    while (s < end)
    {
      char* e2 = nullptr;
      const double v = std::strtod(s, &e2);
      if (e2 == s)
      {
        s++;
        continue;
      }
      sink(v);
      s = e2;
    }
  };
  std::vector<float> w;
  int n_sv = 0;
  for (size_t pos = sv;;)
  {
    const size_t lb = txt.find('[', pos);
    if (lb == std::string::npos || lb > df)
      break;
    const size_t rb = txt.find(']', lb);
    if (rb == std::string::npos || rb > df)
    {
      c->err = "malformed support_vectors section";
      return AGH_ERR_IO;
    }
    const size_t before = w.size();
    parse_reals(lb, rb, [&](double v) { w.push_back((float) v); });
    if (w.size() - before != 3528)
    {
      c->err = "expected support vectors of 3528 weights";
      return AGH_ERR_IO;
    }
    n_sv++;
    pos = rb + 1;
  }
  const size_t rp = txt.find("rho:", df), ap = txt.find("alpha:", df);
  if (rp == std::string::npos || ap == std::string::npos || n_sv == 0)
  {
    c->err = "expected support vectors, a rho and alphas";
    return AGH_ERR_IO;
  }
  const double rho = std::strtod(txt.c_str() + rp + 4, nullptr);
  std::vector<double> alpha;
  const size_t alb = txt.find('[', ap), arb = txt.find(']', ap);
  if (alb == std::string::npos || arb == std::string::npos)
  {
    c->err = "malformed alpha section";
    return AGH_ERR_IO;
  }
  parse_reals(alb, arb, [&](double v) { alpha.push_back(v); });
  if ((int) alpha.size() != n_sv)
  {
    c->err = "alpha count differs from the support vector count";
    return AGH_ERR_IO;
  }
  return agh_load_svm_model(ctx, kernel_type, w.data(), n_sv, 3528, alpha.data(), rho);
}

int agh_classify_device(agh_ctx* ctx, uint8_t* d_keep, void* hip_stream)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (!c->has_svm)
  {
    c->err = "agh_classify: no SVM loaded";
    return AGH_ERR_NO_SVM;
  }
  if (!c->d_out_last || !c->d_nout_last)
  {
    c->err = "agh_classify: call agh_find_hands first";
    return AGH_ERR_STATE;
  }
  HIPCHK2(c, hipSetDevice(c->device));
  hipStream_t st = hip_stream ? (hipStream_t) hip_stream : c->stream;
  return hog_svm(c, std::min<int64_t>(c->last_s * 8, c->last_cap), d_keep, st);
}

int agh_classify(agh_ctx* ctx, uint8_t* keep, int64_t cap, int64_t* n_kept)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (n_kept)
    *n_kept = 0;
  if (c->last_nout < 0)
  {
    c->err = "agh_classify: needs the hypotheses of a completed agh_find_hands call";
    return AGH_ERR_STATE;
  }
  const int64_t n = c->last_nout;
  if (n > cap || (n > 0 && !keep))
  {
    c->err = "agh_classify: keep buffer too small";
    return AGH_ERR_CAPACITY;
  }
  if (n > c->keep_cap)
  {
    if (c->d_keep)
      (void) hipFree(c->d_keep);
    if (c->d_svm_sums)
      (void) hipFree(c->d_svm_sums);
    c->d_keep = nullptr;
    c->d_svm_sums = nullptr;
    HIPCHK2(c, hipMalloc((void**) &c->d_keep, n));
    HIPCHK2(c, hipMalloc((void**) &c->d_svm_sums, n * sizeof(double)));
    c->keep_cap = n;
  }
  if (n == 0)
    return AGH_OK;
  // K3 writes the flags straight into pinned host memory of the context: no read-back copy behind the kernel
  if (n > c->h_pin_keep_cap || !c->h_pin_keep)
  {
    if (c->h_pin_keep)
      (void) hipHostFree(c->h_pin_keep);
    c->h_pin_keep = nullptr;
    c->h_pin_keep_cap = 0;
    void* p = nullptr;
    HIPCHK2(c, hipHostMalloc(&p, (size_t) std::max<int64_t>(n, 16384), hipHostMallocDefault));
    c->h_pin_keep = static_cast<uint8_t*>(p);
    c->h_pin_keep_cap = std::max<int64_t>(n, 16384);
  }
  int rc = agh_classify_device(ctx, c->h_pin_keep, nullptr);
  if (rc != AGH_OK)
    return rc;
  HIPCHK2(c, hipStreamSynchronize(c->stream));
  std::memcpy(keep, c->h_pin_keep, (size_t) n);
  int64_t k = 0;
  for (int64_t i = 0; i < n; i++)
    k += keep[i] ? 1 : 0;
  if (n_kept)
    *n_kept = k;
  return AGH_OK;
}

int agh_get_packed_images(agh_ctx* ctx, uint32_t* images, int64_t cap_hyp)
{
  if (!ctx || (cap_hyp > 0 && !images) || cap_hyp < 0)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (c->last_nout < 0)
  {
    c->err = "agh_get_packed_images: needs a completed agh_find_hands call";
    return AGH_ERR_STATE;
  }
  const int64_t n = std::min<int64_t>(cap_hyp, c->last_nout);
  if (n == 0)
    return 0;
  HIPCHK2(c, hipSetDevice(c->device));
  HIPCHK2(c, hipDeviceSynchronize());
  // the images sit in the per-sample slots; hypothesis h is slot d_slot_index[h] (the compaction's record)
  std::vector<int32_t> slot((size_t) n);
  std::vector<uint32_t> words((size_t) c->last_s * 8 * kImageWords);
  HIPCHK2(c, hipMemcpy(slot.data(), c->d_slot_index, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  HIPCHK2(c, hipMemcpy(words.data(), c->d_images, words.size() * 4, hipMemcpyDeviceToHost));
  for (int64_t h = 0; h < n; h++)
    std::memcpy(images + h * kImageWords, &words[(size_t) slot[(size_t) h] * kImageWords], kImageWords * 4);
  return (int) n;
}

int agh_classify_images(agh_ctx* ctx, const uint32_t* images, int64_t n, uint8_t* keep, double* sums)
{
  if (!ctx || n < 0 || (n > 0 && (!images || !keep)))
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (!c->has_svm)
  {
    c->err = "agh_classify_images: no SVM loaded";
    return AGH_ERR_NO_SVM;
  }
  if (n == 0)
    return AGH_OK;
  if (n > (1 << 24))
  {
    c->err = "agh_classify_images: more than 2^24 images in one call";
    return AGH_ERR_CAPACITY;
  }
  HIPCHK2(c, hipSetDevice(c->device));
  if (n > c->cls_images_cap)
  {
    for (void* p : { (void*) c->d_cls_images, (void*) c->d_cls_keep, (void*) c->d_cls_sums })
      if (p)
        (void) hipFree(p);
    c->d_cls_images = nullptr;
    c->d_cls_keep = nullptr;
    c->d_cls_sums = nullptr;
    c->cls_images_cap = 0;
    const int64_t cap = std::max<int64_t>(n, 1024);
    HIPCHK2(c, hipMalloc((void**) &c->d_cls_images, (size_t) cap * kImageWords * 4));
    HIPCHK2(c, hipMalloc((void**) &c->d_cls_keep, (size_t) cap));
    HIPCHK2(c, hipMalloc((void**) &c->d_cls_sums, (size_t) cap * sizeof(double)));
    c->cls_images_cap = cap;
  }
  hipStream_t st = c->stream;
  HIPCHK2(c, hipMemcpyAsync(c->d_cls_images, images, (size_t) n * kImageWords * 4, hipMemcpyHostToDevice, st));
  int rc = AGH_OK;
  if (c->svm_general)
  {
    if (n > c->cls_desc_cap)
    {
      if (c->d_cls_desc)
        (void) hipFree(c->d_cls_desc);
      c->d_cls_desc = nullptr;
      c->cls_desc_cap = 0;
      HIPCHK2(c, hipMalloc((void**) &c->d_cls_desc, (size_t) n * 3528 * sizeof(float)));
      c->cls_desc_cap = n;
    }
    rc = hog_images(c, c->d_cls_images, nullptr, n, c->d_cls_desc, st);
    if (rc == AGH_OK)
      rc = svm_predict_images(c, c->d_cls_desc, n, c->d_cls_keep, c->d_cls_sums, st);
  }
  else
  {
    hipLaunchKernelGGL(k_hog_svm, dim3((unsigned) n), dim3(256), 0, st, (const uint32_t*) c->d_cls_images,
      (const int32_t*) nullptr, (const int64_t*) nullptr, c->d_hog, (const float*) c->d_svm_w, c->svm_rho,
      (agh_hypothesis*) nullptr, c->d_cls_keep, c->d_cls_sums, (float*) nullptr, 0);
    rc = hipGetLastError() == hipSuccess ? AGH_OK : AGH_ERR_HIP;
  }
  if (rc != AGH_OK)
  {
    c->err = "agh_classify_images: launch failed";
    return rc;
  }
  HIPCHK2(c, hipMemcpyAsync(keep, c->d_cls_keep, (size_t) n, hipMemcpyDeviceToHost, st));
  if (sums)
    HIPCHK2(c, hipMemcpyAsync(sums, c->d_cls_sums, (size_t) n * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK2(c, hipStreamSynchronize(st));
  return AGH_OK;
}

int agh_get_images(agh_ctx* ctx, uint8_t* images, int64_t cap_hyp)
{
  if (!ctx || !images)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (c->last_nout < 0)
  {
    c->err = "agh_get_images: needs a completed agh_find_hands call";
    return AGH_ERR_STATE;
  }
  const int64_t n = std::min<int64_t>(cap_hyp, c->last_nout);
  HIPCHK2(c, hipDeviceSynchronize());
  std::vector<int32_t> slot(n);
  std::vector<uint32_t> words((size_t) c->last_s * 8 * kImageWords);
  if (n > 0)
  {
    HIPCHK2(c, hipMemcpy(slot.data(), c->d_slot_index, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    HIPCHK2(c, hipMemcpy(words.data(), c->d_images, words.size() * 4, hipMemcpyDeviceToHost));
  }
  for (int64_t h = 0; h < n; h++)
  {
    const uint32_t* w = &words[(size_t) slot[h] * kImageWords];
    uint8_t* im = images + h * 8000;
    for (int b = 0; b < 8000; b++)
      im[b] = ((w[b >> 5] >> (b & 31)) & 1u) ? 255 : 0;
  }
  return (int) n;
}

int agh_get_hog(agh_ctx* ctx, float* desc, double* sums, int64_t cap_hyp)
{
  if (!ctx)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  if (c->last_nout < 0 || !c->has_svm)
  {
    c->err = "agh_get_hog: needs an SVM and a completed agh_find_hands call";
    return AGH_ERR_STATE;
  }
  const int64_t n = std::min<int64_t>(cap_hyp, c->last_nout);
  if (n == 0)
    return 0;
  float* d_desc = nullptr;
  double* d_sums = nullptr;
  uint8_t* d_keep = nullptr;
  HIPCHK2(c, hipMalloc((void**) &d_desc, (size_t) c->last_nout * 3528 * sizeof(float)));
  HIPCHK2(c, hipMalloc((void**) &d_sums, (size_t) c->last_nout * sizeof(double)));
  HIPCHK2(c, hipMalloc((void**) &d_keep, (size_t) c->last_nout));
  float* save_desc = c->d_desc_out;
  double* save_sums = c->d_svm_sums;
  c->d_desc_out = d_desc;
  c->d_svm_sums = d_sums;
  int rc = agh_classify_device(ctx, d_keep, nullptr);
  c->d_desc_out = save_desc;
  c->d_svm_sums = save_sums;
  if (rc == AGH_OK && hipStreamSynchronize(c->stream) != hipSuccess)
    rc = AGH_ERR_HIP;
  if (rc == AGH_OK && desc)
    (void) hipMemcpy(desc, d_desc, (size_t) n * 3528 * sizeof(float), hipMemcpyDeviceToHost);
  if (rc == AGH_OK && sums)
    (void) hipMemcpy(sums, d_sums, (size_t) n * sizeof(double), hipMemcpyDeviceToHost);
  (void) hipFree(d_desc);
  (void) hipFree(d_sums);
  (void) hipFree(d_keep);
  return rc == AGH_OK ? (int) n : rc;
}

}  // extern "C"
