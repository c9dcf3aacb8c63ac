// vc_detect.hip -- first slice of the image front-end: dot detection on the GPU.
//
// What the reference runs per image before the solver sees a corner (VicalibTask::AddImageMeasurements,
// vicalib-task.cc:264-270, parameters :116-122):
//     image_processing_[ii].Process(img->data(), w, h, pitch);      // calibu::ImageProcessing: threshold, gradient, labels
//     conic_finder_[ii].Find(image_processing_[ii]);                // calibu::ConicFinder: one conic per dot
// and then uses conics[i].center.  Calibu's source is not in /root/reference: the kernels below implement the published algorithms
// behind those calls (adaptive threshold against the local mean from an integral image; 4-connected components; dual-conic fit
// on the image gradient, Ouellet & Hebert 2009) exactly as the numpy restatement used by the tests has them -- the parity bar for this row is
// that restatement plus rendered images with known ellipse centres; against Calibu itself it is, and stays, unpinned.
// Grid matching (TargetGridDot::FindTarget, :274) is not part of this slice.
//
// Data flow on the device (one upload, eight launches, one download and ONE host synchronisation per image):
//   k_det_rows / k_det_cols   integral image S (uint32, exact); columns: 64-column strips, 16 wavefronts per strip, every wavefront sums
//                             its band of rows with all loads in flight, bands combined through LDS
//   k_det_tile                32 x 32 tile per workgroup: dot mask = (I * window_area < at_threshold * window_sum); union-find of the
//                             tile's dot pixels IN LDS (atomicMin links to the smaller index), area and bounding box of every tile-local
//                             component in LDS; out: label = pixel index of the tile-local root, the local root carries the piece's
//                             statistics
//   k_det_border              the pixel pairs across tile edges: union-find on the global labels (a few per cent of the pixels)
//   k_det_push                every tile-local root that is not its component's root adds its piece's statistics to the root's (integer
//                             atomics: order-independent).  Every component is named by its smallest pixel index (what the
//                             restatement's fixed-point iteration converges to): row-major order within a tile agrees with the image's
//   k_det_select              roots that pass area / density / aspect / border tests -> candidate list (atomic append)
//   k_det_order               candidates ranked by label on the device (the order of atomic appends is not deterministic, the ranked
//                             list is)
//   k_det_fit                 one wavefront per candidate: the 5 x 5 normal equations of the dual conic over the box's pixels
//                             (lane-strided, fixed-order wave sums), solved by lane 0; one record per candidate (centre, conic, box)
// The count and the first kFastRecs records come back in one copy into pinned memory (more candidates: one more copy).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <vector>
#include "../../include/vicalib_amd.h"

namespace {

constexpr int kGrow = 2;              // pixels the bounding box is grown by before the fit   (the restatement's GROW)
constexpr double kMinGrad2 = 1.0;     // squared gradient magnitude below which a pixel is ignored (MIN_GRAD2)

// what comes back per dot: centre, primal conic (image coordinates), the component's bounding box (x0, y0, x1, y1 inclusive)
struct DetRec { double cx, cy; double conic[9]; int box[4]; };
static_assert(sizeof(DetRec) == 104, "record layout");
constexpr int kHeadBytes = 16;         // the candidate count, in front of the records
constexpr int kFastRecs = 512;         // records that travel with the count in the first copy
constexpr int kTW = 32, kTH = 32;      // labelling tile
constexpr int kMaxCand = 4096;         // candidates per image (more: VC_ERR_UNSUPPORTED)

struct DetView {
  int w, h, pitch;
  const unsigned char* img;      // h x pitch
  unsigned* S;                   // (h + 1) x (w + 1) integral image
  int* lab;                      // h x w
  int* area; int* x0; int* x1; int* y0; int* y1;      // per root pixel
  int* cand; int* cand_sorted; int* n_cand; int max_cand;
  double thr; int rad;
  double min_area, min_density, min_aspect;
  int black_on_white;
  DetRec* recs;                    // one per candidate, in the order of the ranked list
};

// integral image, rows: one wavefront per row, eight consecutive pixels per lane and round, exclusive scan of the lane sums.
// The sums are kept modulo 2^32 (images above 16.8 MP would overflow a 32-bit total): every use takes the four-corner difference in
// unsigned arithmetic, which is exact as long as the WINDOW's sum fits -- it does, a window has at most (2 rad + 1)^2 pixels
__global__ __launch_bounds__(64) void k_det_rows(DetView v) {
  const int y = blockIdx.x, lane = threadIdx.x;
  unsigned carry = 0;
  unsigned* row = v.S + (size_t)(y + 1) * (v.w + 1);
  if (lane == 0) row[0] = 0;
  if (y == 0 && lane == 0) *v.n_cand = 0;
  for (int x0 = 0; x0 < v.w; x0 += 64 * 8) {
    unsigned px[8], s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int x = x0 + lane * 8 + k;
      unsigned val = x < v.w ? v.img[(size_t)y * v.pitch + x] : 0;
      if (!v.black_on_white) val = 255 - val;
      s += val; px[k] = s;
    }
    unsigned incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    const unsigned excl = incl - s + carry;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int x = x0 + lane * 8 + k; if (x < v.w) row[x + 1] = excl + px[k]; }
    carry += __shfl(incl, 63, 64);
  }
}
// ... columns: a strip of 64 columns per workgroup, 16 wavefronts, each with its own band of rows: band totals (loads independent of
// each other: all in flight) -> LDS -> every band starts from the sum of the bands above it and writes its rows' running sums
__global__ __launch_bounds__(1024) void k_det_cols(DetView v) {
  __shared__ unsigned tot[16][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int x = blockIdx.x * 64 + lane;
  const bool in = x <= v.w;
  const int W1 = v.w + 1;
  const int rpw = (v.h + 15) / 16, ya = 1 + wv * rpw, yb = min(ya + rpw, v.h + 1);
  unsigned s = 0;
  for (int y = ya; y < yb; y += 8) {
    unsigned t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = (in && y + k < yb) ? v.S[(size_t)(y + k) * W1 + x] : 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += t[k];
  }
  tot[wv][lane] = s;
  __syncthreads();
  unsigned run = 0;
  for (int k = 0; k < wv; ++k) run += tot[k][lane];
  if (wv == 0 && in) v.S[x] = 0;
  for (int y = ya; y < yb; y += 8) {
    unsigned t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = (in && y + k < yb) ? v.S[(size_t)(y + k) * W1 + x] : 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) { run += t[k]; if (in && y + k < yb) v.S[(size_t)(y + k) * W1 + x] = run; }
  }
}
__device__ __forceinline__ int det_find(const int* L, int x) {
  int r = x;
  while (true) { const int q = L[r]; if (q == r) return r; r = q; }
}
__device__ __forceinline__ void det_unite(int* L, int a, int b) {
  while (true) {
    a = det_find(L, a); b = det_find(L, b);
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(&L[b], a);          // link the larger root to the smaller one
    if (old == b) return;
    b = old;                                      // somebody linked b in the meantime: go on from where it points
  }
}
// threshold + labelling of one tile in LDS + the statistics of the tile's pieces
__global__ __launch_bounds__(256) void k_det_tile(DetView v) {
  __shared__ int L[kTW * kTH];
  __shared__ int s_area[kTW * kTH], s_x0[kTW * kTH], s_x1[kTW * kTH], s_y0[kTW * kTH], s_y1[kTW * kTH];
  const int tiles_x = (v.w + kTW - 1) / kTW;
  const int tx0 = (blockIdx.x % tiles_x) * kTW, ty0 = (blockIdx.x / tiles_x) * kTH;
  const int W1 = v.w + 1;
  int root[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = threadIdx.x + 256 * k, lx = i & (kTW - 1), ly = i / kTW, x = tx0 + lx, y = ty0 + ly;
    bool dot = false;
    if (x < v.w && y < v.h) {
      const int xa = max(x - v.rad, 0), xb = min(x + v.rad + 1, v.w), ya = max(y - v.rad, 0), yb = min(y + v.rad + 1, v.h);
      // (unsigned, then widened: the wrap-around of the running sums cancels modulo 2^32)
      const unsigned tot_u = v.S[(size_t)yb * W1 + xb] - v.S[(size_t)ya * W1 + xb] - v.S[(size_t)yb * W1 + xa] + v.S[(size_t)ya * W1 + xa];
      const long long tot = (long long)tot_u;
      const double area = (double)((xb - xa) * (yb - ya));
      unsigned val = v.img[(size_t)y * v.pitch + x];
      if (!v.black_on_white) val = 255 - val;
      dot = (double)val * area < v.thr * (double)tot;
    }
    L[i] = dot ? i : -1;
    s_area[i] = 0; s_x0[i] = v.w; s_x1[i] = -1; s_y0[i] = v.h; s_y1[i] = -1;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {       // (a label's sign never changes: the neighbour tests need no ordering against the links)
    const int i = threadIdx.x + 256 * k, lx = i & (kTW - 1), ly = i / kTW;
    if (L[i] < 0) continue;
    if (lx + 1 < kTW && L[i + 1] >= 0) det_unite(L, i, i + 1);
    if (ly + 1 < kTH && L[i + kTW] >= 0) det_unite(L, i, i + kTW);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = threadIdx.x + 256 * k, lx = i & (kTW - 1), ly = i / kTW, x = tx0 + lx, y = ty0 + ly;
    root[k] = -1;
    if (L[i] < 0) continue;
    const int r = det_find(L, i);
    root[k] = r;
    atomicAdd(&s_area[r], 1);
    atomicMin(&s_x0[r], x); atomicMax(&s_x1[r], x + 1); atomicMin(&s_y0[r], y); atomicMax(&s_y1[r], y + 1);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = threadIdx.x + 256 * k, lx = i & (kTW - 1), ly = i / kTW, x = tx0 + lx, y = ty0 + ly;
    if (x >= v.w || y >= v.h) continue;
    const int p = y * v.w + x, r = root[k];
    v.lab[p] = r >= 0 ? (ty0 + r / kTW) * v.w + tx0 + (r & (kTW - 1)) : -1;
    const bool is_root = r == i;
    v.area[p] = is_root ? s_area[i] : 0;              // (area > 0 marks a tile-local root: only there are the box entries meaningful)
    if (is_root) { v.x0[p] = s_x0[i]; v.x1[p] = s_x1[i]; v.y0[p] = s_y0[i]; v.y1[p] = s_y1[i]; }
  }
}
// the pairs of dot pixels across a tile's right and bottom edge: 64 threads per tile
__global__ __launch_bounds__(64) void k_det_border(DetView v) {
  const int tiles_x = (v.w + kTW - 1) / kTW;
  const int tx0 = (blockIdx.x % tiles_x) * kTW, ty0 = (blockIdx.x / tiles_x) * kTH;
  const int t = threadIdx.x;
  if (t < kTH) {
    const int x = tx0 + kTW - 1, y = ty0 + t;
    if (x + 1 < v.w && y < v.h) { const int p = y * v.w + x; if (v.lab[p] >= 0 && v.lab[p + 1] >= 0) det_unite(v.lab, p, p + 1); }
  } else {
    const int x = tx0 + (t - kTH), y = ty0 + kTH - 1;
    if (x < v.w && y + 1 < v.h) { const int p = y * v.w + x; if (v.lab[p] >= 0 && v.lab[p + v.w] >= 0) det_unite(v.lab, p, p + v.w); }
  }
}
// a piece (tile-local root) that is not the component's root hands its statistics to the root
__global__ __launch_bounds__(256) void k_det_push(DetView v) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= v.w * v.h) return;
  const int l = v.lab[p];
  if (l < 0 || l == p) return;
  const int a = v.area[p];                            // (l != p: nobody adds to this entry)
  if (a <= 0) return;
  const int r = det_find(v.lab, p);
  atomicAdd(&v.area[r], a);
  atomicMin(&v.x0[r], v.x0[p]); atomicMax(&v.x1[r], v.x1[p]); atomicMin(&v.y0[r], v.y0[p]); atomicMax(&v.y1[r], v.y1[p]);
}
__global__ __launch_bounds__(256) void k_det_select(DetView v) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= v.w * v.h || v.lab[p] != p) return;
  const int area = v.area[p], bw = v.x1[p] - v.x0[p], bh = v.y1[p] - v.y0[p];
  if ((double)area < v.min_area || (double)area < v.min_density * (double)(bw * bh)) return;
  if ((double)min(bw, bh) < v.min_aspect * (double)max(bw, bh)) return;
  if (v.x0[p] - kGrow < 1 || v.y0[p] - kGrow < 1 || v.x1[p] + kGrow > v.w - 1 || v.y1[p] + kGrow > v.h - 1) return;
  const int slot = atomicAdd(v.n_cand, 1);
  if (slot < v.max_cand) v.cand[slot] = p;
}
// the candidates ranked by label (= smallest pixel index of the component: a deterministic order); labels are distinct
__global__ __launch_bounds__(256) void k_det_order(DetView v) {
  __shared__ int sc[kMaxCand];
  const int n = min(*v.n_cand, v.max_cand), i = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= n) return;                  // (uniform over the workgroup)
  for (int j = threadIdx.x; j < n; j += 256) sc[j] = v.cand[j];
  __syncthreads();
  if (i >= n) return;
  const int mine = sc[i];
  int rank = 0;
  for (int j = 0; j < n; ++j) rank += (sc[j] < mine) ? 1 : 0;
  v.cand_sorted[rank] = mine;
}
__device__ __forceinline__ double det_wave_sum(double x) {       // all lanes, fixed order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
// dual conic through the box of candidate c (Ouellet & Hebert): sum over pixels of |g|^2 K K^T theta = -|g|^2 K c^2 with
// l = (g_x, g_y, c), c = -g . (x - box centre), K = (a^2, a b, b^2, a c, b c); centre = box centre + (theta_3, theta_4) / 2
__global__ __launch_bounds__(64) void k_det_fit(DetView v) {
  const int n = min(*v.n_cand, v.max_cand), lane = threadIdx.x;
  for (int ci = blockIdx.x; ci < n; ci += gridDim.x) {
  const int r = v.cand_sorted[ci];
  const int x0 = v.x0[r] - kGrow, x1 = v.x1[r] + kGrow, y0 = v.y0[r] - kGrow, y1 = v.y1[r] + kGrow;
  const double cx = 0.5 * (double)(x0 + x1 - 1), cy = 0.5 * (double)(y0 + y1 - 1);
  const int bw = x1 - x0, npx = bw * (y1 - y0);
  double acc[20];
#pragma unroll
  for (int k = 0; k < 20; ++k) acc[k] = 0.0;
  for (int q = lane; q < npx; q += 64) {
    const int y = y0 + q / bw, x = x0 + q % bw;
    const unsigned char* row = v.img + (size_t)y * v.pitch;
    double a = 0.5 * ((double)row[x + 1] - (double)row[x - 1]);
    double b = 0.5 * ((double)v.img[(size_t)(y + 1) * v.pitch + x] - (double)v.img[(size_t)(y - 1) * v.pitch + x]);
    if (!v.black_on_white) { a = -a; b = -b; }
    const double w2 = a * a + b * b;
    if (w2 < kMinGrad2) continue;
    const double c = -(a * ((double)x - cx) + b * ((double)y - cy));
    const double K[5] = {a * a, a * b, b * b, a * c, b * c};
    int e = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = i; j < 5; ++j) acc[e++] += w2 * K[i] * K[j];
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[15 + i] += w2 * K[i] * (-(c * c));
  }
#pragma unroll
  for (int k = 0; k < 20; ++k) acc[k] = det_wave_sum(acc[k]);
  if (lane == 0) {
    double M[5][6];
    int e = 0;
    for (int i = 0; i < 5; ++i) for (int j = i; j < 5; ++j) { M[i][j] = acc[e]; M[j][i] = acc[e]; ++e; }
    for (int i = 0; i < 5; ++i) M[i][5] = acc[15 + i];
    bool ok = true;
    for (int c = 0; c < 5; ++c) {                   // Gaussian elimination with partial pivoting
      int p = c;
      for (int rr = c + 1; rr < 5; ++rr) if (fabs(M[rr][c]) > fabs(M[p][c])) p = rr;
      if (M[p][c] == 0.0) { ok = false; break; }
      if (p != c) for (int j = 0; j < 6; ++j) { const double t = M[p][j]; M[p][j] = M[c][j]; M[c][j] = t; }
      for (int rr = c + 1; rr < 5; ++rr) {
        const double f = M[rr][c] / M[c][c];
        for (int j = c; j < 6; ++j) M[rr][j] -= f * M[c][j];
      }
    }
    double th[5] = {0, 0, 0, 0, 0};
    if (ok) for (int i = 4; i >= 0; --i) { double s = M[i][5]; for (int j = i + 1; j < 5; ++j) s -= M[i][j] * th[j]; th[i] = s / M[i][i]; }
    DetRec& rec = v.recs[ci];
    rec.cx = ok ? cx + 0.5 * th[3] : -1.0;
    rec.cy = ok ? cy + 0.5 * th[4] : -1.0;
    rec.box[0] = v.x0[r]; rec.box[1] = v.y0[r]; rec.box[2] = v.x1[r] - 1; rec.box[3] = v.y1[r] - 1;
    {
      // the fitted dual conic [[A, B/2, D/2], [B/2, C, E/2], [D/2, E/2, 1]] lives in box-centred coordinates: moved to image
      // coordinates (T Q T^T, T = translation by the box centre), inverted (adjugate: the scale of a conic is free) and scaled
      // to unit Frobenius norm with a positive first entry -- what calibu::Conic keeps as `C` (`Dual` is its inverse)
      const double A = th[0], B = th[1], Cc = th[2], Dd = th[3], E = th[4];
      const double q00 = A + cx * Dd + cx * cx, q01 = 0.5 * B + 0.5 * (cx * E + cy * Dd) + cx * cy, q11 = Cc + cy * E + cy * cy;
      const double q02 = 0.5 * Dd + cx, q12 = 0.5 * E + cy, q22 = 1.0;
      double c00 = q11 * q22 - q12 * q12, c01 = q02 * q12 - q01 * q22, c02 = q01 * q12 - q02 * q11;
      double c11 = q00 * q22 - q02 * q02, c12 = q01 * q02 - q00 * q12, c22 = q00 * q11 - q01 * q01;
      const double nrm = sqrt(c00 * c00 + c11 * c11 + c22 * c22 + 2.0 * (c01 * c01 + c02 * c02 + c12 * c12));
      const double sc = (ok && nrm > 0.0) ? (c00 < 0.0 ? -1.0 : 1.0) / nrm : 0.0;
      double* o = rec.conic;
      o[0] = sc * c00; o[1] = sc * c01; o[2] = sc * c02; o[3] = sc * c01; o[4] = sc * c11; o[5] = sc * c12; o[6] = sc * c02; o[7] = sc * c12; o[8] = sc * c22;
    }
  }
  }
}

}  // namespace

struct vc_detector {
  int device = 0, w = 0, h = 0;
  hipStream_t stream = nullptr;
  unsigned char* d_img = nullptr;
  unsigned* d_S = nullptr;
  int* d_lab = nullptr; int* d_stats = nullptr; int* d_cand = nullptr; int* d_cand_sorted = nullptr;
  unsigned char* d_out = nullptr;      // [count | records]
  unsigned char* h_out = nullptr;      // pinned mirror of d_out
  unsigned char* h_img = nullptr;      // pinned staging of the image (rows packed): one asynchronous copy instead of the runtime's pageable path
  int max_cand = kMaxCand;
  bool in_flight = false;              // a call is in progress or left through an error path: the stream may still read the staging buffers
  // calibu::ImageProcessing / ConicFinder parameters as VicalibTask sets them (vicalib-task.cc:116-122)
  int black_on_white = 1;
  double at_threshold = 0.9, at_window_ratio = 30.0, conic_min_area = 4.0, conic_min_density = 0.6, conic_min_aspect = 0.2;
};

extern "C" {

int vc_detector_create(int device, int width, int height, vc_detector** out) {
  if (!out || width < 8 || height < 8 || (long long)width * height > (1 << 26)) return VC_ERR_BAD_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return VC_ERR_NO_DEVICE;      // no CPU fallback
  if (hipSetDevice(device) != hipSuccess) return VC_ERR_NO_DEVICE;
  vc_detector* d = new vc_detector;
  d->device = device; d->w = width; d->h = height;
  const size_t np = (size_t)width * height, out_bytes = kHeadBytes + (size_t)d->max_cand * sizeof(DetRec);
  bool ok = hipStreamCreate(&d->stream) == hipSuccess && hipMalloc((void**)&d->d_img, np) == hipSuccess &&
            hipMalloc((void**)&d->d_S, (size_t)(width + 1) * (height + 1) * 4) == hipSuccess && hipMalloc((void**)&d->d_lab, np * 4) == hipSuccess &&
            hipMalloc((void**)&d->d_stats, np * 4 * 5) == hipSuccess && hipMalloc((void**)&d->d_cand, (size_t)d->max_cand * 4) == hipSuccess &&
            hipMalloc((void**)&d->d_cand_sorted, (size_t)d->max_cand * 4) == hipSuccess && hipMalloc((void**)&d->d_out, out_bytes) == hipSuccess &&
            hipHostMalloc((void**)&d->h_out, out_bytes, hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc((void**)&d->h_img, np, hipHostMallocDefault) == hipSuccess;
  if (!ok) { vc_detector_destroy(d); return VC_ERR_NO_DEVICE; }
  *out = d;
  return VC_OK;
}
void vc_detector_destroy(vc_detector* d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  (void)hipFree(d->d_img); (void)hipFree(d->d_S); (void)hipFree(d->d_lab); (void)hipFree(d->d_stats); (void)hipFree(d->d_cand);
  (void)hipFree(d->d_cand_sorted); (void)hipFree(d->d_out);
  if (d->h_out) (void)hipHostFree(d->h_out);
  if (d->h_img) (void)hipHostFree(d->h_img);
  delete d;
}
int vc_detector_set_params(vc_detector* d, int black_on_white, double at_threshold, double at_window_ratio, double conic_min_area,
                           double conic_min_density, double conic_min_aspect) {
  if (!d || !(at_threshold > 0.0) || !(at_window_ratio >= 1.0) || conic_min_area < 0.0 || conic_min_density < 0.0 || conic_min_aspect < 0.0) return VC_ERR_BAD_ARG;
  d->black_on_white = black_on_white ? 1 : 0; d->at_threshold = at_threshold; d->at_window_ratio = at_window_ratio;
  d->conic_min_area = conic_min_area; d->conic_min_density = conic_min_density; d->conic_min_aspect = conic_min_aspect;
  return VC_OK;
}
int vc_detector_find_conics(vc_detector* d, const unsigned char* image, int pitch, double* centres, double* conics, int* boxes, int max_conics,
                            int* n_found) {
  if (!d || !image || pitch < d->w || !n_found || max_conics < 0 || (max_conics > 0 && !centres)) return VC_ERR_BAD_ARG;
  if (hipSetDevice(d->device) != hipSuccess) return VC_ERR_NO_DEVICE;
  const int w = d->w, h = d->h, np = w * h;
  // A detector is SINGLE-THREADED: one image at a time per handle (the staging buffers and the stream belong to the call in progress) --
  // one handle per thread.  A clean call has synchronised on its download, the staging buffers are free; a call that left through an
  // error path may have copies in flight that still read them: it leaves the handle marked and this call drains the stream first
  // (advice round 5).
  if (d->in_flight) (void)hipStreamSynchronize(d->stream);
  d->in_flight = true;
  if (pitch == w) std::memcpy(d->h_img, image, (size_t)np);
  else for (int y = 0; y < h; ++y) std::memcpy(d->h_img + (size_t)y * w, image + (size_t)y * pitch, (size_t)w);
  // (one copy: bands of 256 KB, each sent while the host packs the next, measured slower -- 8.7 k against 9.6 k images/s at 640 x 480)
  if (hipMemcpyAsync(d->d_img, d->h_img, (size_t)np, hipMemcpyHostToDevice, d->stream) != hipSuccess) return VC_ERR_NO_DEVICE;
  DetView v;
  v.w = w; v.h = h; v.pitch = w; v.img = d->d_img; v.S = d->d_S; v.lab = d->d_lab;
  v.area = d->d_stats; v.x0 = d->d_stats + np; v.x1 = d->d_stats + 2 * (size_t)np; v.y0 = d->d_stats + 3 * (size_t)np; v.y1 = d->d_stats + 4 * (size_t)np;
  v.cand = d->d_cand; v.cand_sorted = d->d_cand_sorted; v.n_cand = reinterpret_cast<int*>(d->d_out); v.max_cand = d->max_cand;
  v.thr = d->at_threshold; v.rad = (int)((double)w / d->at_window_ratio);
  v.min_area = d->conic_min_area; v.min_density = d->conic_min_density; v.min_aspect = d->conic_min_aspect;
  v.black_on_white = d->black_on_white; v.recs = reinterpret_cast<DetRec*>(d->d_out + kHeadBytes);
  const int gb = (np + 255) / 256, tiles = ((w + kTW - 1) / kTW) * ((h + kTH - 1) / kTH);
  hipLaunchKernelGGL(k_det_rows, dim3(h), dim3(64), 0, d->stream, v);                  // (also zeroes the candidate count)
  hipLaunchKernelGGL(k_det_cols, dim3((w + 1 + 63) / 64), dim3(1024), 0, d->stream, v);
  hipLaunchKernelGGL(k_det_tile, dim3(tiles), dim3(256), 0, d->stream, v);
  hipLaunchKernelGGL(k_det_border, dim3(tiles), dim3(64), 0, d->stream, v);
  hipLaunchKernelGGL(k_det_push, dim3(gb), dim3(256), 0, d->stream, v);
  hipLaunchKernelGGL(k_det_select, dim3(gb), dim3(256), 0, d->stream, v);
  hipLaunchKernelGGL(k_det_order, dim3((d->max_cand + 255) / 256), dim3(256), 0, d->stream, v);
  hipLaunchKernelGGL(k_det_fit, dim3(1024), dim3(64), 0, d->stream, v);
  // the count and the first records in one copy; a second one only for an image with more than kFastRecs dots
  const size_t fast_bytes = kHeadBytes + (size_t)std::min(kFastRecs, d->max_cand) * sizeof(DetRec);
  if (hipMemcpyAsync(d->h_out, d->d_out, fast_bytes, hipMemcpyDeviceToHost, d->stream) != hipSuccess || hipStreamSynchronize(d->stream) != hipSuccess) return VC_ERR_NO_DEVICE;
  const int n = *reinterpret_cast<const int*>(d->h_out);
  if (n > d->max_cand) { d->in_flight = false; return VC_ERR_UNSUPPORTED; }      // (synchronised: nothing pending)
  *n_found = n;
  const int nout = std::min(n, max_conics);
  if (nout > kFastRecs) {
    if (hipMemcpyAsync(d->h_out + fast_bytes, d->d_out + fast_bytes, (size_t)(nout - kFastRecs) * sizeof(DetRec), hipMemcpyDeviceToHost, d->stream) != hipSuccess ||
        hipStreamSynchronize(d->stream) != hipSuccess) return VC_ERR_NO_DEVICE;
  }
  d->in_flight = false;                          // (everything this call queued has completed)
  const DetRec* recs = reinterpret_cast<const DetRec*>(d->h_out + kHeadBytes);
  for (int i = 0; i < nout; ++i) {
    centres[2 * i] = recs[i].cx; centres[2 * i + 1] = recs[i].cy;
    if (conics) std::memcpy(conics + 9 * (size_t)i, recs[i].conic, 72);
    if (boxes) std::memcpy(boxes + 4 * (size_t)i, recs[i].box, 16);
  }
  return VC_OK;
}
int vc_detector_find(vc_detector* d, const unsigned char* image, int pitch, double* centres, int max_conics, int* n_found) {
  return vc_detector_find_conics(d, image, pitch, centres, nullptr, nullptr, max_conics, n_found);
}

}  // extern "C"
