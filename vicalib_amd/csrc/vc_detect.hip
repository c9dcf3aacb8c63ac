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
// Data flow on the device (640 x 480: every kernel is a few hundred wavefronts; the slice is about correctness and the boundary):
//   k_det_rows / k_det_cols   integral image S (uint32, exact)
//   k_det_threshold           dot mask = (I * window_area < at_threshold * window_sum); label[p] = p for dot pixels, -1 otherwise
//   k_det_merge / k_det_flatten   union-find with atomicMin links to the smaller index: every component is named by its smallest
//                             pixel index (what the restatement's fixed-point iteration converges to)
//   k_det_stats               per component: area, bounding box (integer atomics: order-independent)
//   k_det_select              roots that pass area / density / aspect / border tests -> candidate list (sorted on the host: the
//                             order of atomic appends is not deterministic, the sorted list is)
//   k_det_fit                 one wavefront per candidate: the 5 x 5 normal equations of the dual conic over the box's pixels
//                             (lane-strided, fixed-order wave sums), solved by lane 0
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <vector>
#include "../../include/vicalib_amd.h"

namespace {

constexpr int kGrow = 2;              // pixels the bounding box is grown by before the fit   (the restatement's GROW)
constexpr double kMinGrad2 = 1.0;     // squared gradient magnitude below which a pixel is ignored (MIN_GRAD2)

struct DetView {
  int w, h, pitch;
  const unsigned char* img;      // h x pitch
  unsigned* S;                   // (h + 1) x (w + 1) integral image
  int* lab;                      // h x w
  int* area; int* x0; int* x1; int* y0; int* y1;      // per root pixel
  int* cand; int* n_cand; int max_cand;
  double thr; int rad;
  double min_area, min_density, min_aspect;
  int black_on_white;
  double* centres;
  double* conics;                  // 9 per candidate (primal conic, image coordinates) or nullptr
  int* boxes;                      // 4 per candidate (x0, y0, x1, y1 inclusive: the component's bounding box) or nullptr
};

// integral image, rows: one wavefront per row, eight consecutive pixels per lane and round, exclusive scan of the lane sums.
// The sums are kept modulo 2^32 (images above 16.8 MP would overflow a 32-bit total): every use takes the four-corner difference in
// unsigned arithmetic, which is exact as long as the WINDOW's sum fits -- it does, a window has at most (2 rad + 1)^2 pixels
__global__ __launch_bounds__(64) void k_det_rows(DetView v) {
  const int y = blockIdx.x, lane = threadIdx.x;
  unsigned carry = 0;
  unsigned* row = v.S + (size_t)(y + 1) * (v.w + 1);
  if (lane == 0) row[0] = 0;
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
// ... columns: one thread per column, running sum down the rows (coalesced across the wavefront)
__global__ __launch_bounds__(64) void k_det_cols(DetView v) {
  const int x = blockIdx.x * 64 + threadIdx.x;
  if (x > v.w) return;
  unsigned s = 0;
  v.S[x] = 0;
  for (int y = 1; y <= v.h; ++y) { s += v.S[(size_t)y * (v.w + 1) + x]; v.S[(size_t)y * (v.w + 1) + x] = s; }
}
__global__ __launch_bounds__(256) void k_det_threshold(DetView v) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= v.w * v.h) return;
  const int y = p / v.w, x = p % v.w;
  const int xa = max(x - v.rad, 0), xb = min(x + v.rad + 1, v.w), ya = max(y - v.rad, 0), yb = min(y + v.rad + 1, v.h);
  const int W1 = v.w + 1;
  // (unsigned, then widened: the wrap-around of the running sums cancels modulo 2^32)
  const unsigned tot_u = v.S[(size_t)yb * W1 + xb] - v.S[(size_t)ya * W1 + xb] - v.S[(size_t)yb * W1 + xa] + v.S[(size_t)ya * W1 + xa];
  const long long tot = (long long)tot_u;
  const double area = (double)((xb - xa) * (yb - ya));
  unsigned val = v.img[(size_t)y * v.pitch + x];
  if (!v.black_on_white) val = 255 - val;
  const bool dot = (double)val * area < v.thr * (double)tot;
  v.lab[p] = dot ? p : -1;
  v.area[p] = 0; v.x0[p] = v.w; v.x1[p] = -1; v.y0[p] = v.h; v.y1[p] = -1;
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
__global__ __launch_bounds__(256) void k_det_merge(DetView v) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= v.w * v.h || v.lab[p] < 0) return;
  const int y = p / v.w, x = p % v.w;
  if (x + 1 < v.w && v.lab[p + 1] >= 0) det_unite(v.lab, p, p + 1);
  if (y + 1 < v.h && v.lab[p + v.w] >= 0) det_unite(v.lab, p, p + v.w);
}
__global__ __launch_bounds__(256) void k_det_flatten(DetView v) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= v.w * v.h || v.lab[p] < 0) return;
  v.lab[p] = det_find(v.lab, p);
}
__global__ __launch_bounds__(256) void k_det_stats(DetView v) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= v.w * v.h) return;
  const int r = v.lab[p];
  if (r < 0) return;
  const int y = p / v.w, x = p % v.w;
  atomicAdd(&v.area[r], 1);
  atomicMin(&v.x0[r], x); atomicMax(&v.x1[r], x + 1); atomicMin(&v.y0[r], y); atomicMax(&v.y1[r], y + 1);
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
__device__ __forceinline__ double det_wave_sum(double x) {       // all lanes, fixed order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
// dual conic through the box of candidate c (Ouellet & Hebert): sum over pixels of |g|^2 K K^T theta = -|g|^2 K c^2 with
// l = (g_x, g_y, c), c = -g . (x - box centre), K = (a^2, a b, b^2, a c, b c); centre = box centre + (theta_3, theta_4) / 2
__global__ __launch_bounds__(64) void k_det_fit(DetView v, int n) {
  const int ci = blockIdx.x, lane = threadIdx.x;
  if (ci >= n) return;
  const int r = v.cand[ci];
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
    v.centres[2 * ci] = ok ? cx + 0.5 * th[3] : -1.0;
    v.centres[2 * ci + 1] = ok ? cy + 0.5 * th[4] : -1.0;
    if (v.boxes) { v.boxes[4 * ci] = v.x0[r]; v.boxes[4 * ci + 1] = v.y0[r]; v.boxes[4 * ci + 2] = v.x1[r] - 1; v.boxes[4 * ci + 3] = v.y1[r] - 1; }
    if (v.conics) {
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
      double* o = v.conics + 9 * (size_t)ci;
      o[0] = sc * c00; o[1] = sc * c01; o[2] = sc * c02; o[3] = sc * c01; o[4] = sc * c11; o[5] = sc * c12; o[6] = sc * c02; o[7] = sc * c12; o[8] = sc * c22;
    }
  }
}

}  // namespace

struct vc_detector {
  int device = 0, w = 0, h = 0;
  hipStream_t stream = nullptr;
  unsigned char* d_img = nullptr;
  unsigned* d_S = nullptr;
  int* d_lab = nullptr; int* d_stats = nullptr; int* d_cand = nullptr; int* d_ncand = nullptr;
  double* d_centres = nullptr;
  double* d_conics = nullptr;
  int* d_boxes = nullptr;
  int max_cand = 4096;
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
  const size_t np = (size_t)width * height;
  bool ok = hipStreamCreate(&d->stream) == hipSuccess && hipMalloc((void**)&d->d_img, np) == hipSuccess &&
            hipMalloc((void**)&d->d_S, (size_t)(width + 1) * (height + 1) * 4) == hipSuccess && hipMalloc((void**)&d->d_lab, np * 4) == hipSuccess &&
            hipMalloc((void**)&d->d_stats, np * 4 * 5) == hipSuccess && hipMalloc((void**)&d->d_cand, (size_t)d->max_cand * 4) == hipSuccess &&
            hipMalloc((void**)&d->d_ncand, 4) == hipSuccess && hipMalloc((void**)&d->d_centres, (size_t)d->max_cand * 16) == hipSuccess &&
            hipMalloc((void**)&d->d_conics, (size_t)d->max_cand * 72) == hipSuccess && hipMalloc((void**)&d->d_boxes, (size_t)d->max_cand * 16) == hipSuccess;
  if (!ok) { vc_detector_destroy(d); return VC_ERR_NO_DEVICE; }
  *out = d;
  return VC_OK;
}
void vc_detector_destroy(vc_detector* d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  (void)hipFree(d->d_img); (void)hipFree(d->d_S); (void)hipFree(d->d_lab); (void)hipFree(d->d_stats); (void)hipFree(d->d_cand);
  (void)hipFree(d->d_ncand); (void)hipFree(d->d_centres); (void)hipFree(d->d_conics); (void)hipFree(d->d_boxes);
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
  if (hipMemcpy2DAsync(d->d_img, (size_t)w, image, (size_t)pitch, (size_t)w, (size_t)h, hipMemcpyHostToDevice, d->stream) != hipSuccess) return VC_ERR_NO_DEVICE;
  if (hipMemsetAsync(d->d_ncand, 0, 4, d->stream) != hipSuccess) return VC_ERR_NO_DEVICE;
  DetView v;
  v.w = w; v.h = h; v.pitch = w; v.img = d->d_img; v.S = d->d_S; v.lab = d->d_lab;
  v.area = d->d_stats; v.x0 = d->d_stats + np; v.x1 = d->d_stats + 2 * (size_t)np; v.y0 = d->d_stats + 3 * (size_t)np; v.y1 = d->d_stats + 4 * (size_t)np;
  v.cand = d->d_cand; v.n_cand = d->d_ncand; v.max_cand = d->max_cand;
  v.thr = d->at_threshold; v.rad = (int)((double)w / d->at_window_ratio);
  v.min_area = d->conic_min_area; v.min_density = d->conic_min_density; v.min_aspect = d->conic_min_aspect;
  v.black_on_white = d->black_on_white; v.centres = d->d_centres; v.conics = conics ? d->d_conics : nullptr; v.boxes = boxes ? d->d_boxes : nullptr;
  const int gb = (np + 255) / 256;
  hipLaunchKernelGGL(k_det_rows, dim3(h), dim3(64), 0, d->stream, v);
  hipLaunchKernelGGL(k_det_cols, dim3((w + 1 + 63) / 64), dim3(64), 0, d->stream, v);
  hipLaunchKernelGGL(k_det_threshold, dim3(gb), dim3(256), 0, d->stream, v);
  hipLaunchKernelGGL(k_det_merge, dim3(gb), dim3(256), 0, d->stream, v);
  hipLaunchKernelGGL(k_det_flatten, dim3(gb), dim3(256), 0, d->stream, v);
  hipLaunchKernelGGL(k_det_stats, dim3(gb), dim3(256), 0, d->stream, v);
  hipLaunchKernelGGL(k_det_select, dim3(gb), dim3(256), 0, d->stream, v);
  int n = 0;
  if (hipMemcpyAsync(&n, d->d_ncand, 4, hipMemcpyDeviceToHost, d->stream) != hipSuccess || hipStreamSynchronize(d->stream) != hipSuccess) return VC_ERR_NO_DEVICE;
  if (n > d->max_cand) return VC_ERR_UNSUPPORTED;
  *n_found = n;
  if (n == 0) return VC_OK;
  std::vector<int> cand((size_t)n);
  if (hipMemcpy(cand.data(), d->d_cand, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) return VC_ERR_NO_DEVICE;
  std::sort(cand.begin(), cand.end());            // by label = smallest pixel index of the component: a deterministic order
  if (hipMemcpyAsync(d->d_cand, cand.data(), (size_t)n * 4, hipMemcpyHostToDevice, d->stream) != hipSuccess) return VC_ERR_NO_DEVICE;
  hipLaunchKernelGGL(k_det_fit, dim3(n), dim3(64), 0, d->stream, v, n);
  std::vector<double> out((size_t)n * 2);
  if (hipMemcpyAsync(out.data(), d->d_centres, out.size() * 8, hipMemcpyDeviceToHost, d->stream) != hipSuccess || hipStreamSynchronize(d->stream) != hipSuccess) return VC_ERR_NO_DEVICE;
  const int nout = std::min(n, max_conics);
  std::memcpy(centres, out.data(), (size_t)nout * 16);
  if (conics && nout > 0 && hipMemcpy(conics, d->d_conics, (size_t)nout * 72, hipMemcpyDeviceToHost) != hipSuccess) return VC_ERR_NO_DEVICE;
  if (boxes && nout > 0 && hipMemcpy(boxes, d->d_boxes, (size_t)nout * 16, hipMemcpyDeviceToHost) != hipSuccess) return VC_ERR_NO_DEVICE;
  return VC_OK;
}
int vc_detector_find(vc_detector* d, const unsigned char* image, int pitch, double* centres, int max_conics, int* n_found) {
  return vc_detector_find_conics(d, image, pitch, centres, nullptr, nullptr, max_conics, n_found);
}

}  // extern "C"
