// Probe (GPU box): what HBM WRITE bandwidth does an MI355X sustain, (a) for a plain streaming fill and (b) for the store pattern of
// the lane-wise kernels' dense output -- every wavefront owns 64 rows (trajectories) of N_t samples and writes them in steps of ~34
// samples per row, a store instruction covering runs of 4 consecutive 16-byte samples of 16 different rows?
//   hipcc --offload-arch=gfx950 -O3 hbm_write.hip -o hbm_write && ./hbm_write [rows] [nt] [lds bytes per workgroup]
// The lane-wise kernels are priced against the 8 TB/s peak (BASELINE.json); this probe says how much of that peak a kernel that does
// NOTHING but these stores reaches, i.e. where the practical ceiling of the pattern is.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// (a) streaming fill: consecutive lanes write consecutive 16 bytes, grid-stride
__global__ void fill(double2 *out, size_t n) {
  const double2 v = make_double2(1.0, 2.0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = v;
}

// (b) the dense-output pattern: wavefront w owns rows 64 w .. 64 w + 63 of [rows][nt] double2.  Per "attempt" every row advances by
// `step` samples; a pass serves 16 chunks of 8 samples: lane = 4 c + kk writes samples kk and kk + 4 of chunk c (the work-list
// emission's store shape).  Chunks are dealt out row after row.
template <int NT_STORE>
__global__ void __launch_bounds__(256) rows(double2 *out, int n_rows, int nt, int step) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tile = blockIdx.x * 4 + wv;
  if (tile * 64 >= n_rows) return;
  const int slot = lane >> 2, kk = lane & 3;
  const double2 v = make_double2(1.0, 2.0);
  const int cpr = (step + 7) / 8;   // chunks per row and attempt
  for (int o = 0; o < nt; o += step) {
    const int total = 64 * cpr;
    for (int c0 = 0; c0 < total; c0 += 16) {
      const int c = c0 + slot;
      if (c < total) {
        const int r = c / cpr, ch = c - r * cpr;
        double2 *row = out + (size_t)(tile * 64 + r) * nt;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int idx = o + ch * 8 + kk + 4 * u;
          if (idx < nt && idx < o + step) {
            if (NT_STORE) { __builtin_nontemporal_store(v.x, &row[idx].x); __builtin_nontemporal_store(v.y, &row[idx].y); }
            else row[idx] = v;
          }
        }
      }
    }
  }
}

// (c) the 6-state pattern: rows of [nt] samples of 48 bytes; a pass serves 32 chunks of 8 samples, two lanes per chunk.
//   mode 0 (the kernel's): lane kk writes samples kk, kk + 2, kk + 4, kk + 6 whole (3 x 16 bytes each): the two lanes of a chunk are
//           48 bytes apart in every store instruction -- 64 separate 16-byte pieces per instruction
//   mode 1: the chunk's 384 bytes as 24 pieces of 16; store instruction i writes pieces 2 i + kk: the two lanes write 32 contiguous bytes
__global__ void __launch_bounds__(256) rows6(double2 *out, int n_rows, int nt, int step, int mode) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tile = blockIdx.x * 4 + wv;
  if (tile * 64 >= n_rows) return;
  const int slot = lane >> 1, kk = lane & 1;
  const double2 v = make_double2(1.0, 2.0);
  const int cpr = (step + 7) / 8;
  for (int o = 0; o < nt; o += step) {
    const int total = 64 * cpr;
    for (int c0 = 0; c0 < total; c0 += 32) {
      const int c = c0 + slot;
      if (c < total) {
        const int r = c / cpr, ch = c - r * cpr;
        double2 *row = out + (size_t)(tile * 64 + r) * nt * 3;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          int smp, piece;
          if (mode == 0) { smp = kk + 2 * (i / 3); piece = i % 3; }
          else { const int p = 2 * i + kk; smp = p / 3; piece = p % 3; }
          const int idx = o + ch * 8 + smp;
          if (idx < nt && idx < o + step) row[(size_t)idx * 3 + piece] = v;
        }
      }
    }
  }
}

// (d) TIME-MAJOR output [nt][rows] (torchdiffeq's own layout): lane = trajectory, a wavefront writes 64 x 16 = 1024 contiguous bytes
// per sample index; lanes run ahead of / behind each other by up to `jitter` samples (adaptive steps), `step` samples per attempt
__global__ void __launch_bounds__(256) cols(double2 *out, int n_rows, int nt, int step, int jitter) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tile = blockIdx.x * 4 + wv;
  if (tile * 64 >= n_rows) return;
  const double2 v = make_double2(1.0, 2.0);
  const size_t col = (size_t)tile * 64 + lane;
  const int off = jitter > 0 ? (int)((lane * 2654435761u >> 16) % (unsigned)(jitter + 1)) : 0;   // this lane's lag
  int lo_prev = 0;
  for (int o = 0; o < nt + jitter; o += step) {
    // my samples of this attempt: [o - off, o - off + step) clipped to [lo_prev, nt)
    int lo = o - off, hi = o - off + step;
    if (lo < lo_prev) lo = lo_prev;
    if (hi > nt) hi = nt;
    const int kmin = o - jitter < 0 ? 0 : o - jitter, kmax = o + step < nt ? o + step : nt;   // wave-uniform bounds
    for (int k = kmin; k < kmax; ++k)
      if (k >= lo && k < hi) out[(size_t)k * n_rows + col] = v;
    if (hi > lo_prev) lo_prev = hi;
  }
}

int main(int argc, char **argv) {
  const int n_rows = argc > 1 ? atoi(argv[1]) : 393216, nt = argc > 2 ? atoi(argv[2]) : 20001;
  const size_t n = (size_t)n_rows * nt;
  const size_t lds = argc > 3 ? (size_t)atoi(argv[3]) : 0;   // dynamic LDS per workgroup of the row kernels: caps the workgroups per CU (e.g. 34816: 4 per CU = 16 wavefronts, the lean 2-state kernel's residency)
  double2 *buf;
  if (hipMalloc(&buf, n * sizeof(double2)) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  if (lds > 64 * 1024) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(rows<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void *>(rows<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void *>(rows6), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void *>(cols), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float ms;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    fill<<<256 * 16, 256>>>(buf, n);
    hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    printf("fill      %8.3f ms  %7.1f GB/s\n", ms, n * 16.0 / ms / 1e6);
  }
  for (int step : {34, 8, 64, 128}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      rows<0><<<(n_rows / 64 + 3) / 4, 256, lds>>>(buf, n_rows, nt, step);
      hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
      printf("rows s=%-3d %8.3f ms  %7.1f GB/s\n", step, ms, n * 16.0 / ms / 1e6);
    }
  }
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    rows<1><<<(n_rows / 64 + 3) / 4, 256, lds>>>(buf, n_rows, nt, 34);
    hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    printf("rows s=34 nontemporal %8.3f ms  %7.1f GB/s\n", ms, n * 16.0 / ms / 1e6);
  }
  for (int jitter : {0, 8, 24}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      cols<<<(n_rows / 64 + 3) / 4, 256, lds>>>(buf, n_rows, nt, 34, jitter);
      hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
      printf("cols s=34 jitter=%-2d %8.3f ms  %7.1f GB/s\n", jitter, ms, n * 16.0 / ms / 1e6);
    }
  }
  {
    const int r6 = n_rows / 3;   // the same bytes: 48-byte samples
    for (int mode = 0; mode < 2; ++mode)
      for (int step : {20, 8}) {
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(a);
          rows6<<<(r6 / 64 + 3) / 4, 256, lds>>>(buf, r6, nt, step, mode);
          hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
          printf("rows6 mode=%d s=%-3d %8.3f ms  %7.1f GB/s\n", mode, step, ms, (size_t)r6 * nt * 48.0 / ms / 1e6);
        }
      }
  }
  hipFree(buf);
  return 0;
}
