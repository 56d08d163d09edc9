// Micro-benchmark: how fast can a "one column per workgroup" kernel write a row-major
// [P][ld] complex128 array (each workgroup owns CPW adjacent columns and writes CPW*16 bytes to
// every row), depending on how the workgroups that complete one 128-byte line are placed
// (same XCD, consecutive dispatch) -- and the mirror-image read.  Decides whether a
// full-column fused pass (no second column sub-pass) is viable.  Build:
//   hipcc -O3 --offload-arch=gfx950 -o scatter_write scatter_write.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double2 cplx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// map block -> first column.  mode 0: naive.  mode 1: the 8/CPW workgroups sharing a line are
// consecutive blocks of ONE XCD (blocks go round-robin over 8 XCDs).
__device__ __forceinline__ int col0_of(int b, int cpw, int mode, int ncols) {
  if (mode == 0) return b * cpw;
  const int k = 8 / cpw;            // workgroups per 128-byte line
  const int xcd = b & 7, q = b >> 3;
  const int grp = q / k, sub = q - grp * k;
  return ((grp * 8 + xcd) * k + sub) * cpw;
}

template <int CPW>
__global__ void k_write(cplx* out, int P, int ld, int ncols, int mode) {
  const int c0 = col0_of(blockIdx.x, CPW, mode, ncols);
  if (c0 >= ncols) return;
  for (int r = threadIdx.x; r < P; r += blockDim.x) {
#pragma unroll
    for (int u = 0; u < CPW; ++u) out[(size_t)r * ld + c0 + u] = make_double2((double)r, (double)(c0 + u));
  }
}
template <int CPW>
__global__ void k_read(const cplx* in, double* sink, int P, int ld, int ncols, int mode) {
  const int c0 = col0_of(blockIdx.x, CPW, mode, ncols);
  if (c0 >= ncols) return;
  double acc = 0.0;
  for (int r = threadIdx.x; r < P; r += blockDim.x) {
#pragma unroll
    for (int u = 0; u < CPW; ++u) { const cplx v = in[(size_t)r * ld + c0 + u]; acc += v.x + v.y; }
  }
  if (acc == 12345.678) sink[0] = acc;
}
// reference: coalesced row-major streaming write / read of the same array
__global__ void k_stream_write(cplx* out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = make_double2((double)i, 1.0);
}
__global__ void k_stream_read(const cplx* in, double* sink, size_t n) {
  double acc = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const cplx v = in[i]; acc += v.x + v.y; }
  if (acc == 12345.678) sink[0] = acc;
}

template <typename F>
float timeit(F f, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const int P = 5184, H = P / 2 + 1, ld = (H + 7) & ~7;
  const size_t n = (size_t)P * ld;
  cplx *buf, *buf2; double* sink;
  CK(hipMalloc(&buf, n * sizeof(cplx))); CK(hipMalloc(&buf2, n * sizeof(cplx))); CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 0, n * sizeof(cplx)));
  const double mb = n * sizeof(cplx) / 1e6;
  printf("array %d x %d complex128 = %.1f MB\n", P, ld, mb);
  float t = timeit([&] { hipLaunchKernelGGL(k_stream_write, dim3(4096), dim3(256), 0, 0, buf, n); }, 10);
  printf("stream write            %.1f us  %.2f TB/s\n", t * 1e3, mb / t / 1e3);
  t = timeit([&] { hipLaunchKernelGGL(k_stream_read, dim3(4096), dim3(256), 0, 0, buf, sink, n); }, 10);
  printf("stream read             %.1f us  %.2f TB/s\n", t * 1e3, mb / t / 1e3);
  for (int thr : {256, 512}) for (int mode = 0; mode < 2; ++mode) {
#define RUN(CPW)                                                                                           \
    {                                                                                                      \
      const int nb = ((ld / CPW + 63) / 64) * 64;                                                          \
      t = timeit([&] { hipLaunchKernelGGL(k_write<CPW>, dim3(nb), dim3(thr), 0, 0, buf, P, ld, ld, mode); }, 10); \
      printf("col write cpw=%d thr=%d mode=%d  %.1f us  %.2f TB/s\n", CPW, thr, mode, t * 1e3, mb / t / 1e3);       \
      t = timeit([&] { hipLaunchKernelGGL(k_read<CPW>, dim3(nb), dim3(thr), 0, 0, buf, sink, P, ld, ld, mode); }, 10); \
      printf("col read  cpw=%d thr=%d mode=%d  %.1f us  %.2f TB/s\n", CPW, thr, mode, t * 1e3, mb / t / 1e3);       \
    }
    RUN(1) RUN(2) RUN(4) RUN(8)
  }
  // back-to-back: write column-wise then read row-wise (what the pipeline would do): does the
  // 256 MB memory-side cache help?
  t = timeit([&] {
    hipLaunchKernelGGL(k_write<1>, dim3(((ld + 63) / 64) * 64), dim3(512), 0, 0, buf, P, ld, ld, 1);
    hipLaunchKernelGGL(k_stream_read, dim3(4096), dim3(256), 0, 0, buf, sink, n); }, 10);
  printf("col write cpw=1 + stream read  %.1f us\n", t * 1e3);
  return 0;
}
