// Stand-alone HBM calibration (not part of the product): what a hand-written streaming kernel sustains on
// this box for read-only, write-only and copy, at several launch shapes.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/bw_read tools/bw_read.hip && /tmp/bw_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float vf4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_read(const vf4* __restrict__ p, size_t n, float* sink) {
  vf4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    vf4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  for (; i < n; i += stride) acc += p[i];
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) *sink = 1.f;
}
template <bool NT>
__global__ void __launch_bounds__(256) k_write(vf4* __restrict__ p, size_t n) {
  const vf4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
  }
}
__global__ void __launch_bounds__(256) k_copy(const vf4* __restrict__ s, vf4* __restrict__ d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
}
template <typename F>
static double timeit(F f, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int r = 0; r < reps; ++r) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps * 1e-3;
}
int main() {
  const size_t bytes = (size_t)4 << 30, n = bytes / 16;   // 4 GiB: well past the 256 MB Infinity Cache
  vf4 *a, *b; float* sink;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 4);
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  for (int blocks : {2048, 8192, 32768, 131072}) {
    double t;
    t = timeit([&] { k_read<1, false><<<blocks, 256>>>(a, n, sink); }, 5); printf("blocks %6d read u1      %6.2f TB/s\n", blocks, bytes / t / 1e12);
    t = timeit([&] { k_read<4, false><<<blocks, 256>>>(a, n, sink); }, 5); printf("blocks %6d read u4      %6.2f TB/s\n", blocks, bytes / t / 1e12);
    t = timeit([&] { k_read<8, true><<<blocks, 256>>>(a, n, sink); }, 5);  printf("blocks %6d read u8 nt   %6.2f TB/s\n", blocks, bytes / t / 1e12);
    t = timeit([&] { k_write<false><<<blocks, 256>>>(b, n); }, 5);         printf("blocks %6d write        %6.2f TB/s\n", blocks, bytes / t / 1e12);
    t = timeit([&] { k_write<true><<<blocks, 256>>>(b, n); }, 5);          printf("blocks %6d write nt     %6.2f TB/s\n", blocks, bytes / t / 1e12);
    t = timeit([&] { k_copy<<<blocks, 256>>>(a, b, n); }, 5);              printf("blocks %6d copy (r+w)   %6.2f TB/s\n", blocks, 2.0 * bytes / t / 1e12);
  }
  // the sizes the hot path moves per launch: 128 MB (one batch of rows)
  const size_t nb = ((size_t)128 << 20) / 16;
  double t = timeit([&] { k_read<4, false><<<8192, 256>>>(a, nb, sink); }, 20); printf("128 MB read  (cache-warm) %6.2f TB/s  %.1f us\n", (nb * 16) / t / 1e12, t * 1e6);
  t = timeit([&] { k_write<true><<<8192, 256>>>(b, nb); }, 20);                 printf("128 MB write nt          %6.2f TB/s  %.1f us\n", (nb * 16) / t / 1e12, t * 1e6);
  return 0;
}
