// Round-3 calibration (hipcc -O3 --offload-arch=gfx950 -o /tmp/calib_r03 tools/calib_r03.hip): which access
// patterns of the lookup + apply step cost what on this box.
//   A  1 M random 128-B row reads against footprints of 128 MB .. 32 GB (is the random rate a DRAM or a
//      translation effect?)
//   B  the same 1 M rows read tile-locally: block b reads rows [2048 b, 2048 b + 2048) in a permuted order
//   C  the per-key state update of GroupAdam: U keys x {var row 128 B, slot row 384 B, record 16 B} read + written
//   D  U random 16-B index probes into 2 GB
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned long long mix(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}

// A: LPR = 8 lanes per row, INF rows in flight per group
template <int INF>
__global__ void __launch_bounds__(256) k_rand(const float4* tab, unsigned long long nrows, unsigned long long nacc, float* sink, unsigned long long salt) {
  const int lane = threadIdx.x & 7;
  const unsigned long long grp = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) >> 3;
  const unsigned long long ngrp = (unsigned long long)gridDim.x * 32;
  float acc = 0.f;
  for (unsigned long long i = grp * INF; i < nacc; i += ngrp * INF) {
    float4 v[INF];
#pragma unroll
    for (int j = 0; j < INF; ++j) { const unsigned long long r = mix((i + j) * 0x9E3779B97F4A7C15ULL + salt) % nrows; v[j] = tab[r * 8 + lane]; }
#pragma unroll
    for (int j = 0; j < INF; ++j) acc += v[j].x;
  }
  if (acc == 123.456f) *sink = acc;
}
// B: tile-local permuted order (odd multiplier mod 2048 inside the block's window)
template <int INF>
__global__ void __launch_bounds__(256) k_tile_local(const float4* tab, unsigned long long nacc, float* sink, unsigned mul) {
  const int lane = threadIdx.x & 7, g = threadIdx.x >> 3;   // 32 groups per block
  float acc = 0.f;
  for (unsigned long long base = (unsigned long long)blockIdx.x * 2048; base < nacc; base += (unsigned long long)gridDim.x * 2048) {
    for (int j0 = g * INF; j0 < 2048; j0 += 32 * INF) {
      float4 v[INF];
#pragma unroll
      for (int j = 0; j < INF; ++j) { const unsigned r = ((unsigned)(j0 + j) * mul + 977u) & 2047u; v[j] = tab[(base + r) * 8 + lane]; }
#pragma unroll
      for (int j = 0; j < INF; ++j) acc += v[j].x;
    }
  }
  if (acc == 123.456f) *sink = acc;
}
// C: state update, one key per 8-lane group: var row (8 x 16 B), slot row (3 x 8 x 16 B), record (16 B, lane 0)
template <int WRITE>
__global__ void __launch_bounds__(256) k_state(float4* var, float4* slot, uint4* rec, unsigned long long nrows, unsigned long long nkeys, float* sink, unsigned long long salt) {
  const int lane = threadIdx.x & 7;
  const unsigned long long grp = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) >> 3;
  const unsigned long long ngrp = (unsigned long long)gridDim.x * 32;
  float acc = 0.f;
  for (unsigned long long i = grp; i < nkeys; i += ngrp) {
    const unsigned long long r = mix(i * 0x9E3779B97F4A7C15ULL + salt) % nrows;
    const unsigned long long r2 = mix(r + 12345) % nrows;
    float4 x = var[r * 8 + lane];
    float4 m = slot[r2 * 24 + lane], v = slot[r2 * 24 + 8 + lane], z = slot[r2 * 24 + 16 + lane];
    uint4 rc = make_uint4(0, 0, 0, 0);
    if (lane == 0) rc = rec[r2];
    acc += x.x + m.x + v.x + z.x + (float)rc.x;
    if (WRITE) {
      x.x += 1.f; m.x += 1.f; v.x += 1.f; z.x += 1.f;
      var[r * 8 + lane] = x;
      slot[r2 * 24 + lane] = m; slot[r2 * 24 + 8 + lane] = v; slot[r2 * 24 + 16 + lane] = z;
      if (lane == 0) { rc.z += 1; reinterpret_cast<unsigned*>(&rec[r2])[2] = rc.z; }
    }
  }
  if (acc == 123.456f) *sink = acc;
}
// D: one probe per lane
__global__ void __launch_bounds__(256) k_probe(const uint4* idx, unsigned long long nent, unsigned long long nkeys, float* sink, unsigned long long salt) {
  unsigned acc = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < nkeys; i += (unsigned long long)gridDim.x * 256) {
    const uint4 e = idx[mix(i * 0x9E3779B97F4A7C15ULL + salt) & (nent - 1)];
    acc += e.x + e.z;
  }
  if (acc == 0x12345u) *sink = (float)acc;
}

template <typename F>
static float timeit(F launch) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(1); hipDeviceSynchronize();
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(a); launch(7 + rep); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  return best * 1e3f;
}

int main() {
  float* sink; hipMalloc(&sink, 4);
  const unsigned long long big = 32ull << 30;
  float4* tab; if (hipMalloc(&tab, big) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(tab, 0, big);
  const unsigned long long nacc = 1ull << 20;
  printf("A: 1 M random 128-B row reads, 4 / 8 rows in flight per 8-lane group, grid 2048 / 1280\n");
  for (unsigned long long mb : {128ull, 512ull, 2048ull, 8192ull, 32768ull}) {
    const unsigned long long nrows = (mb << 20) / 128;
    const float t4 = timeit([&](int s) { k_rand<4><<<2048, 256>>>(tab, nrows, nacc, sink, s); });
    const float t8 = timeit([&](int s) { k_rand<8><<<1280, 256>>>(tab, nrows, nacc, sink, s); });
    printf("  footprint %6llu MB   %6.1f us (%5.2f TB/s)   %6.1f us (%5.2f TB/s)\n", mb, t4, nacc * 128 / t4 / 1e6, t8, nacc * 128 / t8 / 1e6);
  }
  printf("B: the same 1 M rows, tile-local permuted order (block = 2048 consecutive rows)\n");
  for (int grid : {512, 1024}) {
    const float t4 = timeit([&](int s) { k_tile_local<4><<<grid, 256>>>(tab, nacc, sink, 2 * s + 1); });
    const float t8 = timeit([&](int s) { k_tile_local<8><<<grid, 256>>>(tab, nacc, sink, 2 * s + 1); });
    printf("  grid %5d   4 in flight %6.1f us (%5.2f TB/s)   8 in flight %6.1f us (%5.2f TB/s)\n", grid, t4, nacc * 128 / t4 / 1e6, t8, nacc * 128 / t8 / 1e6);
  }
  printf("   ... and 340 k random rows of 128 MB + 43 MB (what a key-side pass would read instead)\n");
  {
    const float t = timeit([&](int s) { k_rand<4><<<2048, 256>>>(tab, (171ull << 20) / 128, 340000, sink, s); });
    printf("  340 k rows  %6.1f us\n", t);
  }
  printf("C: per-key state update (var 128 B of 6.4 GB, slot 384 B of 19.2 GB, record 16 B of 0.8 GB)\n");
  {
    const unsigned long long nrows = 50ull << 20;
    float4* var = tab; float4* slot = tab + nrows * 8;   // 6.4 GB, then 19.2 GB
    uint4* rec; hipMalloc(&rec, nrows * 16); hipMemset(rec, 0, nrows * 16);
    for (unsigned long long u : {109000ull, 340000ull, 1000000ull}) {
      for (int grid : {512, 1280, 2560}) {
        const float tr = timeit([&](int s) { k_state<0><<<grid, 256>>>(var, slot, rec, nrows, u, sink, s); });
        const float tw = timeit([&](int s) { k_state<1><<<grid, 256>>>(var, slot, rec, nrows, u, sink, s); });
        printf("  U %8llu grid %5d   read %6.1f us   read + write %6.1f us (%5.2f TB/s)\n", u, grid, tr, tw, u * (528.0 * 2) / tw / 1e6);
      }
    }
    hipFree(rec);
  }
  printf("D: random 16-B index probes into 2 GB\n");
  for (unsigned long long u : {109000ull, 340000ull, 1000000ull}) {
    const float t = timeit([&](int s) { k_probe<<<(unsigned)((u + 255) / 256), 256>>>(reinterpret_cast<const uint4*>(tab), 1ull << 27, u, sink, s); });
    printf("  %8llu probes  %6.1f us\n", u, t);
  }
  return 0;
}
