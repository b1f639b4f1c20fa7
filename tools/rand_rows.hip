// Calibration: what the box sustains for RANDOM row accesses of a table far larger than the caches
// (hipcc -O3 --offload-arch=gfx950 -o /tmp/rand_rows tools/rand_rows.hip).  Rows of ROWB bytes, 8 lanes x 16 B per
// 128 B; modes: read (sum), read-modify-write, rmw + 1-byte side write (the RowMeta flag pattern).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned long long mix(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}
// one lane group of LPR lanes per row; INF rows in flight per group
template <int LPR, int INF, int MODE>
__global__ void __launch_bounds__(256) k(float4* tab, unsigned char* side, unsigned long long nrows, unsigned long long nacc, float* sink, unsigned long long salt) {
  const int lane = threadIdx.x % LPR;
  const unsigned long long grp = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) / LPR;
  const unsigned long long ngrp = (unsigned long long)gridDim.x * 256 / LPR;
  float acc = 0.f;
  for (unsigned long long i = grp * INF; i < nacc; i += ngrp * INF) {
    float4 v[INF];
    unsigned long long r[INF];
#pragma unroll
    for (int j = 0; j < INF; ++j) { r[j] = mix((i + j) * 0x9E3779B97F4A7C15ULL + salt) % nrows; v[j] = tab[r[j] * LPR + lane]; }
#pragma unroll
    for (int j = 0; j < INF; ++j) {
      acc += v[j].x;
      if (MODE >= 1) { v[j].x += 1.f; tab[r[j] * LPR + lane] = v[j]; }
      if (MODE == 2 && lane == 0) side[r[j] * 128] = (unsigned char)j;
      if (MODE == 3 && lane == 0) *reinterpret_cast<unsigned*>(side + r[j] * 128) = (unsigned)j;
      if (MODE == 4 && lane == 0) *reinterpret_cast<float4*>(side + r[j] * 128) = v[j];
      if (MODE == 5 && lane < 2) *reinterpret_cast<float4*>(side + r[j] * 128 + lane * 16) = v[j];
      if (MODE == 6 && lane < 4) *reinterpret_cast<float4*>(side + r[j] * 128 + lane * 16) = v[j];
      if (MODE == 7) *reinterpret_cast<float4*>(side + r[j] * 128 + lane * 16) = v[j];
      if (MODE == 8 && lane == 0) acc += *reinterpret_cast<float*>(side + r[j] * 128);   // side READ (16 B line fetch)
    }
  }
  if (acc == 123.456f) *sink = acc;
}

template <int LPR, int INF, int MODE>
void run(const char* name, float4* tab, unsigned char* side, unsigned long long nrows, unsigned long long nacc, float* sink, int grid) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<LPR, INF, MODE><<<grid, 256>>>(tab, side, nrows, nacc, sink, 1);
  hipDeviceSynchronize();
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    k<LPR, INF, MODE><<<grid, 256>>>(tab, side, nrows, nacc, sink, 7 + rep);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  const double bytes = (double)nacc * LPR * 16 * (MODE >= 1 ? 2 : 1);
  printf("%-44s grid %5d  %8.1f us  %7.2f TB/s  %7.1f M rows/s\n", name, grid, best * 1e3, bytes / best / 1e9, nacc / best / 1e3);
}

int main() {
  const unsigned long long nrows128 = 100ull << 20;   // 100 M rows x 128 B = 12.8 GB
  float4* tab; unsigned char* side; float* sink;
  hipMalloc(&tab, nrows128 * 128); hipMalloc(&side, nrows128 * 128); hipMalloc(&sink, 4);
  hipMemset(tab, 0, nrows128 * 128); hipMemset(side, 0, nrows128 * 128);
  const unsigned long long nacc = 1ull << 20;   // 1 M row accesses per launch
  for (int grid : {2048}) {
    run<8, 4, 0>("128 B rows, read, 4 in flight/group", tab, side, nrows128, nacc, sink, grid);
    run<8, 4, 1>("128 B rows, read+write back (rmw)", tab, side, nrows128, nacc, sink, grid);
    run<8, 4, 2>("rmw + side write 1 B", tab, side, nrows128, nacc, sink, grid);
    run<8, 4, 3>("rmw + side write 4 B", tab, side, nrows128, nacc, sink, grid);
    run<8, 4, 4>("rmw + side write 16 B", tab, side, nrows128, nacc, sink, grid);
    run<8, 4, 5>("rmw + side write 32 B", tab, side, nrows128, nacc, sink, grid);
    run<8, 4, 6>("rmw + side write 64 B", tab, side, nrows128, nacc, sink, grid);
    run<8, 4, 7>("rmw + side write 128 B (full line)", tab, side, nrows128, nacc, sink, grid);
    run<8, 4, 8>("rmw + side read 4 B", tab, side, nrows128, nacc, sink, grid);
  }
  return 0;
}
