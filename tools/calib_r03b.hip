// Round-3 calibration, part 2: what makes the per-key state update (k_apply's finish) expensive.
// U keys, one 8-lane group per key; arrays: var rows 128 B (6.4 GB), slot rows 384 B (19.2 GB), records 16 B (0.8 GB),
// or ONE co-located record of 640 B per key (32 GB).  hipcc -O3 --offload-arch=gfx950 -o build/tools/calib_r03b ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned long long mix(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}
__device__ __forceinline__ void st(float4* p, float4 v, int nt) {
  if (nt) {
    __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
    __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
  } else {
    *p = v;
  }
}
// what: bit 0 var read, 1 slot read, 2 rec read, 3 var write, 4 slot write, 5 rec write (4 B), 6 rec write 16 B
template <int NT>
__global__ void __launch_bounds__(256) k_state(float4* var, float4* slot, uint4* rec, unsigned long long nrows, unsigned long long nkeys,
                                               float* sink, unsigned long long salt, int what) {
  const int lane = threadIdx.x & 7;
  const unsigned long long grp = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) >> 3;
  const unsigned long long ngrp = (unsigned long long)gridDim.x * 32;
  float acc = 0.f;
  for (unsigned long long i = grp; i < nkeys; i += ngrp) {
    const unsigned long long r = mix(i * 0x9E3779B97F4A7C15ULL + salt) % nrows;
    const unsigned long long r2 = mix(r + 12345) % nrows;
    float4 x = make_float4(1, 2, 3, 4), m = x, v = x, z = x;
    uint4 rc = make_uint4(0, 0, 0, 0);
    if (what & 1) x = var[r * 8 + lane];
    if (what & 2) { m = slot[r2 * 24 + lane]; v = slot[r2 * 24 + 8 + lane]; z = slot[r2 * 24 + 16 + lane]; }
    if ((what & 4) && lane == 0) rc = rec[r2];
    acc += x.x + m.x + v.x + z.x + (float)rc.x;
    x.x += 1.f; m.x += 1.f; v.x += 1.f; z.x += 1.f;
    if (what & 8) st(&var[r * 8 + lane], x, NT);
    if (what & 16) { st(&slot[r2 * 24 + lane], m, NT); st(&slot[r2 * 24 + 8 + lane], v, NT); st(&slot[r2 * 24 + 16 + lane], z, NT); }
    if ((what & 32) && lane == 0) { rc.z += 1; reinterpret_cast<unsigned*>(&rec[r2])[2] = rc.z; }
    if ((what & 64) && lane == 0) { rc.z += 1; rec[r2] = rc; }
  }
  if (acc == 123.456f) *sink = acc;
}
// co-located: 640 B per key = 40 float4; lane l takes float4 l, l + 8, ... (5 per lane)
template <int NT, int WRITE>
__global__ void __launch_bounds__(256) k_coloc(float4* tab, unsigned long long nrows, unsigned long long nkeys, float* sink, unsigned long long salt) {
  const int lane = threadIdx.x & 7;
  const unsigned long long grp = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) >> 3;
  const unsigned long long ngrp = (unsigned long long)gridDim.x * 32;
  float acc = 0.f;
  for (unsigned long long i = grp; i < nkeys; i += ngrp) {
    const unsigned long long r = mix(i * 0x9E3779B97F4A7C15ULL + salt) % nrows;
    float4 a[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) a[j] = tab[r * 40 + j * 8 + lane];
#pragma unroll
    for (int j = 0; j < 5; ++j) { acc += a[j].x; a[j].x += 1.f; }
    if (WRITE) {
#pragma unroll
      for (int j = 0; j < 5; ++j) st(&tab[r * 40 + j * 8 + lane], a[j], NT);
    }
  }
  if (acc == 123.456f) *sink = acc;
}
__global__ void k_empty() {}

template <typename F>
static float timeit(F launch, bool tail = false) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(1); hipDeviceSynchronize();
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(a); launch(7 + rep); if (tail) k_empty<<<1, 64>>>(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  return best * 1e3f;
}

int main() {
  float* sink; (void)hipMalloc(&sink, 4);
  const unsigned long long big = 32ull << 30;
  float4* tab; if (hipMalloc(&tab, big) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(tab, 0, big);
  const unsigned long long nrows = 50ull << 20;
  float4* var = tab; float4* slot = tab + nrows * 8;
  uint4* rec; (void)hipMalloc(&rec, nrows * 16); (void)hipMemset(rec, 0, nrows * 16);
  const unsigned long long U = 109000;
  struct { const char* name; int what; } cases[] = {
    {"read var", 1}, {"read slot", 2}, {"read rec", 4}, {"read all", 7},
    {"write var (no read)", 8}, {"write slot (no read)", 16}, {"write rec 4 B (no read)", 32}, {"write rec 16 B (no read)", 64},
    {"write var+slot (no read)", 24}, {"write all (no read)", 56},
    {"rmw var", 9}, {"rmw slot", 18}, {"rmw rec 4 B", 36}, {"rmw var+slot", 27}, {"rmw var+slot, read rec", 31},
    {"rmw all (rec 4 B)", 63}, {"rmw all (rec 16 B)", 95},
  };
  printf("U = %llu keys, grid 1280 x 256; plain stores | nontemporal stores | plain + an empty kernel behind\n", U);
  for (auto& c : cases) {
    const float t0 = timeit([&](int s) { k_state<0><<<1280, 256>>>(var, slot, rec, nrows, U, sink, s, c.what); });
    const float t1 = timeit([&](int s) { k_state<1><<<1280, 256>>>(var, slot, rec, nrows, U, sink, s, c.what); });
    const float t2 = timeit([&](int s) { k_state<0><<<1280, 256>>>(var, slot, rec, nrows, U, sink, s, c.what); }, true);
    printf("  %-28s %6.1f us | %6.1f us | %6.1f us\n", c.name, t0, t1, t2);
  }
  printf("co-located 640-B records (32 GB): read | rmw plain | rmw nontemporal\n");
  {
    const unsigned long long nr = big / 640;
    const float t0 = timeit([&](int s) { k_coloc<0, 0><<<1280, 256>>>(tab, nr, U, sink, s); });
    const float t1 = timeit([&](int s) { k_coloc<0, 1><<<1280, 256>>>(tab, nr, U, sink, s); });
    const float t2 = timeit([&](int s) { k_coloc<1, 1><<<1280, 256>>>(tab, nr, U, sink, s); });
    printf("  %6.1f us | %6.1f us | %6.1f us\n", t0, t1, t2);
  }
  printf("same kernels, 3 back to back (per launch): rmw all plain | nontemporal\n");
  {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int nt = 0; nt < 2; ++nt) {
      hipEventRecord(a);
      for (int r = 0; r < 10; ++r) {
        if (nt) k_state<1><<<1280, 256>>>(var, slot, rec, nrows, U, sink, 100 + r, 63);
        else k_state<0><<<1280, 256>>>(var, slot, rec, nrows, U, sink, 100 + r, 63);
      }
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b);
      printf("  %s %6.1f us per launch\n", nt ? "nontemporal" : "plain", ms * 100.f);
    }
  }
  return 0;
}
