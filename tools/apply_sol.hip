// apply_sol.hip — an in-loop SPEED OF LIGHT for the GroupAdam apply of one batch (tools/apply_sol.py
// drives it, profiles/r06_apply_sol.txt holds the output).  A calibration, never the product:
//
//   the batch index is FREE (built by torch outside the timed kernel): every distinct key's positions are contiguous in
//   `kpos`, its row is known, keys are pre-classified — what the product's tile pass + tile sums + partition fronts exist
//   to find out.  What is left is the algorithmic work of SURVEY 8(d)'s apply figure and nothing else:
//     * every gradient row read ONCE (N x 4D bytes), straight into its key's sum;
//     * every key's state read and written once: var row (D), slot row m | v | z (3D), both 16-byte records.
//
//   cold key (<= HOTMIN positions): one 8-lane group per key, 8 keys per wave; its gradient rows 4 at a time.
//   hot key: chunks of CH positions, one wave per chunk (8 groups x CH / 8 rows, xor-shuffle reduce), the chunk's sum added
//   to the key's accumulator with float atomics (memory-side on gfx950), a returning counter add tells the LAST chunk,
//   whose wave reads the accumulator past its L2 (agent-scope atomic loads), clears it and does the key's update.
//
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o build/apply_sol.so tools/apply_sol.hip
#include <hip/hip_runtime.h>
#include <cstdint>

namespace {
constexpr int D = 32, LPR = 8, G = 64 / LPR;   // dim 32: 8 lanes x float4 per row

struct Hp { float lr, b1, b2, eps, alpha; };

__device__ __forceinline__ float4 ld_stream(const float4* p) {
  float4 v;
  v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y);
  v.z = __builtin_nontemporal_load(&p->z); v.w = __builtin_nontemporal_load(&p->w);
  return v;
}
__device__ __forceinline__ void add4(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// the state of one key, requested by its lane group (lane = 0..7) BEFORE its gradient rows are summed: one round trip
struct KeyState { float4 x, m, v, z; uint4 vm, sm; };
__device__ __forceinline__ KeyState load_state(const float* __restrict__ vrows, const float* __restrict__ srows,
                                               const uint4* __restrict__ vmeta, const uint4* __restrict__ smeta, unsigned row, bool act,
                                               int lane) {
  const size_t r = act ? row : 0u;
  const float4* xr = reinterpret_cast<const float4*>(vrows + r * D) + lane;
  const float4* sr = reinterpret_cast<const float4*>(srows + r * 3 * D) + lane;
  KeyState k;
  k.x = *xr; k.m = sr[0]; k.v = sr[D / 4]; k.z = sr[2 * (D / 4)];
  k.vm = vmeta[r]; k.sm = smeta[r];
  return k;
}
// GroupAdam V4 with l1 = l2 = l21 = 0, t >= 2 branch (training_ops.cc:7166-7195), and the stores
__device__ __forceinline__ void update_key(float* __restrict__ vrows, float* __restrict__ srows, uint4* __restrict__ vmeta,
                                           uint4* __restrict__ smeta, unsigned row, bool act, int lane, float4 g, const Hp& h,
                                           unsigned day, const KeyState& k) {
  const size_t r = act ? row : 0u;
  float4* xr = reinterpret_cast<float4*>(vrows + r * D) + lane;
  float4* sr = reinterpret_cast<float4*>(srows + r * 3 * D) + lane;
  const float4 x = k.x, m = k.m, v = k.v, z = k.z;
  const uint4 vm = k.vm, sm = k.sm;
  float4 mn, vn, zn, xn;
  const float ob1 = 1.f - h.b1, ob2 = 1.f - h.b2;
#define EL(c)                                                                  \
  {                                                                            \
    mn.c = h.b1 * m.c + ob1 * g.c;                                             \
    vn.c = h.b2 * v.c + ob2 * (g.c * g.c);                                     \
    const float s = sqrtf(vn.c);                                               \
    zn.c = z.c + (h.alpha * mn.c - (s - sqrtf(v.c)) * x.c);                    \
    xn.c = (0.f - zn.c) / (s + h.eps);                                         \
  }
  EL(x) EL(y) EL(z) EL(w)
#undef EL
  if (act) {
    *xr = xn; sr[0] = mn; sr[D / 4] = vn; sr[2 * (D / 4)] = zn;
    if (lane == 0) {   // frequency words: the lookup's bookkeeping on the var record, AddFrequency on the slot record
      unsigned lo = (vm.z & 0xFFFFu) + 1u; if (lo > 65535u) lo = 65535u;
      reinterpret_cast<unsigned*>(vmeta + r)[2] = (day << 16) | lo;
      unsigned ls = (sm.z & 0xFFFFu) + 1u; if (ls > 65535u) ls = 65535u;
      reinterpret_cast<unsigned*>(smeta + r)[2] = (day << 16) | ls;
    }
  }
}

// items: [0, nhot) hot chunks {start, cnt, row, hot index | chunks of the key << 16}; then the cold keys, 8 per wave item:
// cold[k] = {start, cnt, row, -}
template <int RB>
__global__ void __launch_bounds__(256) k_apply_sol(const float* __restrict__ grad, const unsigned* __restrict__ kpos,
                                                   const uint4* __restrict__ hot, unsigned nhot, const uint4* __restrict__ cold,
                                                   unsigned ncold, float* __restrict__ vrows, float* __restrict__ srows,
                                                   uint4* __restrict__ vmeta, uint4* __restrict__ smeta, float* __restrict__ hotacc,
                                                   unsigned* __restrict__ hotcnt, Hp h, unsigned day, int state) {
  const int wl = threadIdx.x & 63, lane = wl % LPR, g = wl / LPR;
  const unsigned nwaves = gridDim.x * (blockDim.x / 64);
  const unsigned nitems = nhot + (ncold + G - 1) / G;
  const float4* gr = reinterpret_cast<const float4*>(grad);
  for (unsigned it = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6); it < nitems; it += nwaves) {
    if (it < nhot) {
      const uint4 c = hot[it];
      const unsigned start = c.x, cnt = c.y;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (unsigned j0 = 0; j0 < cnt; j0 += G * RB) {
        unsigned p[RB];
        float4 v[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) { const unsigned jj = j0 + j * G + g; p[j] = kpos[start + (jj < cnt ? jj : 0u)]; }
#pragma unroll
        for (int j = 0; j < RB; ++j) v[j] = ld_stream(gr + (size_t)p[j] * LPR + lane);
#pragma unroll
        for (int j = 0; j < RB; ++j) if (j0 + j * G + g < cnt) add4(acc, v[j]);
      }
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) {
        acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o); acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
      }
      const unsigned hi = c.w & 0xFFFFu, nch = c.w >> 16;
      float* a = hotacc + (size_t)hi * D + lane * 4;
      bool last = nch == 1u;
      if (nch > 1u) {
        if (g == 0) { atomicAdd(a, acc.x); atomicAdd(a + 1, acc.y); atomicAdd(a + 2, acc.z); atomicAdd(a + 3, acc.w); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the adds are performed (memory side) before the chunk is counted
        unsigned old = 0;
        if (wl == 0) old = __hip_atomic_fetch_add(&hotcnt[hi], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
        last = old == nch - 1u;
        if (last) {
          if (g == 0) {
            acc.x = __hip_atomic_exchange(a, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc.y = __hip_atomic_exchange(a + 1, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc.z = __hip_atomic_exchange(a + 2, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc.w = __hip_atomic_exchange(a + 3, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (wl == 0) __hip_atomic_store(&hotcnt[hi], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (last && state) {
        const KeyState ks = load_state(vrows, srows, vmeta, smeta, c.z, g == 0, lane);
        update_key(vrows, srows, vmeta, smeta, c.z, g == 0, lane, acc, h, day, ks);
      }
    } else {
      const unsigned k = (it - nhot) * G + g;
      const bool have = k < ncold;
      const uint4 c = cold[have ? k : ncold - 1u];
      const unsigned start = c.x, cnt = have ? c.y : 0u;
      KeyState ks{};
      if (state) ks = load_state(vrows, srows, vmeta, smeta, c.z, have, lane);   // with the first gradient rows: one round trip
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (unsigned j0 = 0; __ballot(j0 < cnt) != 0ull; j0 += RB) {
        unsigned p[RB];
        float4 v[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) p[j] = kpos[start + (j0 + j < cnt ? j0 + j : 0u)];
#pragma unroll
        for (int j = 0; j < RB; ++j) v[j] = ld_stream(gr + (size_t)p[j] * LPR + lane);
#pragma unroll
        for (int j = 0; j < RB; ++j) if (j0 + j < cnt) add4(acc, v[j]);
      }
      if (state) update_key(vrows, srows, vmeta, smeta, c.z, have, lane, acc, h, day, ks);
      else if (have && acc.x == 1.2345e-30f) hotacc[0] = acc.x;   // (keeps the sums alive)
    }
  }
}
}  // namespace

// state = 0: the gradient stream alone (sums dropped); 1: the whole apply.  rb: gradient rows in flight per lane group (2 / 4 / 8)
extern "C" int sol_apply(const float* grad, const unsigned* kpos, const void* hot, unsigned nhot, const void* cold, unsigned ncold,
                         float* vrows, float* srows, void* vmeta, void* smeta, float* hotacc, unsigned* hotcnt, float lr, float b1p,
                         float b2p, unsigned day, int grid, int rb, int state, void* stream) {
  Hp h;
  h.lr = lr; h.b1 = 0.9f; h.b2 = 0.999f; h.eps = 1e-8f;
  h.alpha = lr * sqrtf(1.f - b2p) / (1.f - b1p);
  hipStream_t s = (hipStream_t)stream;
#define GO(RB)                                                                                                                  \
  k_apply_sol<RB><<<grid, 256, 0, s>>>(grad, kpos, (const uint4*)hot, nhot, (const uint4*)cold, ncold, vrows, srows,            \
                                       (uint4*)vmeta, (uint4*)smeta, hotacc, hotcnt, h, day, state)
  if (rb <= 2) GO(2); else if (rb <= 4) GO(4); else GO(8);
#undef GO
  return (int)hipGetLastError();
}
