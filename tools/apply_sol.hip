// apply_sol.hip — an in-loop SPEED OF LIGHT for the GroupAdam apply of one batch (tools/apply_sol.py
// drives it, profiles/r06_apply_sol.txt holds the output).  A calibration, never the product:
//
//   the batch index is FREE (built by torch outside the timed kernel): every distinct key's positions are contiguous in
//   `kpos`, its row is known, keys are pre-classified — what the product's tile pass + tile sums + partition fronts exist
//   to find out.  What is left is the algorithmic work of SURVEY 8(d)'s apply figure and nothing else:
//     * every gradient row read ONCE (N x 4D bytes), straight into its key's sum;
//     * every key's state read and written once: var row (D), slot row m | v | z (3D), both 16-byte records.
//
//   cold key (<= HOTMIN positions): one 8-lane group per key, 8 keys per wave; its gradient rows RB at a time.
//   hot key: a tree of fan-in 64 over leaves of 256 positions (below): no atomics on data, one counter add per node.
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

// Hot keys are reduced by a TREE (memory-side float atomics into one row serialise at ~95 ns per 128-B add: the head key's
// 2.9 k chunk sums took 274 us that way — measured with this tool's first version).  A leaf = one wave summing up to LEAF
// positions of one key (64 rows per round trip, the next step's positions requested with this step's rows); the leaves of a
// key form groups of at most 64 siblings; a leaf writes its sum to its slot of `hpart` past the L2 (sc1), counts itself on the
// group's counter, and the wave whose count came LAST sums the group's slots (one round trip) and goes on as a node of the
// next level — until the root, which does the key's state update.
//   leaf[it]  = {start, cnt, group, slot}
//   group[g]  = {first slot, siblings, parent group or NONE, its slot in the parent | row of the key (root)}
// then the cold keys, 8 per wave item: cold[k] = {start, cnt, row, first position}.
constexpr unsigned NONE = 0xFFFFFFFFu;
constexpr int LEAF = 256;

template <int RB>
__global__ void __launch_bounds__(256) k_apply_sol(const float* __restrict__ grad, const unsigned* __restrict__ kpos,
                                                   const uint4* __restrict__ leaf, unsigned nhot, const uint4* __restrict__ group,
                                                   const uint4* __restrict__ cold, unsigned ncold, float* __restrict__ vrows,
                                                   float* __restrict__ srows, uint4* __restrict__ vmeta, uint4* __restrict__ smeta,
                                                   float* __restrict__ hpart, unsigned* __restrict__ gcnt, Hp h, unsigned day, int state) {
  const int wl = threadIdx.x & 63, lane = wl % LPR, g = wl / LPR;
  const unsigned nwaves = gridDim.x * (blockDim.x / 64);
  const unsigned nitems = nhot + (ncold + G - 1) / G;
  const float4* gr = reinterpret_cast<const float4*>(grad);
  const bool nograd = (state & 2) != 0;   // the state read-modify-write alone (no gradient row is read)
  state &= 1;
  auto record = [&](unsigned it) -> uint4 {
    if (it >= nitems) return make_uint4(0u, 0u, 0u, 0u);
    if (it < nhot) return leaf[it];
    const unsigned k = (it - nhot) * G + g;
    uint4 c = cold[k < ncold ? k : ncold - 1u];
    if (k >= ncold) c.y = 0u;
    return c;
  };
  unsigned it = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  uint4 cn = record(it);
  for (; it < nitems; it += nwaves) {
    const uint4 c = cn;
    cn = record(it + nwaves);
    if (it < nhot) {
      const unsigned start = c.x, cnt = c.y;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      unsigned kp = kpos[start + ((unsigned)wl < cnt ? (unsigned)wl : 0u)];
      for (unsigned j0 = 0; j0 < cnt; j0 += 64) {   // wave-uniform
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned p = __shfl(kp, g * 8 + j);
          v[j] = nograd ? make_float4(1e-3f, 1e-3f, 1e-3f, 1e-3f) : ld_stream(gr + (size_t)p * LPR + lane);
        }
        const unsigned nx = j0 + 64 + (unsigned)wl;
        kp = kpos[start + (nx < cnt ? nx : 0u)];
#pragma unroll
        for (int j = 0; j < 8; ++j) if (j0 + (unsigned)(g * 8 + j) < cnt) add4(acc, v[j]);
      }
      unsigned gi = c.z, slot = c.w;
      for (;;) {   // up the tree while this wave is the last of its group
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
          acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o); acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
        }
        const uint4 gd = group[gi];
        if (gd.y == 1u && gd.z == NONE) {   // the root: the key's sum is complete
          if (state) {
            const KeyState ks = load_state(vrows, srows, vmeta, smeta, gd.w, g == 0, lane);
            update_key(vrows, srows, vmeta, smeta, gd.w, g == 0, lane, acc, h, day, ks);
          }
          break;
        }
        float* a = hpart + (size_t)slot * D + lane * 4;
        if (g == 0) {
          __hip_atomic_store(a, acc.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(a + 1, acc.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(a + 2, acc.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(a + 3, acc.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the slot is written (past the L2) before the leaf counts itself
        unsigned old = 0;
        if (wl == 0) old = __hip_atomic_fetch_add(&gcnt[gi], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
        if (old != gd.y - 1u) break;
        if (wl == 0) __hip_atomic_store(&gcnt[gi], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the group's slots: 8 lane groups x 8 rows, read past the L2
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned sidx = (unsigned)(g * 8 + j);
          const float* q = hpart + (size_t)(gd.x + (sidx < gd.y ? sidx : 0u)) * D + lane * 4;
          float4 v;
          v.x = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v.y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          v.z = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v.w = __hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (sidx < gd.y) add4(acc, v);
        }
        if (gd.z == NONE) {   // this group was the key's top level: its sum is the key's (reduced below), gd.w = the row
#pragma unroll
          for (int o = LPR; o < 64; o <<= 1) {
            acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o); acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
          }
          if (state) {
            const KeyState ks = load_state(vrows, srows, vmeta, smeta, gd.w, g == 0, lane);
            update_key(vrows, srows, vmeta, smeta, gd.w, g == 0, lane, acc, h, day, ks);
          }
          break;
        }
        gi = gd.z; slot = gd.w;
      }
    } else {
      const unsigned start = c.x, cnt = c.y;
      const bool have = cnt != 0u;
      KeyState ks{};
      if (state) ks = load_state(vrows, srows, vmeta, smeta, c.z, have, lane);   // with the first gradient row: one round trip
      float4 acc = nograd ? make_float4(1e-3f, 1e-3f, 1e-3f, 1e-3f) : ld_stream(gr + (size_t)c.w * LPR + lane);
      if (!have) acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (unsigned j0 = 1; !nograd && __ballot(j0 < cnt) != 0ull; j0 += RB) {
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
      else if (have && acc.x == 1.2345e-30f) hpart[0] = acc.x;   // (keeps the sums alive)
    }
  }
}
}  // namespace

// state = 0: the gradient stream alone (sums dropped); 1: the whole apply; 3: the state read-modify-write alone.  rb: gradient rows in flight per lane group (2 / 4 / 8)
extern "C" int sol_apply(const float* grad, const unsigned* kpos, const void* leaf, unsigned nhot, const void* group, const void* cold,
                         unsigned ncold, float* vrows, float* srows, void* vmeta, void* smeta, float* hpart, unsigned* gcnt, float lr,
                         float b1p, float b2p, unsigned day, int grid, int rb, int state, void* stream) {
  Hp h;
  h.lr = lr; h.b1 = 0.9f; h.b2 = 0.999f; h.eps = 1e-8f;
  h.alpha = lr * sqrtf(1.f - b2p) / (1.f - b1p);
  hipStream_t s = (hipStream_t)stream;
#define GO(RB)                                                                                                                  \
  k_apply_sol<RB><<<grid, 256, 0, s>>>(grad, kpos, (const uint4*)leaf, nhot, (const uint4*)group, (const uint4*)cold, ncold,    \
                                       vrows, srows, (uint4*)vmeta, (uint4*)smeta, hpart, gcnt, h, day, state)
  if (rb <= 2) GO(2); else if (rb <= 4) GO(4); else GO(8);
#undef GO
  return (int)hipGetLastError();
}
