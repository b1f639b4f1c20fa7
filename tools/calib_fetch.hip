// calib_fetch.hip — what rocprofv3's FETCH_SIZE / WRITE_SIZE report for NARROW scattered requests on gfx950 (VERDICT r5
// item 2a).  MI355X_MICROARCH.md calibrates the counters for wide coalesced streams only ("FETCH_SIZE reports exactly half
// of the bytes of a 16 B/lane streaming read ... other access widths are uncalibrated"); the apply kernels' excess traffic
// ("1.40 x") is mostly 16-byte records and index entries, so the correction for THEM has to be measured.
//
// Every kernel touches a KNOWN number of distinct, random, 128-byte-aligned lines of a 4 GiB buffer (far beyond the 256 MB
// Infinity Cache; each line at most once per launch, so nothing is served on-die), with a fixed request shape:
//   k_rd16_scatter   one 16-B load per lane, every lane another line           (index entry / RowMeta pattern)
//   k_rd16x2_scatter two 16-B loads per lane, the two in ONE 64-B half of a line: does a second record of the half cost more
//   k_rd64_scatter   four lanes x 16 B = one 64-B half line per 4-lane group
//   k_rd128_rows     eight lanes x 16 B = one whole 128-B line per group         (embedding row pattern)
//   k_rd_stream      16 B per lane, fully coalesced                              (the guide's calibrated case: x 2)
//   k_wr4_scatter / k_wr16_scatter / k_wr128_rows / k_wr_stream    the same shapes as stores
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); tools/calib_fetch_summary.py divides the
// counters by the known requests.  Build: hipcc -O3 --offload-arch=gfx950 -o build/calib_fetch tools/calib_fetch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// a permutation of [0, nlines): line = (i * A + B) mod nlines with A odd and nlines a power of two — every line at most once
__device__ __forceinline__ unsigned long long line_of(unsigned long long i, unsigned long long nlines, unsigned long long salt) {
  return (i * 0x9E3779B97F4A7C15ULL + salt) & (nlines - 1);
}

__global__ void __launch_bounds__(256) k_rd16_scatter(const uint4* buf, unsigned long long nlines, unsigned long long nreq, unsigned long long salt, unsigned* sink) {
  unsigned acc = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < nreq; i += (unsigned long long)gridDim.x * 256)
    acc += buf[line_of(i, nlines, salt) * 8 + (i & 7)].x;   // (any of the line's eight 16-B slots)
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void __launch_bounds__(256) k_rd16x2_scatter(const uint4* buf, unsigned long long nlines, unsigned long long nreq, unsigned long long salt, unsigned* sink) {
  unsigned acc = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < nreq; i += (unsigned long long)gridDim.x * 256) {
    const uint4* l = buf + line_of(i, nlines, salt) * 8 + (i & 4);
    acc += l[0].x + l[2].x;
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void __launch_bounds__(256) k_rd64_scatter(const uint4* buf, unsigned long long nlines, unsigned long long nreq, unsigned long long salt, unsigned* sink) {
  unsigned acc = 0;
  const unsigned long long g0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) >> 2, ng = ((unsigned long long)gridDim.x * 256) >> 2;
  for (unsigned long long i = g0; i < nreq; i += ng) acc += buf[line_of(i, nlines, salt) * 8 + (i & 1) * 4 + (threadIdx.x & 3)].x;
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void __launch_bounds__(256) k_rd128_rows(const uint4* buf, unsigned long long nlines, unsigned long long nreq, unsigned long long salt, unsigned* sink) {
  unsigned acc = 0;
  const unsigned long long g0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) >> 3, ng = ((unsigned long long)gridDim.x * 256) >> 3;
  for (unsigned long long i = g0; i < nreq; i += ng) acc += buf[line_of(i, nlines, salt) * 8 + (threadIdx.x & 7)].x;
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void __launch_bounds__(256) k_rd_stream(const uint4* buf, unsigned long long n16, unsigned* sink) {
  unsigned acc = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * 256) acc += buf[i].x;
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void __launch_bounds__(256) k_wr4_scatter(uint4* buf, unsigned long long nlines, unsigned long long nreq, unsigned long long salt) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < nreq; i += (unsigned long long)gridDim.x * 256)
    reinterpret_cast<unsigned*>(buf + line_of(i, nlines, salt) * 8 + (i & 7))[2] = (unsigned)i;
}
__global__ void __launch_bounds__(256) k_wr16_scatter(uint4* buf, unsigned long long nlines, unsigned long long nreq, unsigned long long salt) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < nreq; i += (unsigned long long)gridDim.x * 256)
    buf[line_of(i, nlines, salt) * 8 + (i & 7)] = make_uint4((unsigned)i, 1u, 2u, 3u);
}
__global__ void __launch_bounds__(256) k_wr128_rows(uint4* buf, unsigned long long nlines, unsigned long long nreq, unsigned long long salt) {
  const unsigned long long g0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) >> 3, ng = ((unsigned long long)gridDim.x * 256) >> 3;
  for (unsigned long long i = g0; i < nreq; i += ng) buf[line_of(i, nlines, salt) * 8 + (threadIdx.x & 7)] = make_uint4((unsigned)i, 1u, 2u, 3u);
}
__global__ void __launch_bounds__(256) k_wr_stream(uint4* buf, unsigned long long n16) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * 256)
    buf[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
}

int main(int argc, char** argv) {
  const unsigned long long bytes = 4ull << 30, nlines = bytes / 128, nreq = 1ull << 20, n16 = (256ull << 20) / 16;
  uint4* buf = nullptr;
  unsigned* sink = nullptr;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 1, bytes));
  CK(hipDeviceSynchronize());
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int grid = 2048, reps = 3;
  std::printf("buffer %llu MiB, %llu requests per scattered launch (each another 128-B line), stream launches move %llu MiB\n",
              bytes >> 20, nreq, (n16 * 16) >> 20);
#define RUN(name, call, reqbytes, lines)                                                         \
  for (int r = 0; r < reps; ++r) {                                                                 \
    const unsigned long long salt = 0x51ED270B9ULL * (unsigned long long)(r + 1) * 7919ULL; (void)salt;  \
    CK(hipEventRecord(a)); call; CK(hipEventRecord(b)); CK(hipEventSynchronize(b));               \
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));                                              \
    if (r == reps - 1) std::printf("%-18s requested %12llu B in %9llu lines  %8.1f us\n", name, (unsigned long long)(reqbytes), (unsigned long long)(lines), ms * 1e3); \
  }
  RUN("k_rd16_scatter", (k_rd16_scatter<<<grid, 256>>>(buf, nlines, nreq, salt, sink)), nreq * 16, nreq)
  RUN("k_rd16x2_scatter", (k_rd16x2_scatter<<<grid, 256>>>(buf, nlines, nreq, salt, sink)), nreq * 32, nreq)
  RUN("k_rd64_scatter", (k_rd64_scatter<<<grid, 256>>>(buf, nlines, nreq, salt, sink)), nreq * 64, nreq)
  RUN("k_rd128_rows", (k_rd128_rows<<<grid, 256>>>(buf, nlines, nreq, salt, sink)), nreq * 128, nreq)
  RUN("k_rd_stream", (k_rd_stream<<<grid, 256>>>(buf, n16, sink)), n16 * 16, n16 / 8)
  RUN("k_wr4_scatter", (k_wr4_scatter<<<grid, 256>>>(buf, nlines, nreq, salt)), nreq * 4, nreq)
  RUN("k_wr16_scatter", (k_wr16_scatter<<<grid, 256>>>(buf, nlines, nreq, salt)), nreq * 16, nreq)
  RUN("k_wr128_rows", (k_wr128_rows<<<grid, 256>>>(buf, nlines, nreq, salt)), nreq * 128, nreq)
  RUN("k_wr_stream", (k_wr_stream<<<grid, 256>>>(buf, n16)), n16 * 16, n16 / 8)
  CK(hipDeviceSynchronize());
  return 0;
}
