// hipemu -- a minimal single-process CPU emulator of the HIP device model, TEST INFRASTRUCTURE ONLY.
//
// The product kernels under ddsp_svc_amd/csrc/*.hip are pure HIP for gfx950.  There is no GPU in
// the build container, so tests/hipemu compiles those *same source files* as host C++ (clang++
// -x c++ -I tests/hipemu) against this header: every workgroup runs as a set of cooperatively
// scheduled fibers (one per work-item), __syncthreads() and the wave64 cross-lane operations
// (__shfl*, f32 MFMA) are rendez-vous points between fibers.  That lets the CPU test-suite check
// kernel *logic* (indexing, LDS staging, wave scans, MFMA fragment layouts, tails) under
// AddressSanitizer before a GPU minute is spent.  It is never loaded by the product package and
// is not a fallback: ddsp_svc_amd/_ffi.py only ever loads the hipcc-built libddsp_hip.so.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>

// ---- qualifiers -------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __shared__ static
#define __constant__ static
#define __launch_bounds__(...)
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hipemu::dyn_shared());

// ---- vector types -----------------------------------------------------------------------------
struct uint3 { unsigned x, y, z; };
struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) double2 { double x, y; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return {x, y}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }

// ---- runtime API subset -----------------------------------------------------------------------
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorLaunchFailure = 719 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
// streams execute synchronously here, so events have nothing to order
typedef void* hipEvent_t;
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0 };
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 2; return hipSuccess; }   // a chip of two CUs:
template <class K>                                                                                                 // small grids, several
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K, int, size_t) { *n = 2; return hipSuccess; }   // chunks per row
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { static int token; *e = &token; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
enum { hipStreamNonBlocking = 1 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { static int token; *s = &token; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemset2DAsync(void* p, size_t pitch, int v, size_t width, size_t height, hipStream_t) {
  for (size_t r = 0; r < height; ++r) memset(static_cast<char*>(p) + r * pitch, v, width);
  return hipSuccess;
}
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }

namespace hipemu {
struct Barrier { int count = 0; unsigned gen = 0; };
struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  uint3 tid{0, 0, 0};
  int lin = 0, wave = 0, lane = 0;
  bool done = false;
  Barrier* wait = nullptr;
  unsigned wait_gen = 0;
};
struct Wave {
  Barrier bar;
  int alive = 0;
  unsigned parity = 0;
  alignas(16) unsigned char slot[2][64][64];   // per-lane deposit area, double buffered
};
struct State {
  Fiber* cur = nullptr;
  uint3 bid{0, 0, 0};
  dim3 bdim, gdim;
  Barrier blockbar;
  int alive = 0;
  Wave* waves = nullptr;
  void* dynsh = nullptr;
};
extern State g;
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void block_sync();
void wave_sync();            // all live lanes of the calling fiber's wave
inline void* dyn_shared() { return g.dynsh; }
inline Wave& wave() { return g.waves[g.cur->wave]; }
// deposit a value, rendez-vous, return the wave's slot array for this collective
template <class T> inline const unsigned char (*exchange(const T& v))[64] {
  static_assert(sizeof(T) <= 64, "slot too small");
  Wave& w = wave();
  unsigned p = w.parity & 1u;      // same for all lanes of the wave at this collective
  memcpy(w.slot[p][g.cur->lane], &v, sizeof(T));
  wave_sync();                     // the releasing lane flips w.parity
  return w.slot[p];
}
template <class T> inline T peek(const unsigned char (*s)[64], int lane) { T r; memcpy(&r, s[lane], sizeof(T)); return r; }
}  // namespace hipemu

#define threadIdx (hipemu::g.cur->tid)
#define blockIdx (hipemu::g.bid)
#define blockDim (hipemu::g.bdim)
#define gridDim (hipemu::g.gdim)
static constexpr int warpSize = 64;

template <class K, class... Args>
static inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args) {
  hipemu::launch(grid, block, shmem, [=]() { kernel(args...); });
}

// ---- synchronisation and cross-lane -----------------------------------------------------------
static inline void __syncthreads() { hipemu::block_sync(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

template <class T> static inline T __shfl(T v, int src, int width = 64) {
  auto s = hipemu::exchange(v);
  int lane = hipemu::g.cur->lane;
  int base = lane - (lane % width);
  return hipemu::peek<T>(s, base + (((src % width) + width) % width));
}
template <class T> static inline T __shfl_up(T v, unsigned delta, int width = 64) {
  auto s = hipemu::exchange(v);
  int lane = hipemu::g.cur->lane;
  int in = lane % width;
  return in >= (int)delta ? hipemu::peek<T>(s, lane - (int)delta) : v;
}
template <class T> static inline T __shfl_down(T v, unsigned delta, int width = 64) {
  auto s = hipemu::exchange(v);
  int lane = hipemu::g.cur->lane;
  int in = lane % width;
  return in + (int)delta < width ? hipemu::peek<T>(s, lane + (int)delta) : v;
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
  auto s = hipemu::exchange(v);
  int lane = hipemu::g.cur->lane;
  int tgt = lane ^ mask;
  return (tgt / width == lane / width) ? hipemu::peek<T>(s, tgt) : v;
}
static inline unsigned long long __ballot(int pred) {
  auto s = hipemu::exchange(pred);
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l)
    if (l < hipemu::wave().alive && hipemu::peek<int>(s, l)) m |= 1ull << l;
  return m;
}
template <class T> static inline T __builtin_amdgcn_readfirstlane_emu(T v) {
  auto s = hipemu::exchange(v);
  return hipemu::peek<T>(s, 0);
}
#define __builtin_amdgcn_readfirstlane(v) __builtin_amdgcn_readfirstlane_emu(v)
// DPP subset used by the kernels: quad_perm (ctrl < 0x100), row_shr:n (0x111 .. 0x11F), row_mirror (0x140), row_half_mirror
// (0x141), row_bcast15 (0x142), row_bcast31 (0x143).  A lane whose row (bank) the row_mask (bank_mask) disables keeps `old`; a lane
// without a source lane gets 0 with bound_ctrl, `old` without.
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  auto s = hipemu::exchange(src);
  const int l = hipemu::g.cur->lane;
  int from = -1;
  if (ctrl < 0x100) from = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
  else if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl - 0x110; from = (l & 15) >= n ? l - n : -1; }
  else if (ctrl == 0x141) from = (l & ~7) | (7 - (l & 7));
  else if (ctrl == 0x140) from = (l & ~15) | (15 - (l & 15));
  else if (ctrl == 0x142) from = (l >> 4) >= 1 ? ((l >> 4) - 1) * 16 + 15 : -1;
  else if (ctrl == 0x143) from = (l >> 5) >= 1 ? 31 : -1;
  else { fprintf(stderr, "hipemu: unsupported dpp_ctrl 0x%x\n", ctrl); abort(); }
  const int got = from >= 0 ? hipemu::peek<int>(s, from) : 0;       // (every lane takes part in the exchange)
  if (!((row_mask >> (l >> 4)) & 1) || !((bank_mask >> ((l & 15) >> 2)) & 1)) return old;
  if (from < 0) return bound_ctrl ? 0 : old;
  return got;
}
static inline int __builtin_amdgcn_mov_dpp(int src, int ctrl, int rm, int bm, bool bc) {
  return __builtin_amdgcn_update_dpp(0, src, ctrl, rm, bm, bc);
}
static inline int __builtin_amdgcn_readlane(int v, int lane) {
  auto s = hipemu::exchange(v);
  return hipemu::peek<int>(s, lane);
}
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)  /* the emulator's memory is sequentially consistent per thread */
#define __builtin_amdgcn_s_getreg(x) (0)          /* hardware status registers: placement only, never data */
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_sched_barrier(a) ((void)0)
#define __builtin_amdgcn_s_barrier() __syncthreads()

// ---- f32 MFMA (exact k-ordered fmaf chain, as on gfx950) ----------------------------------------
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
// A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D: col=l&15, row=(l>>4)*4+reg
static inline hipemu_f32x4 hipemu_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
  float2 ab{a, b};
  auto s = hipemu::exchange(ab);
  int l = hipemu::g.cur->lane;
  int col = l & 15;
  hipemu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) {
      float av = hipemu::peek<float2>(s, k * 16 + row).x;
      float bv = hipemu::peek<float2>(s, k * 16 + col).y;
      acc = fmaf(av, bv, acc);
    }
    d[r] = acc;
  }
  return d;
}
// A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D: col=l&31, row=(reg&3)+8*(reg>>2)+4*(l>>5)
static inline hipemu_f32x16 hipemu_mfma_f32_32x32x2f32(float a, float b, hipemu_f32x16 c, int, int, int) {
  float2 ab{a, b};
  auto s = hipemu::exchange(ab);
  int l = hipemu::g.cur->lane;
  int col = l & 31;
  hipemu_f32x16 d = c;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    for (int k = 0; k < 2; ++k) {
      float av = hipemu::peek<float2>(s, k * 32 + row).x;
      float bv = hipemu::peek<float2>(s, k * 32 + col).y;
      acc = fmaf(av, bv, acc);
    }
    d[r] = acc;
  }
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu_mfma_f32_16x16x4f32
#define __builtin_amdgcn_mfma_f32_32x32x2f32 hipemu_mfma_f32_32x32x2f32

// ---- device math / intrinsics subset ------------------------------------------------------------
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
static inline double __dmul_rn(double a, double b) { volatile double r = a * b; return r; }
static inline double __dadd_rn(double a, double b) { volatile double r = a + b; return r; }
static inline double __fma_rn(double a, double b, double c) { return fma(a, b, c); }
// buffer descriptors (raw, stride 0): base + byte count; out-of-range loads return 0, out-of-range stores are dropped
struct hipemu_rsrc { char* base; unsigned bytes; };
typedef hipemu_rsrc __amdgpu_buffer_rsrc_t;
static inline hipemu_rsrc __builtin_amdgcn_make_buffer_rsrc(void* p, short, int bytes, int) {
  return hipemu_rsrc{static_cast<char*>(p), bytes > 0 ? (unsigned)bytes : 0u};
}
static inline unsigned __builtin_amdgcn_raw_buffer_load_b32(hipemu_rsrc r, int voff, int soff, int) {
  const unsigned long long o = (unsigned long long)(unsigned)voff + (unsigned)soff;
  if (o + 4 > r.bytes) return 0u;
  unsigned v;
  memcpy(&v, r.base + o, 4);
  return v;
}
static inline void __builtin_amdgcn_raw_buffer_store_b32(unsigned v, hipemu_rsrc r, int voff, int soff, int) {
  const unsigned long long o = (unsigned long long)(unsigned)voff + (unsigned)soff;
  if (o + 4 > r.bytes) return;
  memcpy(r.base + o, &v, 4);
}
static inline float __builtin_amdgcn_sinf(float turns) { return (float)sin(2.0 * M_PI * (double)turns); }   // v_sin_f32
static inline float __builtin_amdgcn_cosf(float turns) { return (float)cos(2.0 * M_PI * (double)turns); }   // v_cos_f32
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }                                            // v_rcp_f32
static inline float __builtin_amdgcn_logf(float x) { return log2f(x); }                                         // v_log_f32
static inline float __builtin_amdgcn_sqrtf(float x) { return sqrtf(x); }                                        // v_sqrt_f32
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }                                         // v_exp_f32
static inline float __builtin_amdgcn_fractf(float x) { return x - floorf(x); }                                    // v_fract_f32
#define __sinf(x) sinf(x)
#define __cosf(x) cosf(x)
#define __expf(x) expf(x)
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline void sincospif(float x, float* s, float* c) { *s = (float)sin(M_PI * (double)x); *c = (float)cos(M_PI * (double)x); }
static inline float sinpif(float x) { return (float)sin(M_PI * (double)x); }
static inline float cospif(float x) { return (float)cos(M_PI * (double)x); }
static inline double sinpi(double x) { return sin(M_PI * x); }
static inline void sincospi(double x, double* s, double* c) { *s = sin(M_PI * x); *c = cos(M_PI * x); }
static inline double cospi(double x) { return cos(M_PI * x); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline int __float2int_rn(float x) { return (int)rintf(x); }
static inline int __float2int_rd(float x) { return (int)floorf(x); }
static inline float __int2float_rn(int x) { return (float)x; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T min(T a, T b) { return a < b ? a : b; }
template <class T> static inline T max(T a, T b) { return a > b ? a : b; }
