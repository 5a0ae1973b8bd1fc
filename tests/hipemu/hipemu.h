// hipemu.h — a minimal HIP-on-CPU shim for THIS repo's kernels (TEST INFRASTRUCTURE; host clang++, -DF5_HIPEMU).
//
// Purpose: run the very kernel source that hipcc compiles for gfx950 (csrc/*.h kernels) on the CPU, thread for thread, so index
// arithmetic, LDS images, MFMA fragment layouts and epilogues can be checked without a GPU.  One fiber per HIP thread, switched at
// barriers only; __syncthreads() is a block barrier; the few amdgcn builtins the kernels use are emulated with their DOCUMENTED semantics:
//   * v_mfma_f32_32x32x16_f16 / v_mfma_f32_32x32x2_f32: A[i][k] from lane i + 32 (k / kpl), B[k][j] from lane j + 32 (k / kpl)
//     (kpl = k elements per lane: 8 / 1), D[i][j] in lane j + 32 ((i / 4) % 2), register (i % 4) + 4 (i / 8)   (common.h; proven on
//     the GPU by the DiT parity suite — tests/test_hipemu.py first runs the GPU-proven gemm_kernel through this shim);
//   * raw buffer loads: dwords at byte offsets >= num_records read as zero;
//   * LDS: plain per-block memory; exp2 / rcp: libm.
// Not emulated: inline asm (LDS-DMA ring / stream-K kernels are compiled out under F5_HIPEMU), timing, caches, memory ordering.
#pragma once
#define F5_HIPEMU 1
#include <chrono>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
// a kernel's static LDS array: one block at a time runs on an OS thread (hipemu::launch), so per-thread storage is per-block storage
// (NOT valid under launch_coop, whose kernels use dynamic LDS only)
#define __shared__ static thread_local
static inline unsigned __brev(unsigned x) { return __builtin_bitreverse32(x); }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }

struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
struct int2 { int x, y; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef void* hipGraphExec_t;
// the slice of the HIP runtime API the host side of the BigVGAN path calls (bigvgan_api.cpp, DevBuf): "device" memory is host memory
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
// every allocation sits between two 4 KiB guard zones of a known pattern, checked when it is freed: a kernel that WRITES outside its
// buffers aborts the test instead of silently corrupting a neighbour (reads outside surface as wrong results)
namespace hipemu {
// stream capture: while a capture is open on this thread, launches and async copies are RECORDED (closures over their by-value
// arguments, exactly what a graph node holds) instead of executed; hipGraphLaunch runs the record in order.  Serial replay is one valid
// schedule of the captured graph — fork / join through events (the two-chain schedule) therefore needs nothing extra.
struct GraphRec { std::vector<std::function<void()>> nodes; };
inline thread_local GraphRec* capturing = nullptr;
constexpr size_t kGuard = 4096;
struct AllocHdr { size_t n; };
inline void check_guards(void* p) {
  char* base = static_cast<char*>(p) - kGuard;
  const size_t n = reinterpret_cast<AllocHdr*>(base)->n;
  for (size_t i = sizeof(AllocHdr); i < kGuard; ++i)
    if (base[i] != (char)0xA5) { fprintf(stderr, "hipemu: write BEFORE a device buffer (offset -%zu)\n", kGuard - i); abort(); }
  for (size_t i = 0; i < kGuard; ++i)
    if (base[kGuard + n + i] != (char)0xA5) { fprintf(stderr, "hipemu: write PAST a device buffer of %zu bytes (+%zu)\n", n, i); abort(); }
}
}  // namespace hipemu
static inline hipError_t hipMalloc(void** p, size_t n) {
  char* base = static_cast<char*>(aligned_alloc(4096, ((n + 4095) & ~size_t(4095)) + 2 * hipemu::kGuard));
  if (!base) return hipErrorOutOfMemory;
  memset(base, 0xA5, hipemu::kGuard);
  memset(base + hipemu::kGuard + n, 0xA5, hipemu::kGuard);
  reinterpret_cast<hipemu::AllocHdr*>(base)->n = n;
  *p = base + hipemu::kGuard;
  return hipSuccess;
}
static inline hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  hipemu::check_guards(p);
  free(static_cast<char*>(p) - hipemu::kGuard);
  return hipSuccess;
}
// (a synchronous copy inside a thread-local capture is an error on the GPU as well: hipErrorStreamCaptureImplicit)
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n); return *p ? hipSuccess : 2; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { if (hipemu::capturing) return 906; memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (hipemu::capturing) return 906; memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
  if (hipemu::capturing) hipemu::capturing->nodes.push_back([=] { memcpy(d, s, n); });
  else memcpy(d, s, n);
  return hipSuccess;
}
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu error"; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
// streams / events / graphs: everything runs synchronously on the calling thread; stream capture records closures (hipemu::GraphRec)
typedef void* hipGraph_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipStreamCaptureModeThreadLocal = 1, hipErrorNotSupported = 801 };
enum { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(uintptr_t(1)); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
// launches run synchronously on the shim, so an event's "completion time" is the host clock at the moment it is recorded
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new double(0.0); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete static_cast<double*>(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  *static_cast<double*>(e) = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  return hipSuccess;
}
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = (float)(*static_cast<double*>(b) - *static_cast<double*>(a));
  return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
  if (hipemu::capturing) hipemu::capturing->nodes.push_back([=] { memset(p, v, n); });
  else memset(p, v, n);
  return hipSuccess;
}
static inline hipError_t hipStreamBeginCapture(hipStream_t, int) {
  if (hipemu::capturing) return hipErrorInvalidValue;
  hipemu::capturing = new hipemu::GraphRec();
  return hipSuccess;
}
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) {
  if (!hipemu::capturing) return hipErrorInvalidValue;
  *g = hipemu::capturing;
  hipemu::capturing = nullptr;
  return hipSuccess;
}
static inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t) {
  if (!g) return hipErrorInvalidValue;
  *e = new hipemu::GraphRec(*static_cast<hipemu::GraphRec*>(g));
  return hipSuccess;
}
static inline hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) {
  if (!e || hipemu::capturing) return hipErrorInvalidValue;
  for (auto& node : static_cast<hipemu::GraphRec*>(e)->nodes) node();
  return hipSuccess;
}
static inline hipError_t hipGraphDestroy(hipGraph_t g) { delete static_cast<hipemu::GraphRec*>(g); return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete static_cast<hipemu::GraphRec*>(e); return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 256; return hipSuccess; }
using std::max;
using std::min;

namespace hipemu {
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

struct Rsrc {
  const char* base;
  uint32_t num_records;
};
struct Idx { unsigned x, y, z; };

// Execution model: the HIP threads of one block are FIBERS of one OS thread (hand-rolled x86-64 context switch, ~10 ns), switched
// only at barriers: __syncthreads() and the wave-wide rendezvous inside the emulated MFMA / shuffle.  Blocks are independent (the
// kernels run here never communicate across blocks) and are dealt to a few OS threads.
struct Barrier {
  int expected = 0, arrived = 0;
  unsigned gen = 0;
};
struct Fiber {
  void* sp = nullptr;
  Idx tidx{};
  int lane = 0, wave = 0;
  bool done = false;
  Barrier* wait_bar = nullptr;
  unsigned wait_gen = 0;
};
struct WaveCtx {
  Barrier bar;
  float a[64][8], b[64][8], sh[64];
  float qa[64][32], qb[64][32];  // decoded MX operands (mfma_scale_32x32x64_fp6)
};
struct BlockCtx {
  Barrier bar;
  std::vector<WaveCtx> waves;
  std::vector<char> lds;
  std::vector<Fiber> fibers;
  Fiber* cur = nullptr;
  void* sched_sp = nullptr;
  Idx bidx{};
  const std::function<void()>* fn = nullptr;
};
inline thread_local BlockCtx* blk = nullptr;
inline thread_local Idx b_dim, g_dim;

extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.weak hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipemu_switch,.-hipemu_switch
)");

inline void yield_to_scheduler() { hipemu_switch(&blk->cur->sp, blk->sched_sp); }
inline void barrier_wait(Barrier& b) {
  const unsigned g = b.gen;
  if (++b.arrived == b.expected) { b.arrived = 0; ++b.gen; return; }
  blk->cur->wait_bar = &b;
  blk->cur->wait_gen = g;
  yield_to_scheduler();
}
inline void barrier_drop(Barrier& b) {  // a thread that has returned no longer takes part in barriers (as on the GPU)
  if (--b.expected > 0 && b.arrived == b.expected) { b.arrived = 0; ++b.gen; }
}
inline void fiber_entry() {
  BlockCtx* c = blk;
  Fiber* f = c->cur;
  (*c->fn)();
  f->done = true;
  barrier_drop(c->waves[f->wave].bar);
  barrier_drop(c->bar);
  yield_to_scheduler();
  abort();  // a finished fiber is never resumed
}

inline char* dyn_lds() { return reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(blk->lds.data()) + 63) & ~uintptr_t(63)); }
inline WaveCtx& wv() { return blk->waves[blk->cur->wave]; }

inline Rsrc make_rsrc(void* p, short, int num, int) { return {reinterpret_cast<const char*>(p), (uint32_t)num}; }
inline u4 raw_buffer_load_b128(Rsrc r, int off, int soff, int) {
  u4 v = {0, 0, 0, 0};
  const uint32_t o = (uint32_t)off + (uint32_t)soff;
  for (int i = 0; i < 4; ++i) {
    const uint64_t b = (uint64_t)o + 4u * i;
    if (b + 4 <= r.num_records) { unsigned t; memcpy(&t, r.base + b, 4); v[i] = t; }
  }
  return v;
}
inline f16v mfma_32x32x16_f16(h8 a, h8 b, f16v c, int, int, int) {
  WaveCtx& w = wv();
  const int lane = blk->cur->lane;
  for (int e = 0; e < 8; ++e) { w.a[lane][e] = (float)a[e]; w.b[lane][e] = (float)b[e]; }
  barrier_wait(w.bar);
  const int j = lane & 31, hi = lane >> 5;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float s = 0.f;
    for (int k = 0; k < 16; ++k) s += w.a[i + 32 * (k >> 3)][k & 7] * w.b[j + 32 * (k >> 3)][k & 7];
    c[r] += s;
  }
  barrier_wait(w.bar);
  return c;
}
inline f16v mfma_32x32x2_f32(float a, float b, f16v c, int, int, int) {
  WaveCtx& w = wv();
  const int lane = blk->cur->lane;
  w.a[lane][0] = a; w.b[lane][0] = b;
  barrier_wait(w.bar);
  const int j = lane & 31, hi = lane >> 5;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
    c[r] += w.a[i][0] * w.b[j][0] + w.a[i + 32][0] * w.b[j + 32][0];
  }
  barrier_wait(w.bar);
  return c;
}
// ---- MX-fp6 (gfx950; semantics measured on the GPU by tools/probes/mx6_probe.hip, profiles/r04a_mx6_probe.log) ---------------------
// e2m3: sign | 2 exponent bits (bias 1) | 3 mantissa bits; values 0, 0.125 .. 0.875 (subnormal), 1 .. 7.5.  Element e of a 32-element
// operand sits in bits [6 e, 6 e + 6) of the little-endian register tuple.
typedef unsigned u6 __attribute__((ext_vector_type(6)));
typedef int i8v __attribute__((ext_vector_type(8)));
inline float fp6_decode(unsigned c) {
  const unsigned e = (c >> 3) & 3, m = c & 7;
  const float v = e == 0 ? m / 8.0f : ldexpf(1.0f + m / 8.0f, (int)e - 1);
  return (c & 32) ? -v : v;
}
inline unsigned fp6_encode(float x) {  // round to nearest even, saturate at 7.5, keep the sign of zero
  const unsigned s = std::signbit(x) ? 32u : 0u;
  const float a = fabsf(x);
  if (!(a == a)) return s | 31u;
  unsigned c;
  if (a < 2.0f) c = (unsigned)nearbyintf(a * 8.0f);
  else if (a < 4.0f) c = 8u + (unsigned)nearbyintf(a * 4.0f);
  else if (a < 7.75f) c = 16u + (unsigned)nearbyintf(a * 2.0f);
  else c = 31u;
  return s | (c > 31u ? 31u : c);
}
inline unsigned fp6_get(const unsigned* w, int e) {
  const int bit = 6 * e;
  const uint64_t two = w[bit >> 5] | ((uint64_t)((bit >> 5) + 1 < 6 ? w[(bit >> 5) + 1] : 0u) << 32);
  return (unsigned)(two >> (bit & 31)) & 63u;
}
// v_cvt_scalef32_2xpk16_fp6_f32: element 2 i <- src0[i] / 2^floor(log2 scale), element 2 i + 1 <- src1[i] / the same
inline u6 cvt_scalef32_2xpk16_fp6_f32(f16v s0, f16v s1, float scale) {
  int ex;
  frexpf(scale, &ex);
  const float inv = ldexpf(1.0f, -(ex - 1));
  unsigned w[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 32; ++i) {
    const unsigned c = fp6_encode(((i & 1) ? s1[i >> 1] : s0[i >> 1]) * inv);
    const int bit = 6 * i;
    const uint64_t v = (uint64_t)c << (bit & 31);
    w[bit >> 5] |= (unsigned)v;
    w[(bit >> 5) + 1] |= (unsigned)(v >> 32);
  }
  u6 r;
  for (int i = 0; i < 6; ++i) r[i] = w[i];
  return r;
}
// v_cvt_scalef32_pk32_fp6_f16: element i <- src[i] / 2^floor(log2 scale) (sequential; profiles/r04i_cvt_pk32_probe.log)
typedef _Float16 h32v __attribute__((ext_vector_type(32)));
inline u6 cvt_scalef32_pk32_fp6_f16(h32v s0, float scale) {
  int ex;
  frexpf(scale, &ex);
  const float inv = ldexpf(1.0f, -(ex - 1));
  unsigned w[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 32; ++i) {
    const unsigned c = fp6_encode((float)s0[i] * inv);
    const int bit = 6 * i;
    const uint64_t v = (uint64_t)c << (bit & 31);
    w[bit >> 5] |= (unsigned)v;
    w[(bit >> 5) + 1] |= (unsigned)(v >> 32);
  }
  u6 r;
  for (int i = 0; i < 6; ++i) r[i] = w[i];
  return r;
}
// v_mfma_scale_f32_32x32x64_f8f6f4 with cbsz = blgp = 2 (both operands e2m3): lane l of src0 holds A[i = l % 32][k = 32 (l / 32) + e],
// registers 6 and 7 of the 8-register operand are ignored; the per-lane scale is byte 0 of the scale register, 2^(b - 127)
inline f16v mfma_scale_32x32x64_fp6(i8v a, i8v b, f16v c, int cbsz, int blgp, int, int sa, int, int sb) {
  if (cbsz != 2 || blgp != 2) { fprintf(stderr, "hipemu: only e2m3 x e2m3 is emulated\n"); abort(); }
  WaveCtx& w = wv();
  const int lane = blk->cur->lane;
  unsigned ua[6], ub[6];
  for (int i = 0; i < 6; ++i) { ua[i] = (unsigned)a[i]; ub[i] = (unsigned)b[i]; }
  const float fa = ldexpf(1.0f, (sa & 255) - 127), fb = ldexpf(1.0f, (sb & 255) - 127);
  for (int e = 0; e < 32; ++e) { w.qa[lane][e] = fp6_decode(fp6_get(ua, e)) * fa; w.qb[lane][e] = fp6_decode(fp6_get(ub, e)) * fb; }
  barrier_wait(w.bar);
  const int j = lane & 31, hi = lane >> 5;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float s = 0.f;
    for (int k = 0; k < 64; ++k) s += w.qa[i + 32 * (k >> 5)][k & 31] * w.qb[j + 32 * (k >> 5)][k & 31];
    c[r] += s;
  }
  barrier_wait(w.bar);
  return c;
}
inline float shfl_xor(float v, int mask, int) {
  WaveCtx& w = wv();
  const int lane = blk->cur->lane;
  w.sh[lane] = v;
  barrier_wait(w.bar);
  const float r = w.sh[lane ^ mask];
  barrier_wait(w.bar);
  return r;
}

// wave vote: does any lane of the wave hold a non-zero predicate?  (the kernels use it through f5_wave_any, common.h)
inline bool wave_any(bool pred) {
  WaveCtx& w = wv();
  const int lane = blk->cur->lane;
  w.sh[lane] = pred ? 1.0f : 0.0f;
  barrier_wait(w.bar);
  bool r = false;
  for (int l = 0; l < w.bar.expected; ++l) r = r || w.sh[l] != 0.0f;
  barrier_wait(w.bar);
  return r;
}

inline unsigned shfl_xor_u32(unsigned v, int mask) {
  WaveCtx& w = wv();
  const int lane = blk->cur->lane;
  memcpy(&w.sh[lane], &v, 4);
  barrier_wait(w.bar);
  unsigned r;
  memcpy(&r, &w.sh[lane ^ mask], 4);
  barrier_wait(w.bar);
  return r;
}
inline int shfl_xor(int v, int mask, int) { return (int)shfl_xor_u32((unsigned)v, mask); }  // bits, not a value conversion to float
// v_permlane32_swap_b32 a, b: lanes 32-63 of a exchange with lanes 0-31 of b
inline void permlane32_swap(unsigned& a, unsigned& b) {
  WaveCtx& w = wv();
  const int lane = blk->cur->lane;
  memcpy(&w.sh[lane], lane < 32 ? &b : &a, 4);
  barrier_wait(w.bar);
  unsigned r;
  memcpy(&r, &w.sh[lane ^ 32], 4);
  barrier_wait(w.bar);
  if (lane < 32) b = r; else a = r;
}
inline void raw_buffer_store(Rsrc r, uint32_t off, const void* src, int bytes) {  // out-of-range dwords are dropped
  for (int i = 0; i < bytes; i += (bytes < 4 ? bytes : 4)) {
    const int n = bytes < 4 ? bytes : 4;
    if ((uint64_t)off + i + n <= r.num_records) memcpy(const_cast<char*>(r.base) + off + i, static_cast<const char*>(src) + i, n);
  }
}

constexpr size_t kFiberStack = 128 * 1024;

// one block, all its threads as fibers of the calling OS thread; `stacks` = nt * kFiberStack bytes
inline void run_block(dim3 block, Idx bidx, size_t lds_bytes, const std::function<void()>& fn, char* stacks) {
  const int nt = (int)(block.x * block.y * block.z), nw = (nt + 63) / 64;
  BlockCtx ctx;
  ctx.bar.expected = nt;
  ctx.waves.resize(nw);
  for (int w = 0; w < nw; ++w) ctx.waves[w].bar.expected = std::min(64, nt - 64 * w);
  ctx.lds.resize(lds_bytes + 64);
  ctx.fibers.resize(nt);
  ctx.bidx = bidx;
  ctx.fn = &fn;
  for (int t = 0; t < nt; ++t) {
    Fiber& f = ctx.fibers[t];
    f.tidx = {(unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y)};
    f.lane = t & 63;
    f.wave = t >> 6;
    // initial frame for hipemu_switch: six callee-saved registers, then the entry point as the return address (stack 16-aligned + 8
    // at the entry, as after a call)
    uintptr_t top = (reinterpret_cast<uintptr_t>(stacks) + (size_t)(t + 1) * kFiberStack) & ~uintptr_t(15);
    void** sp = reinterpret_cast<void**>(top);
    *--sp = nullptr;
    *--sp = reinterpret_cast<void*>(&fiber_entry);
    for (int r = 0; r < 6; ++r) *--sp = nullptr;
    f.sp = sp;
  }
  BlockCtx* prev = blk;
  blk = &ctx;
  int remaining = nt, idle = 0;
  for (int i = 0; remaining; i = (i + 1 == nt ? 0 : i + 1)) {
    Fiber& f = ctx.fibers[i];
    if (f.done || (f.wait_bar && f.wait_bar->gen == f.wait_gen)) {
      if (++idle > 2 * nt) { fprintf(stderr, "hipemu: deadlock (a barrier some threads never reach)\n"); abort(); }
      continue;
    }
    idle = 0;
    f.wait_bar = nullptr;
    ctx.cur = &f;
    hipemu_switch(&ctx.sched_sp, f.sp);
    if (f.done) --remaining;
  }
  blk = prev;
}

// ---- cooperative launch: ALL blocks alive at once (fibers of one OS thread), for kernels whose blocks talk to each other through
// global memory (stream-K: flag spins).  A spinning thread calls spin_yield(); the scheduler then runs everybody else.
inline thread_local long spin_count = 0;
inline void spin_yield() {
  if (++spin_count > 50000000L) { fprintf(stderr, "hipemu: spin limit (a flag that is never set?)\n"); abort(); }
  blk->cur->wait_bar = nullptr;
  yield_to_scheduler();
}
inline void launch_coop(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& fn, size_t stack_bytes = 64 * 1024) {
  if (capturing) {
    capturing->nodes.push_back([=] { launch_coop(grid, block, lds_bytes, fn, stack_bytes); });
    return;
  }
  const int nt = (int)(block.x * block.y * block.z), nw = (nt + 63) / 64;
  const long nblocks = (long)grid.x * grid.y * grid.z;
  std::unique_ptr<char[]> stacks(new char[(size_t)nblocks * nt * stack_bytes + 64]);
  std::vector<std::unique_ptr<BlockCtx>> ctxs;
  for (long b = 0; b < nblocks; ++b) {
    ctxs.emplace_back(new BlockCtx());
    BlockCtx& ctx = *ctxs.back();
    ctx.bar.expected = nt;
    ctx.waves.resize(nw);
    for (int w = 0; w < nw; ++w) ctx.waves[w].bar.expected = std::min(64, nt - 64 * w);
    ctx.lds.resize(lds_bytes + 64);
    ctx.fibers.resize(nt);
    ctx.bidx = {(unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y))};
    ctx.fn = &fn;
    for (int t = 0; t < nt; ++t) {
      Fiber& f = ctx.fibers[t];
      f.tidx = {(unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y)};
      f.lane = t & 63;
      f.wave = t >> 6;
      uintptr_t top = (reinterpret_cast<uintptr_t>(stacks.get()) + ((size_t)b * nt + t + 1) * stack_bytes) & ~uintptr_t(15);
      void** sp = reinterpret_cast<void**>(top);
      *--sp = nullptr;
      *--sp = reinterpret_cast<void*>(&fiber_entry);
      for (int r = 0; r < 6; ++r) *--sp = nullptr;
      f.sp = sp;
    }
  }
  b_dim = {block.x, block.y, block.z};
  g_dim = {grid.x, grid.y, grid.z};
  spin_count = 0;
  BlockCtx* prev = blk;
  long remaining = nblocks * nt;
  while (remaining) {
    bool progressed = false;
    for (long b = 0; b < nblocks; ++b) {
      BlockCtx& ctx = *ctxs[b];
      for (int t = 0; t < nt; ++t) {
        Fiber& f = ctx.fibers[t];
        if (f.done || (f.wait_bar && f.wait_bar->gen == f.wait_gen)) continue;
        f.wait_bar = nullptr;
        blk = &ctx;
        ctx.cur = &f;
        hipemu_switch(&ctx.sched_sp, f.sp);
        progressed = true;
        if (f.done) --remaining;
      }
    }
    if (!progressed) { fprintf(stderr, "hipemu: deadlock (every live thread waits at a barrier)\n"); abort(); }
  }
  blk = prev;
}

// run fn() once per HIP thread of a grid x block launch; every block has completed when this returns
inline void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& fn) {
  if (capturing) {  // a kernel node: its arguments were captured by value in fn
    GraphRec* rec = capturing;
    rec->nodes.push_back([=] { launch(grid, block, lds_bytes, fn); });  // replay happens with no capture open
    return;
  }
  const int nt = (int)(block.x * block.y * block.z);
  const long nblocks = (long)grid.x * grid.y * grid.z;
  const int nworkers = (int)std::max(1L, std::min<long>(nblocks, std::min(8u, std::max(1u, std::thread::hardware_concurrency()))));
  std::atomic<long> next{0};
  auto worker = [&] {
    std::unique_ptr<char[]> stacks(new char[(size_t)nt * kFiberStack + 64]);  // uninitialised: only the pages the fibers touch are faulted in
    b_dim = {block.x, block.y, block.z};
    g_dim = {grid.x, grid.y, grid.z};
    for (long b = next++; b < nblocks; b = next++) {
      const Idx bidx = {(unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y))};
      run_block(block, bidx, lds_bytes, fn, stacks.get());
    }
  };
  if (nworkers == 1) { worker(); return; }
  std::vector<std::thread> th;
  for (int w = 0; w < nworkers; ++w) th.emplace_back(worker);
  for (auto& t : th) t.join();
}
}  // namespace hipemu

// kernel<<<grid, block, lds, stream>>>(args...): every block runs to completion before the call returns
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) hipemu::launch((grid), (block), (lds), [=] { kern(__VA_ARGS__); })

#define threadIdx hipemu::blk->cur->tidx
#define blockIdx hipemu::blk->bidx
#define blockDim hipemu::b_dim
#define gridDim hipemu::g_dim
#define __syncthreads() hipemu::barrier_wait(hipemu::blk->bar)
#define __shfl_xor(v, m, ...) hipemu::shfl_xor((v), (m), 64)
#define __amdgpu_buffer_rsrc_t hipemu::Rsrc
#define __builtin_amdgcn_make_buffer_rsrc hipemu::make_rsrc
#define __builtin_amdgcn_raw_buffer_load_b128 hipemu::raw_buffer_load_b128
#define __builtin_amdgcn_mfma_f32_32x32x16_f16 hipemu::mfma_32x32x16_f16
#define __builtin_amdgcn_mfma_f32_32x32x2f32 hipemu::mfma_32x32x2_f32
#define __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4 hipemu::mfma_scale_32x32x64_fp6
#define __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32 hipemu::cvt_scalef32_2xpk16_fp6_f32
#define __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16 hipemu::cvt_scalef32_pk32_fp6_f16
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_readfirstlane(x) (x)
// the two global atomics the kernels use (softmax_rows_kernel's statistics): workgroups run on several host threads
static inline double atomicAdd(double* p, double v) {
  unsigned long long* u = reinterpret_cast<unsigned long long*>(p);
  unsigned long long old = __atomic_load_n(u, __ATOMIC_RELAXED), want;
  double o;
  do {
    memcpy(&o, &old, 8);
    const double n = o + v;
    memcpy(&want, &n, 8);
  } while (!__atomic_compare_exchange_n(u, &old, want, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  return o;
}
static inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) {
  unsigned long long old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
