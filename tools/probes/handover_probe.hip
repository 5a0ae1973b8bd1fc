// handover_probe — what does an IN-LAUNCH hand-over of FeedForward's intermediate cost against the kernel boundary it would replace?
// ("one measured prototype of cross-launch fusion that avoids device-scope fences ... sc1 / nt stores + loads of a flag
// and the tile, producer / consumer pairs on the same XCD ... or a number that closes the avenue".)
//
// Geometry of the B = 1 FeedForward pair: FF1 = 15 row panels x 16 column tiles of 192 x 128 outputs (240 workgroups, ONE
// round, 48 KB of MX operand lines each = 11.5 MB), FF2 = 30 x 8 tiles of 96 x 128 (240 workgroups); an FF2 tile reads its 96 rows of ALL 16
// FF1 column tiles (16 x 24 KB).  The arithmetic is left out on purpose: producers write their 48 KB, consumers read their 384 KB and reduce
// them to a checksum — the probe prices the MECHANISM between the two:
//   two    producer kernel, consumer kernel (what the engine does: a dependent kernel boundary)
//   fused  one launch of 240 workgroups; a workgroup writes its tile with write-through (sc1) 16-byte stores, drains them (vmcnt 0), bumps
//          its row panel's counter (one relaxed device-scope atomic), then polls the counter of the panel it CONSUMES with sc1 loads +
//          s_sleep and reads the 16 half-tiles with sc1 loads.  No fence anywhere.  Producer and consumer panels share an XCD by construction
//          (see place()).
//   free   the fused launch with the wait removed (consumers read whatever is there): the floor of the fused form
//   hipcc --offload-arch=gfx950 -O2 -o handover_probe handover_probe.hip && ./handover_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int NP = 15, NC = 16, NWG = NP * NC;     // producer tiles: 15 row panels x 16 column tiles
constexpr int TILE_B = 48 * 1024;                  // bytes a producer writes
constexpr int HALF_B = TILE_B / 2;                 // what one consumer reads of each of its panel's 16 tiles
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// workgroup w of the launch -> (panel, column) in the order of the engine's GEMM kernels: the hardware deals workgroups round-robin over
// the 8 XCDs, and XCD x takes the contiguous run of 30 tiles [30 x, 30 x + 30) = 1.875 row panels x 16 column tiles.  The consumers use the
// same numbering (tile L = row half (L % 16) / 8 of panel L / 16), so a panel's consumers sit on the XCD of (most of) its producers — 7 of the
// 15 panels straddle two XCDs, as they do in the real launch.
__device__ __forceinline__ void place(int w, int& panel, int& col) {
  const int L = (w & 7) * (NWG / 8) + (w >> 3);
  panel = L / NC;
  col = L - panel * NC;
}

__device__ __forceinline__ void store_sc1(u32x4* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ uint32_t load_flag(const uint32_t* p) {
  uint32_t v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <bool SC1>
__device__ __forceinline__ void produce(char* tiles, int panel, int col, uint32_t tag) {
  u32x4* t = reinterpret_cast<u32x4*>(tiles + (size_t)(panel * NC + col) * TILE_B);
  const u32x4 v = {tag, tag + 1u, tag + 2u, (uint32_t)threadIdx.x};
  for (int i = threadIdx.x; i < TILE_B / 16; i += blockDim.x) {
    if (SC1) store_sc1(t + i, v);
    else t[i] = v;
  }
}
// three 16-byte write-through-coherent loads in flight per lane and their wait as ONE asm statement: an asm load's destination is "written" for
// the compiler the moment the statement is issued — split from its wait, hipcc reused the landing registers as addresses of the next loads
// (first version of this probe: wild addresses, the launch never came back)
__device__ __forceinline__ void load3_sc1(const u32x4* p0, const u32x4* p1, const u32x4* p2, u32x4& a, u32x4& b, u32x4& c) {
  asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx4 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(a), "=&v"(b), "=&v"(c)
               : "v"(p0), "v"(p1), "v"(p2)
               : "memory");
}
template <bool SC1>
__device__ __forceinline__ uint32_t consume(const char* tiles, int panel, int half) {
  static_assert(HALF_B / 16 == 6 * 256, "two batches of three 16-byte pieces per lane and tile");
  uint32_t acc = 0;
  for (int c = 0; c < NC; ++c) {
    const u32x4* t = reinterpret_cast<const u32x4*>(tiles + (size_t)(panel * NC + c) * TILE_B + (size_t)half * HALF_B) + threadIdx.x;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      u32x4 a0, a1, a2;
      if (SC1) load3_sc1(t + 768 * b, t + 768 * b + 256, t + 768 * b + 512, a0, a1, a2);
      else { a0 = t[768 * b]; a1 = t[768 * b + 256]; a2 = t[768 * b + 512]; }
      acc += (a0[0] ^ a0[3]) + (a1[0] ^ a1[3]) + (a2[0] ^ a2[3]);
    }
  }
  return acc;
}

__global__ __launch_bounds__(256) void producer_kernel(char* tiles, uint32_t tag) {
  int panel, col;
  place(blockIdx.x, panel, col);
  produce<false>(tiles, panel, col, tag);
}
__global__ __launch_bounds__(256) void consumer_kernel(const char* tiles, uint32_t* out) {
  int panel, col;
  place(blockIdx.x, panel, col);
  const uint32_t a = consume<false>(tiles, panel, col >> 3);  // tiles 0..7 of a panel's run read its first 96 rows, 8..15 the second
  if (a == 0xdeadbeefu) out[blockIdx.x] = a;
}
// MODE 0: wait for the panel's 16 producers (sc1 traffic, no fence); 1: no wait (floor); 2: the round-5 form — plain stores, device-scope
// release fence before the counter, acquire fence behind the wait, plain loads
template <int MODE>
__global__ __launch_bounds__(256) void fused_kernel(char* tiles, uint32_t* flags, uint32_t* out, uint32_t tag, uint32_t round) {
  int panel, col;
  place(blockIdx.x, panel, col);
  constexpr bool SC1 = MODE != 2;
  produce<SC1>(tiles, panel, col, tag);
  if (SC1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my write-through stores have left the CU
  else __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(flags + panel * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // one counter per panel, 128 B apart.
  // SYSTEM scope = global_atomic_add ... sc1, performed at the memory side: the agent-scope add hipcc emits (no sc1) stays in the issuing XCD's
  // L2, and a panel whose 16 producers straddle two XCDs never counts to 16 for a poller that reads memory (first run of this probe: every
  // wait gave up)
  if (MODE != 1) {
    if (threadIdx.x == 0) {
      int spins = 0;
      while (load_flag(flags + panel * 32) < NC * round && ++spins < (1 << 14)) __builtin_amdgcn_s_sleep(2);
      if (spins >= (1 << 14)) atomicAdd(out + NWG, 1u);  // (never hang the box: a wait that gives up is counted and reported)
    }
    __syncthreads();
    if (!SC1) __threadfence();
  }
  const uint32_t a = consume<SC1>(tiles, panel, col >> 3);
  if (a == 0xdeadbeefu) out[blockIdx.x] = a;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  char* tiles; uint32_t *flags, *out;
  CK(hipMalloc(&tiles, (size_t)NWG * TILE_B)); CK(hipMalloc(&flags, NP * 32 * 4)); CK(hipMalloc(&out, (NWG + 1) * 4)); CK(hipMemset(out, 0, (NWG + 1) * 4));
  CK(hipMemset(tiles, 0, (size_t)NWG * TILE_B)); CK(hipMemset(flags, 0, NP * 32 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 100;
  float ms;
  auto run = [&](const char* name, auto body) {
    for (int i = 0; i < 20; ++i) body(i);  // warm-up (continues the flag rounds)
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) body(20 + i);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    uint32_t gave_up = 0;
    hipMemcpy(&gave_up, out + NWG, 4, hipMemcpyDeviceToHost);
    printf("%-58s %7.2f us per FeedForward hand-over%s\n", name, 1e3f * ms / iters, gave_up ? "   (WAITS GAVE UP: not a measurement)" : "");
    fflush(stdout);
    hipMemset(out + NWG, 0, 4);
    return 0;
  };
  run("two launches (producer kernel | consumer kernel)", [&](int i) {
    hipLaunchKernelGGL(producer_kernel, dim3(NWG), dim3(256), 0, 0, tiles, (uint32_t)i);
    hipLaunchKernelGGL(consumer_kernel, dim3(NWG), dim3(256), 0, 0, tiles, out);
  });
  run("producer kernel alone", [&](int i) { hipLaunchKernelGGL(producer_kernel, dim3(NWG), dim3(256), 0, 0, tiles, (uint32_t)i); });
  run("consumer kernel alone", [&](int) { hipLaunchKernelGGL(consumer_kernel, dim3(NWG), dim3(256), 0, 0, tiles, out); });
  auto dump = [&](uint32_t rounds) {
    std::vector<uint32_t> f(NP * 32);
    hipMemcpy(f.data(), flags, f.size() * 4, hipMemcpyDeviceToHost);
    printf("    panel counters after %u launches (16 arrivals each):", rounds);
    for (int p = 0; p < NP; ++p) printf(" %u", f[p * 32]);
    printf("\n");
    fflush(stdout);
  };
  uint32_t round = 0;
  run("one launch, no wait (floor of the fused form)", [&](int i) {
    hipLaunchKernelGGL(fused_kernel<1>, dim3(NWG), dim3(256), 0, 0, tiles, flags, out, (uint32_t)i, 0u);
  });
  CK(hipMemset(flags, 0, NP * 32 * 4));
  round = 0;
  run("one launch, plain tile + release / acquire fences (round 5)", [&](int i) {
    ++round;
    hipLaunchKernelGGL(fused_kernel<2>, dim3(NWG), dim3(256), 0, 0, tiles, flags, out, (uint32_t)i, round);
  });
  dump(round);
  CK(hipMemset(flags, 0, NP * 32 * 4));
  round = 0;
  run("one launch, sc1 tile + counter per panel, same-XCD pairs", [&](int i) {
    ++round;
    hipLaunchKernelGGL(fused_kernel<0>, dim3(NWG), dim3(256), 0, 0, tiles, flags, out, (uint32_t)i, round);
  });
  dump(round);
  CK(hipDeviceSynchronize());
  return 0;
}
