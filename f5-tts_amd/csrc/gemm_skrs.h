// gemm_skrs.h — stream-K GEMM for SMALL grids with the epilogue REDUCE-SCATTERED over a tile's contributors.
//
// Why (DESIGN.md sections 4 and 8): at B = 1 the efficient workgroup shape (8 waves x 64x64, 256x128 / 128x256 tile, 3-stage LDS-DMA
// ring) yields 88-264 tiles for 256 CUs.  Stream-K (gemm_sk.h) balances the k-loops, but there the workgroup that owns the head of a
// tile adds its partners' partial sums and then runs the WHOLE 256x128 epilogue alone — 14-16 us of VALU work (bias, tanh-GELU, hi/lo
// split) on one CU while the partners idle; measured 47 / 59 / 84 us against 33 / 54 / 86 for the plain tiling.  Here every one of
// the c workgroups that contributed k-tiles to a tile finishes 1/c of it.
//
// Schedule (same as gemm_sk.h): workgroup b belongs to XCD class x = b & 7 and is the i = b >> 3 -th of its class; the tiles are cut
// into 8 contiguous runs, one per class (producers and consumers of a tile share an L2); inside a class the iterations
// [0, n_x * KT) are cut into G/8 equal contiguous shares.  A share is a sequence of segments (tile, [k0, k1)):
//     FULL    k0 == 0, k1 == KT   the workgroup has the whole tile: plain epilogue from registers
//     HEAD    k0 == 0, k1 <  KT   only as the LAST segment of a share
//     TAIL /  k0 >  0             only as the FIRST segment of a share (a MIDDLE, k1 < KT as well, is then the whole share)
//     MIDDLE
// The contributors of a tile are the workgroups i0 < i1 < ... of the class whose shares meet it (i0 has the HEAD).  The tile's
// accumulator is 16 UNITS per lane (unit u = one float4 = 4 output channels of one row: the granularity of the epilogue); unit u
// belongs to contributor number u % c.
//
// Protocol per workgroup (tests/test_sk_schedule.py models it event by event; tests/test_hipemu.py RUNS this source on the CPU):
//   * a partial segment's accumulators go to one of the workgroup's two 128 KB workspace slots — slot A for a TAIL / MIDDLE (the
//     first segment), slot B for a HEAD (the last) — then the slot's flag is set to the number of partners that will read it.
//     Publishing never waits for anything.
//   * finishing is done at the END of the share, never earlier: first the HEAD tile, then the tile of the first segment (own partial
//     sums are re-read from the slot they went to, like the partners': one rolled loop, no register pressure).  For each unit it owns, the workgroup adds the c partial sums in
//     contributor order (deterministic) — its partners' from their slots (flag spin, agent-scope loads that miss the per-CU cache and
//     hit the shared L2) — and runs the epilogue on the unit.  After reading a partner's slot it decrements that slot's flag; the
//     last reader leaves it at 0, so the workspace is clean for the next launch and a captured graph can replay the node.
//   * waits are only ever on "published" events, which happen unconditionally after a bounded amount of work => no deadlock as long
//     as a tile's contributors can all be resident; the launcher sizes the grid to the CU count (one 144 KB workgroup per CU).  The
//     spin is bounded (err word) so a logic error cannot hang the GPU.
// Same-L2 hand-over without cache write-backs as in gemm_sk.h: plain stores reach the L2 (write-through L1), acknowledged (vmcnt 0)
// before the flag store.
#pragma once
#include "gemm.h"

// ---- primitives: amdgcn instructions on the GPU, plain memory operations under the host shim (tests/hipemu) ------------------------
#ifdef F5_HIPEMU
namespace skp {
template <int N>
inline void wait_vmcnt() {}  // the shim's LDS-DMA is synchronous
inline void wg_barrier() { __syncthreads(); }
inline uint32_t lds_base(char*) { return 0; }
inline uint4 lds_read_b128(uint32_t addr) { uint4 v; memcpy(&v, hipemu::dyn_lds() + addr, 16); return v; }
inline void lds_wait() {}
inline void dma_b128(BufRsrc r, char* lds_dst, uint32_t voff) {  // each lane: 16 bytes -> lds_dst + lane * 16
  const auto v = hipemu::raw_buffer_load_b128(r, (int)voff, 0, 0);
  memcpy(lds_dst + 16 * hipemu::blk->cur->lane, &v, 16);
}
typedef float f32x4v __attribute__((ext_vector_type(4)));
inline f32x4v load_sc1_async(const float* p) { f32x4v v; memcpy(&v, p, 16); return v; }
inline void settle(f32x4v&) {}
inline int flag_load(const int* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
inline void flag_store(int* p, int v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }
inline void flag_dec(int* p) { __atomic_fetch_sub(p, 1, __ATOMIC_RELAXED); }
inline void err_set(int* p) { __atomic_fetch_or(p, 1, __ATOMIC_RELAXED); }
inline void spin_pause() { hipemu::spin_yield(); }
inline int uniform(int v) { return v; }
}  // namespace skp
#else
namespace skp {
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ uint32_t lds_base(char* smem) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem; }
// inline asm: hipcc drains vmcnt(0) before any ds_read it can see while an LDS-DMA is in flight (gemm.h gemm_glds_kernel)
__device__ __forceinline__ uint4 lds_read_b128(uint32_t addr) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void dma_b128(BufRsrc r, char* lds_dst, uint32_t voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, (int)voff, 0, 0, 0);
}
// agent-scope (sc1) load that misses the per-CU cache and hits the L2 the partner wrote through.  Issued WITHOUT a wait so that the
// loads of all partners are in flight together; settle(v) is the wait, tied to v ("+v") so no use of v can be scheduled before it.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v load_sc1_async(const float* p) {
  f32x4v v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void settle(f32x4v& v) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory"); }
__device__ __forceinline__ int flag_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void flag_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void flag_dec(int* p) { (void)__hip_atomic_fetch_add(p, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void err_set(int* p) { atomicOr(p, 1); }
__device__ __forceinline__ void spin_pause() { __builtin_amdgcn_s_sleep(2); }
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
}  // namespace skp
#endif

struct SkrsArgs {
  float* ws;      // [grid][2 slots][16 units][NT threads] float4: slot A (first segment) and slot B (last segment) of every workgroup
  int* flags;     // [grid][2]: readers still to come (0 = free)
  int* err;       // set to 1 if a spin timed out
  int tiles_n;    // channel tiles per row of tiles
  int tiles;      // total tiles
  int kt;         // k-tiles per tile
};

constexpr int SKRS_UNITS = 16;  // float4 units per lane of a 64x64 wave tile (2 x 2 MFMA tiles x 4 register quads)
constexpr int SKRS_MAXC = 9;    // most contributors a tile can have: the launcher keeps every share >= KT / 8 iterations

template <typename T, int NSPLIT, typename Epi, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_skrs_kernel(GemmCore g, Epi epi, SkrsArgs sk) {
  constexpr int TM = 2, TN = 2;
  constexpr int NT = 64 * WGM * WGN;
  constexpr int BM = 32 * WGM * TM, BN = 32 * WGN * TN;
  constexpr int NPL = (NSPLIT == 3) ? 2 : 1;
  constexpr int CPR = GEMM_KTB / 16;
  constexpr int KSTEPS = NPL == 2 ? 2 : 4;
  constexpr int CA = BM * CPR / NT, CW = BN * CPR / NT;
  constexpr int LPT = CA + CW;
  constexpr int TILE_A = BM * GEMM_KTB, TILE_W = BN * GEMM_KTB;
  constexpr int STAGE = TILE_A + TILE_W;
  constexpr int SLOT = SKRS_UNITS * NT * 4;  // floats per slot
  static_assert(CA >= 1 && CW >= 1 && CA * NT == BM * CPR && CW * NT == BN * CPR, "tile does not split evenly over the threads");
  static_assert(TM * TN * 4 == SKRS_UNITS, "16 units per lane");
  F5_DYN_LDS(char, smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = skp::uniform(tid >> 6);
  const int wm = wave % WGM, wn = wave / WGM;

  // ---- this workgroup's share ------------------------------------------------------------------------------------------------------
  const int bid = blockIdx.x, xcd = bid & 7, wi = bid >> 3, gx = gridDim.x >> 3;  // grid is a multiple of 8
  const int tlo = (int)((int64_t)xcd * sk.tiles / 8), thi = (int)((int64_t)(xcd + 1) * sk.tiles / 8);
  const int KT = sk.kt;
  const int64_t ix = (int64_t)(thi - tlo) * KT;  // iterations of this XCD class
  auto share_begin = [&](int i) { return (int)(ix * i / gx); };
  const int itb = share_begin(wi), ite = share_begin(wi + 1);
  if (itb >= ite) return;

  const int kbytes = g.K * (int)sizeof(T) * NPL;
  const uint32_t a_bytes = (uint32_t)((int64_t)(g.a_rows - 1) * g.lda * (int64_t)sizeof(T) + kbytes);
  const uint32_t w_bytes = (uint32_t)((int64_t)(g.w_rows - 1) * g.ldw * (int64_t)sizeof(T) + kbytes);
  const BufRsrc Ar = make_rsrc(reinterpret_cast<const T*>(g.A), a_bytes);
  const BufRsrc Wr = make_rsrc(reinterpret_cast<const T*>(g.W), w_bytes);

  // ---- issue side: LDS-DMA of the next iteration of the share, tile decode only when the tile changes --------------------------------
  uint32_t a_off[CA], w_off[CW];
  int a_c[CA], w_c[CW];
#pragma unroll
  for (int i = 0; i < CA; ++i) { const int c = tid + i * NT, row = c / CPR; a_c[i] = ((c % CPR) ^ ((row >> 1) & 7)) * 16; }
#pragma unroll
  for (int i = 0; i < CW; ++i) { const int c = tid + i * NT, row = c / CPR; w_c[i] = ((c % CPR) ^ ((row >> 1) & 7)) * 16; }
  int is_it = itb, is_kt = itb % KT, is_tile = -1;
  auto issue_next = [&](int stage) {
    char* base = smem + stage * STAGE + wave * 1024;
    const bool live = is_it < ite;
    if (live && is_tile != is_it / KT) {
      is_tile = is_it / KT;
      const int t = tlo + is_tile, mt = t / sk.tiles_n, m0 = mt * BM, n0 = (t - mt * sk.tiles_n) * BN;
#pragma unroll
      for (int i = 0; i < CA; ++i) {
        const int row = (tid + i * NT) / CPR;
        a_off[i] = (m0 + row) < g.a_rows ? (uint32_t)((int64_t)(m0 + row) * g.lda * (int64_t)sizeof(T) + a_c[i]) : OOB_ROW;
      }
#pragma unroll
      for (int i = 0; i < CW; ++i) {
        const int row = (tid + i * NT) / CPR;
        w_off[i] = (n0 + row) < g.w_rows ? (uint32_t)((int64_t)(n0 + row) * g.ldw * (int64_t)sizeof(T) + w_c[i]) : OOB_ROW;
      }
    }
    const int kb = is_kt * GEMM_KTB;
#pragma unroll
    for (int i = 0; i < CA; ++i) skp::dma_b128(Ar, base + i * NT * 16, (live && (kb + a_c[i]) < kbytes) ? a_off[i] + (uint32_t)kb : OOB_OFF);
#pragma unroll
    for (int i = 0; i < CW; ++i) skp::dma_b128(Wr, base + TILE_A + i * NT * 16, (live && (kb + w_c[i]) < kbytes) ? w_off[i] + (uint32_t)kb : OOB_OFF);
    ++is_it;
    if (++is_kt == KT) is_kt = 0;
  };

  f32x16 acc[TM][TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
  };
  zero_acc();

  const int frow = (lane & 31) * GEMM_KTB, fswz = ((lane & 31) >> 1) & 7, fhi = lane >> 5;
  int foff[NPL][KSTEPS];
#pragma unroll
  for (int p = 0; p < NPL; ++p)
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) foff[p][ks] = frow + (((p * 4 + 2 * ks + fhi) ^ fswz) << 4);
  const uint32_t lds0 = skp::lds_base(smem);
  auto read_frags = [&](uint32_t sA, uint32_t sW, int ks, Frag (&fa)[NPL][TM], Frag (&fw)[NPL][TN]) {
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
#pragma unroll
      for (int j = 0; j < TM; ++j) fa[p][j].u = skp::lds_read_b128(sA + j * 32 * GEMM_KTB + foff[p][ks]);
#pragma unroll
      for (int i = 0; i < TN; ++i) fw[p][i].u = skp::lds_read_b128(sW + i * 32 * GEMM_KTB + foff[p][ks]);
    }
  };
  auto mma_step = [&](const Frag (&fa)[NPL][TM], const Frag (&fw)[NPL][TN]) {
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        Mma32<T>::mma(acc[j][i], fw[0][i], fa[0][j]);
        if constexpr (NPL == 2) {
          Mma32<T>::mma(acc[j][i], fw[0][i], fa[1][j]);
          Mma32<T>::mma(acc[j][i], fw[1][i], fa[0][j]);
        }
      }
  };
  auto compute = [&](int stage) {
    const uint32_t sA = lds0 + stage * STAGE + (wm * 32 * TM) * GEMM_KTB;
    const uint32_t sW = lds0 + stage * STAGE + TILE_A + (wn * 32 * TN) * GEMM_KTB;
    Frag fa0[NPL][TM], fw0[NPL][TN], fa1[NPL][TM], fw1[NPL][TN];
    read_frags(sA, sW, 0, fa0, fw0);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks += 2) {
      skp::lds_wait();
      read_frags(sA, sW, ks + 1, fa1, fw1);
      mma_step(fa0, fw0);
      skp::lds_wait();
      if (ks + 2 < KSTEPS) read_frags(sA, sW, ks + 2, fa0, fw0);
      mma_step(fa1, fw1);
    }
  };

  // ---- tiles, contributors, slots ----------------------------------------------------------------------------------------------------
  // contributors of local tile lt: the workgroups of the class whose shares intersect [lt * KT, (lt + 1) * KT); shares are contiguous and
  // non-empty here (itb < ite for every contributor by construction: an empty share intersects nothing)
  auto first_contrib = [&](int lt) {  // smallest i with share_begin(i + 1) > lt * KT
    const int64_t lo = (int64_t)lt * KT;
    int i = (int)(lo * gx / ix);      // share_begin(i) <= lo for this guess or one below; walk to the exact one
    while (i > 0 && share_begin(i) > lo) --i;
    while (share_begin(i + 1) <= lo) ++i;
    return i;
  };
  auto last_contrib = [&](int lt) {   // largest i with share_begin(i) < (lt + 1) * KT
    const int64_t hi = (int64_t)(lt + 1) * KT;
    int i = (int)((hi - 1) * gx / ix);
    while (i + 1 < gx && share_begin(i + 1) < hi) ++i;
    while (share_begin(i) >= hi) --i;
    return i;
  };
  // slot of contributor p for tile lt: B if p holds the tile's HEAD (its share begins at or before the tile's first iteration and the
  // tile is not the first segment of a share that starts exactly there with the whole tile...), A otherwise.  A contributor holds the
  // HEAD iff it is the first contributor; the head is the LAST segment of that workgroup's share unless the share starts exactly at
  // the tile's first iteration, in which case the head is also its FIRST segment — it is still published in slot B (slot A is only
  // for segments with k0 > 0), so "first contributor <=> slot B" holds without exception.
  auto slot_ptr = [&](int p, int which) { return sk.ws + ((int64_t)(xcd + 8 * p) * 2 + which) * SLOT; };
  auto flag_ptr = [&](int p, int which) { return sk.flags + (xcd + 8 * p) * 2 + which; };
  auto unit_off = [&](int u) { return (u * NT + tid) * 4; };  // float4 per thread, lane-linear per unit

  auto publish = [&](int which, int readers) {  // all 16 units of the current accumulators -> own slot; then the flag
    float* s = slot_ptr(wi, which);
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(s + unit_off((j * TN + i) * 4 + q)) =
              make_float4(acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3]);
    skp::wait_vmcnt<0>();  // stores acknowledged by the L2 (also drains the DMA ring: once or twice per workgroup)
    skp::wg_barrier();
    if (tid == 0 && readers > 0) skp::flag_store(flag_ptr(wi, which), readers);
  };

  // finish the units this workgroup owns of local tile lt; `which` = the slot its own partial went to (B for a HEAD, A otherwise).
  // c contributors i0 .. i0 + c - 1, this workgroup is number r and owns the units r, r + c, r + 2c, ...  Everything — its own partial
  // included — is read back from the slots (the HEAD's registers are not used: one rolled loop, one inlined epilogue, few registers).
  auto finish_owned = [&](int lt, int which) {
    const int i0 = first_contrib(lt), c = last_contrib(lt) - i0 + 1, r = wi - i0;
    if (c == 1) return;  // (not reached: a tile with one contributor is FULL)
    if (c > SKRS_MAXC) {  // the launcher rules this out (shares >= KT / 8); never silently wrong
      if (tid == 0) skp::err_set(sk.err);
      return;
    }
    // wait for every partner's slot (one spinning thread, then the workgroup barrier)
    if (tid == 0) {
      for (int p = 0; p < c; ++p) {
        if (p == r) continue;
        int spins = 0;
        while (skp::flag_load(flag_ptr(i0 + p, p == 0 ? 1 : 0)) == 0) {
          skp::spin_pause();
          if (++spins > (1 << 22)) { skp::err_set(sk.err); break; }
        }
      }
    }
    skp::wg_barrier();
    const int t = tlo + lt, mt = t / sk.tiles_n, m0 = mt * BM, n0 = (t - mt * sk.tiles_n) * BN;
    const float* src[SKRS_MAXC];
#pragma unroll
    for (int p = 0; p < SKRS_MAXC; ++p) src[p] = p < c ? slot_ptr(i0 + p, p == r ? which : (p == 0 ? 1 : 0)) : nullptr;
#pragma unroll 1
    for (int u = r; u < SKRS_UNITS; u += c) {
      // all c partial sums of the unit in flight together, then the sum in contributor order: the same value whoever finishes the unit
      skp::f32x4v v[SKRS_MAXC];
#pragma unroll
      for (int p = 0; p < SKRS_MAXC; ++p)
        if (p < c) v[p] = skp::load_sc1_async(src[p] + unit_off(u));
      skp::f32x4v s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < SKRS_MAXC; ++p)
        if (p < c) {
          skp::settle(v[p]);
          s4 += v[p];
        }
      const int j = u >> 3, i = (u >> 2) & 1, q = u & 3;  // u = (j * TN + i) * 4 + q
      const int m = m0 + wm * 32 * TM + j * 32 + (lane & 31);
      const int n = n0 + wn * 32 * TN + i * 32 + 8 * q + 4 * (lane >> 5);
      if (m < g.M && n < g.N) epi(m, n, make_float4(s4[0], s4[1], s4[2], s4[3]), 0);
    }
    skp::wg_barrier();  // every thread has read what it needs from the partners' slots
    if (tid == 0)
      for (int p = 0; p < c; ++p)
        if (p != r) skp::flag_dec(flag_ptr(i0 + p, p == 0 ? 1 : 0));
  };
  auto readers_of = [&](int lt, int r) {  // partners that will read this workgroup's slot for tile lt
    const int i0 = first_contrib(lt), c = last_contrib(lt) - i0 + 1, owners = min(c, SKRS_UNITS);
    return owners - (r < owners ? 1 : 0);
  };

  auto plain_epilogue = [&](int lt) {
    const int t = tlo + lt, mt = t / sk.tiles_n, m0 = mt * BM, n0 = (t - mt * sk.tiles_n) * BN;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = m0 + wm * 32 * TM + j * 32 + (lane & 31);
      if (m >= g.M) continue;
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * 32 * TN + i * 32 + 8 * q + 4 * (lane >> 5);
          if (n < g.N) epi(m, n, make_float4(acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3]), 0);
        }
    }
  };

  // ---- the flat pipelined loop over the share -----------------------------------------------------------------------------------------
  issue_next(0);
  issue_next(1);
  skp::wait_vmcnt<LPT>();
  skp::wg_barrier();
  int st = 0;
  int c_tile = itb / KT, c_kt = itb - c_tile * KT;
  int job_tile[2] = {-1, -1};  // tiles to finish at the end of the share: [0] the HEAD (last segment), [1] the TAIL / MIDDLE (first segment)
  for (int it = itb; it < ite;) {
    const int seg_k0 = c_kt, seg_n = min(KT - c_kt, ite - it);
#pragma unroll 1
    for (int n = 0; n < seg_n; ++n) {
      const int st2 = st == 0 ? 2 : st - 1;
      issue_next(st2);
      compute(st);
      skp::wait_vmcnt<LPT>();
      skp::wg_barrier();
      st = st == 2 ? 0 : st + 1;
    }
    it += seg_n;
    c_kt += seg_n;
    const bool head_side = seg_k0 == 0, tail_side = c_kt == KT;
    if (head_side && tail_side) {
      plain_epilogue(c_tile);
    } else if (!head_side) {  // TAIL or MIDDLE: always the first segment of the share
      const int i0 = first_contrib(c_tile);
      publish(0, readers_of(c_tile, wi - i0));
      job_tile[1] = c_tile;
    } else {                  // HEAD: always the last segment of the share
      publish(1, readers_of(c_tile, 0));
      job_tile[0] = c_tile;
    }
    zero_acc();
    if (c_kt == KT) { ++c_tile; c_kt = 0; }
  }
  skp::wait_vmcnt<0>();
  // end of the share: first the HEAD tile (its partners published long ago), then the tile of the first segment
#pragma unroll 1
  for (int job = 0; job < 2; ++job)
    if (job_tile[job] >= 0) finish_owned(job_tile[job], job == 0 ? 1 : 0);
}
