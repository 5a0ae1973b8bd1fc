// gemm_sk.h — stream-K schedule of the direct-to-LDS GEMM for SMALL grids (B = 1: M ~ 1.4-2.8k rows).
//
// Why: at B = 1 the best workgroup shape (8 waves, 64x64 per wave, 256x128 or 128x256 tile, 3-stage LDS-DMA ring: 74 % MFMA-pipe
// utilisation for a lone workgroup, 0.9 KB of LDS traffic per MFMA against 1.5 KB for the 4-wave 128x64 tile) yields only 88-264
// tiles for 256 CUs, and the 128x64 tiling that does fill the chip runs its k-loop at ~45 % (LDS-bound) and leaves CUs with 1 or 2
// or 3 tiles each.  Stream-K removes the quantisation: the (tile, k-tile) iteration space is cut into G equal contiguous shares, one
// per resident workgroup, so every CU does the same number of k-tile iterations with the efficient tile shape.
//
// Schedule.  Workgroup b belongs to XCD class x = b & 7 (observed dispatch: consecutive workgroups round-robin over the 8 XCDs) and
// is the i = b >> 3 -th of its class.  The tiles are cut into 8 contiguous runs (channel tiles fastest), one per class, so the
// workgroups that share operand panels share an L2; inside a class the run's iterations [0, n_x * KT) are cut into G/8 equal
// shares.  A share is a sequence of segments (tile, [k0, k1)):
//   * k0 > 0: a TAIL of a tile — only possible for the FIRST segment of a share, so it is computed at the very start of the
//     workgroup's life: the accumulators go to this workgroup's 128 KB workspace slot, then flag[b] = 1.
//   * k0 == 0: this workgroup owns the HEAD of the tile and FINISHES it: if its share ends inside the tile (k1 < KT) it adds the
//     partial sums of the workgroups i+1, i+2, ... of the same class that cover [k1, KT) in a fixed order (deterministic), then
//     runs the epilogue.
// The finisher spins on flag[b'], reads the slot, resets the flag to 0 (one consumer per flag, so the buffers are clean for the
// next launch and a captured graph can replay the node).  Producers never wait and publish early, so a finisher's wait is short;
// a finisher waits on HIGHER-numbered workgroups, which are resident when the grid fits the chip (the launcher sizes it so) and
// otherwise get a CU as soon as any producer-only or satisfied workgroup retires.  The spin is bounded (err word) so that a logic
// error cannot hang the GPU.
//
// The LDS-DMA ring runs straight through segment boundaries: the loads of the next segment's first k-tiles are in flight while
// the epilogue / fix-up of the current one runs.
#pragma once
#include "gemm.h"

struct SkArgs {
  float* ws;       // [grid][8 waves * TM*TN*16 * 64 lanes] fp32 partial accumulators (128 KB per workgroup for 64x64 wave tiles)
  int* flags;      // [grid] 0 = empty, 1 = partial ready
  int* err;        // set to 1 if a spin timed out
  int tiles_n;     // channel tiles per row of tiles
  int tiles;       // total tiles
  int kt;          // k-tiles per tile
  long long* dbg;  // optional [grid][8] wall-clock stamps (100 MHz) of the phases of each workgroup (microbenchmark only)
};

template <typename T, int NSPLIT, int TM, int TN, typename Epi, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_sk_kernel(GemmCore g, Epi epi, SkArgs sk) {
  constexpr int NT = 64 * WGM * WGN;
  constexpr int BM = 32 * WGM * TM, BN = 32 * WGN * TN;
  constexpr int NPL = (NSPLIT == 3) ? 2 : 1;
  constexpr int CPR = GEMM_KTB / 16;
  constexpr int KSTEPS = NPL == 2 ? 2 : 4;
  constexpr int CA = BM * CPR / NT, CW = BN * CPR / NT;
  constexpr int LPT = CA + CW;
  constexpr int TILE_A = BM * GEMM_KTB, TILE_W = BN * GEMM_KTB;
  constexpr int STAGE = TILE_A + TILE_W;
  constexpr int SLOT = NT * TM * TN * 16;  // floats per workspace slot
  static_assert(CA >= 1 && CW >= 1 && CA * NT == BM * CPR && CW * NT == BN * CPR, "tile does not split evenly over the threads");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WGM, wn = wave / WGM;

  // ---- this workgroup's share of the iteration space -----------------------------------------------------------------------------
  const int bid = blockIdx.x, xcd = bid & 7, wi = bid >> 3, gx = gridDim.x >> 3;  // grid is a multiple of 8
  const int tlo = (int)((int64_t)xcd * sk.tiles / 8), thi = (int)((int64_t)(xcd + 1) * sk.tiles / 8);
  const int KT = sk.kt;
  const int64_t ix = (int64_t)(thi - tlo) * KT;  // iterations of this XCD class
  auto share_begin = [&](int i) { return (int)(ix * i / gx); };
  const int itb = share_begin(wi), ite = share_begin(wi + 1);
  if (itb >= ite) return;
  int dbg_n = 0;
  auto stamp = [&]() { if (sk.dbg && tid == 0 && dbg_n < 8) sk.dbg[(int64_t)bid * 8 + dbg_n++] = (long long)wall_clock64(); };
  stamp();

  const int kbytes = g.K * (int)sizeof(T) * NPL;
  const uint32_t a_bytes = (uint32_t)((int64_t)(g.a_rows - 1) * g.lda * (int64_t)sizeof(T) + kbytes);
  const uint32_t w_bytes = (uint32_t)((int64_t)(g.w_rows - 1) * g.ldw * (int64_t)sizeof(T) + kbytes);
  const BufRsrc Ar = make_rsrc(reinterpret_cast<const T*>(g.A), a_bytes);
  const BufRsrc Wr = make_rsrc(reinterpret_cast<const T*>(g.W), w_bytes);

  // ---- issue side: LDS-DMA of iteration `it` (flat index inside the class), tile decode only when the tile changes ----------------
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
    for (int i = 0; i < CA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(Ar, (__attribute__((address_space(3))) void*)(base + i * NT * 16), 16,
                                               (int)((live && (kb + a_c[i]) < kbytes) ? a_off[i] + (uint32_t)kb : OOB_OFF), 0, 0, 0);
#pragma unroll
    for (int i = 0; i < CW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(Wr, (__attribute__((address_space(3))) void*)(base + TILE_A + i * NT * 16), 16,
                                               (int)((live && (kb + w_c[i]) < kbytes) ? w_off[i] + (uint32_t)kb : OOB_OFF), 0, 0, 0);
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
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  auto lds_read = [&](uint32_t addr) -> uint4 {  // inline asm: see gemm_glds_kernel
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return make_uint4(v[0], v[1], v[2], v[3]);
  };
  auto read_frags = [&](uint32_t sA, uint32_t sW, int ks, Frag (&fa)[NPL][TM], Frag (&fw)[NPL][TN]) {
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
#pragma unroll
      for (int j = 0; j < TM; ++j) fa[p][j].u = lds_read(sA + j * 32 * GEMM_KTB + foff[p][ks]);
#pragma unroll
      for (int i = 0; i < TN; ++i) fw[p][i].u = lds_read(sW + i * 32 * GEMM_KTB + foff[p][ks]);
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
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      read_frags(sA, sW, ks + 1, fa1, fw1);
      mma_step(fa0, fw0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 2 < KSTEPS) read_frags(sA, sW, ks + 2, fa0, fw0);
      mma_step(fa1, fw1);
    }
  };

  // ---- segment end: partial to the workspace, or fix-up + epilogue -----------------------------------------------------------------
  // Producer and finisher of a tile are in the same XCD class, i.e. behind the same L2 (asserted by the bit-exact microbenchmark /
  // parity tests: a wrong class assumption shows up as a spin time-out or stale sums).  So the hand-over needs no cache write-back
  // (an agent-scope release is a whole-L2 write-back per wave on a multi-XCD part — measured 10x the kernel's run time — and
  // system-scope accesses go uncached to HBM dword by dword, 40 us per slot): plain stores reach the L2 (the vector L1 is write-
  // through) and are acknowledged (vmcnt) before the flag store; the finisher reads flag and slot with agent-scope loads (sc1),
  // which miss the per-CU L1 and hit that L2.
  float* my_slot = sk.ws + (int64_t)bid * SLOT;
  auto slot_index = [&](int j, int i, int q) { return (((j * TN + i) * 4 + q) * NT + tid) * 4; };  // float4 per thread, lane-linear
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  auto finish_segment = [&](int ltile, int k0, int k1, int free_stage) {
    if (k0 > 0) {  // a tail of the tile (always the FIRST segment of a share, computed right at the start of this workgroup's life):
                   // hand the partial sums to the workgroup that owns the head of the tile
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(my_slot + slot_index(j, i, q)) =
                make_float4(acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3]);
      wait_vmcnt<0>();  // stores acknowledged by the L2 (drains the DMA ring once per workgroup)
      wg_barrier();
      if (tid == 0) __hip_atomic_store(sk.flags + bid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      stamp();
      return;
    }
    if (k1 < KT) {  // this workgroup owns the head [0, k1) of the tile and its share ends here: the rest was computed by the next
                    // workgroups of the class as the first thing they did, so their partial sums are (about to be) there
      const int tile_end = (ltile + 1) * KT;
      for (int p = wi + 1; p < gx; ++p) {
        const int pbeg = share_begin(p), pend = share_begin(p + 1);
        if (pbeg >= tile_end) break;
        if (pbeg == pend) continue;  // empty share: that workgroup exited without writing anything
        const int pb = xcd + 8 * p;
        if (tid == 0) {
          int spins = 0;
          while (__hip_atomic_load(sk.flags + pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 22)) { atomicOr(sk.err, 1); break; }
          }
        }
        wg_barrier();
        stamp();
        const float* src = sk.ws + (int64_t)pb * SLOT;
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int i = 0; i < TN; ++i) {
            f32x4v v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[q]) : "v"(src + slot_index(j, i, q)) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[j][i][4 * q + e] += v[q][e];
          }
        wg_barrier();  // every thread has its values before the slot is released
        if (tid == 0) __hip_atomic_store(sk.flags + pb, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pend >= tile_end) break;
      }
    }
    // Epilogue through LDS.  In the accumulator layout a lane owns 4 channels of ONE row, so a store instruction touches 32 rows with
    // 16 bytes each; the L2 then sees 8 partial writes per 128-byte line and the epilogue of a 256x128 tile takes ~15 us (measured,
    // as long as 14 k-tiles of the main loop) — exposed, since this kernel runs one workgroup per CU.  Each wave therefore transposes
    // one 32x32 accumulator tile at a time through a private 4 KB region of the ring stage that was consumed last (free until the
    // next issue): rows become contiguous over 8 lanes, every store / residual load covers whole 128-byte (fp32) or 64-byte (fp16
    // plane) row segments.  Inline-asm LDS ops for the same reason as in the main loop (no compiler-inserted vmcnt(0) drain).
    const int t = tlo + ltile, mt = t / sk.tiles_n, m0 = mt * BM, n0 = (t - mt * sk.tiles_n) * BN;
    const uint32_t stg = lds0 + (uint32_t)free_stage * STAGE + (uint32_t)wave * 4096u;
    const int wrow = lane & 31, whi = lane >> 5, rrow = lane >> 3, rchunk = lane & 7;
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int i = 0; i < TN; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4v v = {acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3]};
          const uint32_t a = stg + wrow * 128 + (((2 * q + whi) ^ (wrow & 7)) << 4);
          asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(v) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        f32x4v r[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int row = ps * 8 + rrow;
          const uint32_t a = stg + row * 128 + ((rchunk ^ (row & 7)) << 4);
          asm volatile("ds_read_b128 %0, %1" : "=v"(r[ps]) : "v"(a) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const int n = n0 + wn * 32 * TN + i * 32 + rchunk * 4;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int m = m0 + wm * 32 * TM + j * 32 + ps * 8 + rrow;
          if (m < g.M && n < g.N) epi(m, n, make_float4(r[ps][0], r[ps][1], r[ps][2], r[ps][3]), 0);
        }
      }
    wg_barrier();  // the staging stage is about to be overwritten by the next LDS-DMA issue (any wave's pieces)
  };

  // ---- the flat pipelined loop -----------------------------------------------------------------------------------------------------
  issue_next(0);
  issue_next(1);
  wait_vmcnt<LPT>();
  wg_barrier();
  int st = 0;
  int c_tile = itb / KT, c_kt = itb - c_tile * KT;
  for (int it = itb; it < ite;) {
    // one segment: the k-tiles [c_kt, seg_k1) of tile c_tile.  The hot loop holds nothing of the epilogue / fix-up.
    const int seg_k0 = c_kt, seg_n = min(KT - c_kt, ite - it);
#pragma unroll 1
    for (int n = 0; n < seg_n; ++n) {
      const int st2 = st == 0 ? 2 : st - 1;
      issue_next(st2);  // iteration it + 2 (all out of range past the end of the share)
      compute(st);
      wait_vmcnt<LPT>();
      wg_barrier();
      st = st == 2 ? 0 : st + 1;
    }
    it += seg_n;
    c_kt += seg_n;
    stamp();
    finish_segment(c_tile, seg_k0, c_kt, st == 0 ? 2 : st - 1);  // the stage consumed last: nobody reads it any more
    stamp();  // the ring keeps running: the next segment's first two k-tiles are in flight
    zero_acc();
    if (c_kt == KT) { ++c_tile; c_kt = 0; }
  }
  wait_vmcnt<0>();
}
