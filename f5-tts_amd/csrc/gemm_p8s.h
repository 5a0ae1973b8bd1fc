// gemm_p8s.h — the block GEMM of the ONE-ROUND launches (a single utterance: 2812 rows = 240 workgroups on 256 CUs), as a ping-pong of
// the two k-step-split wave groups of gemm_pp.h's tiles 68-70:  epilogue( A[M,K] . W[N,K]^T ), MX lines (fp16m) or plain fp16 rows.
//
// gemm_pp.h's k-step-split tiles put two waves on every SIMD (group g of 2 x 2 waves multiplies the 16-wide k-steps g, g + 2, .. of every
// k-tile of the SAME 64 TM x 64 TN output tile; the partial sums meet in the epilogue), but run them in lockstep: behind the one barrier
// per k-tile all eight waves issue their LDS-DMA pieces and fragment reads together, then all multiply — the matrix pipe idles while both
// waves of a SIMD load (q|k|v at B = 1: 40 us where its MFMAs need 15 and its L2 -> LDS stream 13).  Here the groups run HALF A PHASE apart, as
// in gemm_p8.h: a phase = one k-tile of one group = a MEMORY half (fragment reads of that k-tile: TM + TN reads, + 2 (TM + TN) reads of
// the lane's MX words on the k-tiles whose fp6 correction is this group's turn) and a MATRIX half (TM TN MFMAs, + TM TN fp6 MFMAs on its
// turn, under s_setprio 1), each closed by an s_barrier; group 1 starts one barrier late, so between two barriers one wave of every SIMD
// multiplies while the other one reads.  Fragments live in ONE register buffer.
//   * MX lines: group g multiplies hi k-step g of every line; the line's correction MFMA belongs to group 1 on even k-tiles and to group 0
//     on odd ones (as in gemm_pp.h) — a group's phases alternate light (TM TN MFMAs, TM + TN reads) and heavy (twice the MFMAs, three times
//     the reads), and the stagger pairs a heavy memory half with a heavy matrix half of the other group.
//   * plain fp16 rows: group g multiplies k-steps g and g + 2 of the 64-k line: every phase alike.
//   * the ring: NS stages of one k-tile (BM + BN rows x 128 B), filled by all eight waves (wave w brings the 8-row pieces w, w + 8, ..:
//     TM of the A rows, TN of the W rows — they are 64 rows apart, so ONE per-lane offset per operand serves them and the distance rides in
//     the scalar offset).  k-tile p + NS - 1 goes into the stage k-tile p - 1 leaves, requested during phase p in the interval where group 0
//     multiplies and group 1 reads (group 0: between its MFMAs; group 1: ahead of its reads) — the barrier before that interval is the first
//     one behind which BOTH groups' reads of k-tile p - 1 have been waited for.  The counted wait for k-tile p + 1 (`vmcnt(pieces of one
//     k-tile)`: the tile requested last stays in flight) ends that same interval, one barrier before the first read.
//   * the tile's rows start at the buffer descriptors' base: rows past the operand's end are out of range (zeros), no per-lane row test.
// Epilogue: the partial-sum exchange of gemm_pp.h's split tiles (every wave parks the accumulator tiles it does not finish in the idle ring and
// adds its partner's copy of those it does finish, group 0 + group 1 in that order whoever finishes: deterministic).
#pragma once
#include "gemm_p8.h"

template <int TM, int TN, int NS>
constexpr int gemm_p8s_lds_bytes() {
  return NS * 64 * (TM + TN) * GEMM_KTB;
}

// ABL (microbenchmark ablations): bit 0 = no epilogue, bit 2 = no LDS-DMA after the prologue, bit 3 = no MFMAs
template <int NSPLIT, int TM, int TN, int NS, typename Epi, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_p8s_kernel(GemmCore g, Epi epi) {
  using namespace p8;
  static_assert(NSPLIT == 1 || NSPLIT == 2, "plain fp16 rows or MX lines");
  constexpr bool MX = NSPLIT == 2;
  constexpr int NPL = MX ? 2 : 1;
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int STAGE = (BM + BN) * GEMM_KTB, TILE_A = BM * GEMM_KTB;
  constexpr int PPW = TM + TN;  // LDS-DMA pieces (8 rows x 128 B) per wave per k-tile
  static_assert(NS >= 3 && NS * STAGE <= 160 * 1024 && (NS - 1) * PPW <= 63, "ring shape");
  static_assert(4 * TM * TN * 4096 <= NS * STAGE, "the partial-sum exchange reuses the ring");
  F5_DYN_LDS(char, smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = pp::uniform(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3, wm = w4 & 1, wn = w4 >> 1;
  int m0, n0;
  {  // tile order as gemm_pp_kernel: XCD-contiguous runs, channel tiles fastest, optional groups of row tiles
    const int nt = (g.N + BN - 1) / BN, nwg = gridDim.x;
    const int bid = blockIdx.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    int mt, ntile;
    if (g.group_m > 1) {
      const int mtt = (g.M + BM - 1) / BM, per = g.group_m * nt;
      const int gi = L / per, first = gi * g.group_m, gsz = min(g.group_m, mtt - first), within = L - gi * per;
      ntile = within / gsz;
      mt = first + (within - ntile * gsz);
    } else {
      mt = L / nt;
      ntile = L - mt * nt;
    }
    m0 = mt * BM;
    n0 = ntile * BN;
  }
  const int kbytes = g.K * 2 * NPL;   // bytes of one operand row
  const int nkt = kbytes / GEMM_KTB;  // k-tiles: even, >= NS + 1 (launcher)
  const int arows = g.a_rows - m0, wrows = g.w_rows - n0;
  const BufRsrc Ar = make_rsrc(reinterpret_cast<const char*>(g.A) + (int64_t)m0 * g.lda * 2, arows > 0 ? (uint32_t)((int64_t)(arows - 1) * g.lda * 2 + kbytes) : 0u);
  const BufRsrc Wr = make_rsrc(reinterpret_cast<const char*>(g.W) + (int64_t)n0 * g.ldw * 2, wrows > 0 ? (uint32_t)((int64_t)(wrows - 1) * g.ldw * 2 + kbytes) : 0u);

  // LDS-DMA: piece P of an operand = rows 8P .. 8P+7 -> bytes [1024 P, +1024) of its part of the stage; lane l brings row 8P + l/8, logical
  // chunk (l%8) ^ swz(row).  Wave w brings pieces w + 8 i: 64 rows apart, the same swz — one lane offset per operand.
  uint32_t qa, qw;
  {
    const int row = 8 * wave + (lane >> 3), lc = (lane & 7) ^ ((row >> 1) & 7);
    qa = (uint32_t)((int64_t)row * g.lda * 2 + lc * 16);
    qw = (uint32_t)((int64_t)row * g.ldw * 2 + lc * 16);
  }
  const uint32_t a64 = (uint32_t)(64 * g.lda * 2), w64 = (uint32_t)(64 * g.ldw * 2);
  char* const my_dst = smem + wave * 1024;
  // piece I of this wave's share of k-tile kt -> stage `stage`
  auto issue_piece = [&](auto IC_, int kt, int stage) {
    constexpr int i = decltype(IC_)::value;
    if constexpr (i < TM) pp::dma_b128(Ar, my_dst + stage * STAGE + i * 8192, qa, (uint32_t)kt * GEMM_KTB + (uint32_t)i * a64);
    else pp::dma_b128(Wr, my_dst + stage * STAGE + TILE_A + (i - TM) * 8192, qw, (uint32_t)kt * GEMM_KTB + (uint32_t)(i - TM) * w64);
  };
  auto issue_all = [&](int kt, int stage) { static_for<PPW>([&](auto I) { issue_piece(I, kt, stage); }); };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int j = 0; j < TM; ++j)
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  // fragment addressing: lane (row = lane & 31, half = lane >> 5) reads 16-byte chunks of its row's line: chunk c at ((c ^ swz) << 4)
  const uint32_t lds0 = pp::lds_base(smem);
  const int fr = lane & 31, fswz = (fr >> 1) & 7, fh = lane >> 5;
  const uint32_t rowa = lds0 + (uint32_t)(fr * GEMM_KTB) + (uint32_t)(wm * 32 * TM) * GEMM_KTB;
  const uint32_t roww = lds0 + (uint32_t)(fr * GEMM_KTB) + TILE_A + (uint32_t)(wn * 32 * TN) * GEMM_KTB;
  auto chunk_off = [&](int c) { return (uint32_t)((c ^ fswz) << 4); };

  // the k-loop of group G (compile time: whose turn a k-tile's fp6 correction is depends on the group)
  auto k_loop = [&](auto GC) {
    constexpr int G = decltype(GC)::value;
    constexpr int NX = MX ? 3 : 2;  // 16-byte reads per tile and k-tile: MX: hi k-step G | the lane's two MX words; fp16: k-steps G, G + 2
    uint32_t coff[NX];
#pragma unroll
    for (int x = 0; x < NX; ++x) coff[x] = chunk_off(MX ? (x == 0 ? 2 * G + fh : 4 + 2 * fh + (x - 1)) : 2 * (G + 2 * x) + fh);
    Frag fa[NX][TM], fw[NX][TN];

    // the counted wait of phase p, behind its request: k-tiles p + 2 .. p + NS - 1 may stay in flight (those that exist)
    auto land_next = [&](int p, bool req) {
      if (p + 1 >= nkt) return;
      if (req) pp::wait_vmcnt<(NS - 2) * PPW>();
      else if (NS >= 4 && p + 2 < nkt) pp::wait_vmcnt<(NS >= 4 ? PPW : 0)>();  // (NS = 4: one k-tile behind p + 1 is still to come)
      else pp::wait_vmcnt<0>();
    };
    // one phase = k-tile p of this group; PAR = p & 1.  MX: HEAVY (the fp6 correction is this group's) on the k-tiles of the other parity
    auto phase = [&](auto PARC, int p, uint32_t soff, int stage_req, bool closing) {
      constexpr int PAR = decltype(PARC)::value;
      constexpr bool HEAVY = MX && PAR != G;
      const bool req = (!(ABL & 4) || p == 0) && p + NS - 1 < nkt;  // k-tile p + NS - 1 into the stage k-tile p - 1 left (phase 0: the one stage the prologue did not fill); wave-uniform
      // ---- memory half
      if constexpr (G == 1) {
        if (req) issue_all(p + NS - 1, stage_req);
      }
      static_for<NX>([&](auto X) {
        constexpr int x = decltype(X)::value;
        if constexpr (!MX || x == 0 || HEAVY) {
          const uint32_t aa = rowa + soff + coff[x], ww = roww + soff + coff[x];
          static_for<TM>([&](auto J) { fa[x][decltype(J)::value].u = pp::lds_read_b128<decltype(J)::value * 4096>(aa); });
          static_for<TN>([&](auto I) { fw[x][decltype(I)::value].u = pp::lds_read_b128<decltype(I)::value * 4096>(ww); });
        }
      });
      if constexpr (G == 1) land_next(p, req);  // k-tile p + 1 (this wave's pieces) has landed; the k-tiles requested since stay in flight
      pp::wg_barrier();
      // ---- matrix half
      pp::lds_wait();
      prio<1>();
      static_for<NX>([&](auto X) {
        constexpr int x = decltype(X)::value;
        if constexpr (!MX || x == 0 || HEAVY) {
          static_for<TM>([&](auto J) { pin_after_wait(fa[x][decltype(J)::value].u); });
          static_for<TN>([&](auto I) { pin_after_wait(fw[x][decltype(I)::value].u); });
        }
      });
      // group 0 requests its pieces between its MFMAs: one piece behind every MFMA row until they are out
      auto mma_rows = [&](auto F) {  // F(j, i): the MFMAs of accumulator tile (j, i)
        static_for<TM>([&](auto J) {
          constexpr int j = decltype(J)::value;
#pragma unroll
          for (int i = 0; i < TN; ++i) F(j, i);
          if constexpr (G == 0 && j < PPW) {
            pp::pin();
            if (req) issue_piece(IC<j>{}, p + NS - 1, stage_req);
            pp::pin();
          }
        });
      };
      if constexpr (ABL & 8) {
#ifndef F5_HIPEMU
        static_for<NX>([&](auto X) {
          constexpr int x = decltype(X)::value;
          if constexpr (!MX || x == 0 || HEAVY) {
            const uint32_t k0 = fa[x][0].u.x, k1 = fa[x][TM - 1].u.w, k2 = fw[x][0].u.x, k3 = fw[x][TN - 1].u.w;
            asm volatile("" ::"v"(k0), "v"(k1), "v"(k2), "v"(k3));
          }
        });
#endif
        if constexpr (G == 0) {
          if (req) issue_all(p + NS - 1, stage_req);
        }
      } else {
        mma_rows([&](int j, int i) { Mma32<f16>::mma(acc[j][i], fw[0][i], fa[0][j]); });
        if constexpr (G == 0 && TM < PPW) {  // the pieces the first MFMA rows did not carry
          pp::pin();
          static_for<PPW - TM>([&](auto I) { if (req) issue_piece(IC<TM + decltype(I)::value>{}, p + NS - 1, stage_req); });
          pp::pin();
        }
        if constexpr (!MX) {
#pragma unroll
          for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int i = 0; i < TN; ++i) Mma32<f16>::mma(acc[j][i], fw[1][i], fa[1][j]);
        } else if constexpr (HEAVY) {
#pragma unroll
          for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int i = 0; i < TN; ++i) mx_mma(acc[j][i], fw[1][i].u, fw[2][i].u, fa[1][j].u, fa[2][j].u);  // both correction terms of the line
        }
      }
      prio<0>();
      pp::pin();
      if constexpr (G == 0) land_next(p, req);  // k-tile p + 1 has landed (this wave's pieces); the k-tiles requested since stay in flight
      if (closing) pp::wg_barrier();
    };

    // prologue: k-tiles 0 .. NS-2 requested (phase 0 requests k-tile NS-1 into the one stage left, phase p >= 1 k-tile p + NS - 1 into the
    // stage k-tile p - 1 leaves), k-tile 0 landed and visible
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue_all(s, s);
    pp::wait_vmcnt<(NS - 2) * PPW>();
    pp::wg_barrier();
    if constexpr (G == 1) pp::wg_barrier();  // half a phase behind group 0 from here on
    uint32_t soff = 0;
    int stage_req = NS - 1;  // the stage phase p requests into: (p + NS - 1) % NS
    auto next = [&](uint32_t so) { return so + STAGE == (uint32_t)(NS * STAGE) ? 0u : so + STAGE; };
    for (int p = 0; p < nkt; p += 2) {
      phase(IC<0>{}, p, soff, stage_req, true);
      soff = next(soff);
      stage_req = stage_req == NS - 1 ? 0 : stage_req + 1;
      phase(IC<1>{}, p + 1, soff, stage_req, G == 0 || p + 2 < nkt);  // group 1 started one barrier late: it skips the last one
      soff = next(soff);
      stage_req = stage_req == NS - 1 ? 0 : stage_req + 1;
    }
  };
  if (grp == 0) k_loop(IC<0>{});
  else k_loop(IC<1>{});

  if constexpr (ABL & 1) {
#ifndef F5_HIPEMU
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int r = 0; r < 16; r += 4) asm volatile("" ::"v"(acc[j][i][r]), "v"(acc[j][i][r + 1]), "v"(acc[j][i][r + 2]), "v"(acc[j][i][r + 3]));
#endif
  } else {
    // The two groups hold partial sums of the same tiles (gemm_pp.h, the split tiles' epilogue).  Tile t = j * TN + i is FINISHED by group
    // (t < NT0 ? 0 : 1): every wave parks the tiles it does not finish in the (now idle) ring — [wave][tile][quad][lane] float4 — and adds its
    // partner's copy of the tiles it does finish, always group 0 + group 1 in that order.
    constexpr int NT = TM * TN, NT0 = (NT + 1) / 2;
    pp::wg_barrier();  // every wave has read its last fragments: the ring is free
    float4* xch = reinterpret_cast<float4*>(smem);
    static_for<NT>([&](auto TI) {
      constexpr int ti = decltype(TI)::value, j = ti / TN, i = ti % TN;
      if (grp != (ti < NT0 ? 0 : 1)) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          xch[((w4 * NT + ti) * 4 + q) * 64 + lane] = make_float4(acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3]);
      }
    });
    __syncthreads();
    const uint32_t xbase = lds0 + (uint32_t)((w4 * NT * 4) * 64 + lane) * 16u;
    static_for<NT>([&](auto TI) {
      constexpr int ti = decltype(TI)::value, j = ti / TN, i = ti % TN;
      if (grp == (ti < NT0 ? 0 : 1)) {
        f32x16 one[1][1];
        uint4 ou[4];
        static_for<4>([&](auto Q) { ou[decltype(Q)::value] = pp::lds_read_b128<(ti * 4 + decltype(Q)::value) * 1024>(xbase); });
        pp::lds_wait();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          union { uint4 u; float4 f; } cv;
          cv.u = ou[q];
          const float4 o = cv.f;
          if (grp == 0) {  // own (group 0) + partner (group 1)
            one[0][0][4 * q] = acc[j][i][4 * q] + o.x; one[0][0][4 * q + 1] = acc[j][i][4 * q + 1] + o.y;
            one[0][0][4 * q + 2] = acc[j][i][4 * q + 2] + o.z; one[0][0][4 * q + 3] = acc[j][i][4 * q + 3] + o.w;
          } else {         // partner (group 0) + own
            one[0][0][4 * q] = o.x + acc[j][i][4 * q]; one[0][0][4 * q + 1] = o.y + acc[j][i][4 * q + 1];
            one[0][0][4 * q + 2] = o.z + acc[j][i][4 * q + 2]; one[0][0][4 * q + 3] = o.w + acc[j][i][4 * q + 3];
          }
        }
        pp_unscale<1, 1>(one, g, n0 + wn * 32 * TN + 32 * i, lane);
        epi.template tile<1, 1>(one, m0 + wm * 32 * TM + 32 * j, n0 + wn * 32 * TN + 32 * i, lane);
      }
    });
  }
}
