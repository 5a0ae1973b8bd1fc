// gemm_p8.h — the block GEMM of the many-round launches (rows >= ~10k: batches of utterances):  epilogue( A[M,K] . W[N,K]^T ), plain fp16
// rows or MX lines (fp16m), the operand layouts, LDS image and wave-tile epilogues of gemm_pp.h.
//
// Why a second k-loop (round 5).  gemm_pp.h's 256x256 tile runs its eight waves in LOCKSTEP: one barrier per k-tile, behind which every wave
// issues its 8 LDS-DMA pieces and its next fragment reads at the same moment, then every wave multiplies.  At 90k rows that loop keeps the
// matrix pipe 59 % busy (PMC) and plain fp16 reaches 60 % of hipBLASLt's rate (profiles/r04n_*): the DMA issue (25 % of the launch by the
// no-DMA ablation) and the fragment-read latency of both waves of a SIMD fall into the same gaps.  Here the two waves of a SIMD run HALF A
// PHASE APART ("ping-pong", the 8-phase schedule of the CDNA GEMM playbook):
//   * 8 waves = 2 groups x 4; group g = wave >> 2 owns the rows 128 g .. 128 g + 127 of the tile, wave & 3 a 64-channel strip: the wave
//     tile is 128x64 (4 x 2 MFMA tiles of 32x32), as in gemm_pp.h's tile 50, and waves w and w + 4 share a SIMD;
//   * a k-tile (one 128-byte line per operand row) is multiplied in 4 PHASES, one per 64x32 quadrant of the wave tile; a phase is a MEMORY
//     half (2 LDS-DMA pieces issued, 4 or 8 ds_read_b128 fragment reads, the counted vmcnt wait for the NEXT phase's operands) and a MATRIX
//     half (8 MFMAs — MX lines: 4 fp16 + 2 fp6 — under s_setprio 1), each closed by an s_barrier.  Group 1 starts one barrier late, so
//     between any two barriers one wave of every SIMD multiplies while the other one loads: the matrix pipe sees back-to-back MFMAs, the
//     loader's issue time and LDS latency are covered by its partner, and fragments need ONE register buffer (64 VGPRs), not two;
//   * quadrant order (A01,w0) (A01,w1) (A23,w1) (A23,w0) on even k-tiles and (A01,w1) (A01,w0) (A23,w0) (A23,w1) on odd ones: the weight tile
//     of a k-tile's last quadrant is the one nobody reads in it, so the NEXT k-tile's first weight tile is read there — 8, 4, 8, 4 reads per
//     phase instead of 12, 4, 8, 0;
//   * LDS: two k-tile buffers (A buf 0 | A buf 1 | W buf 0 | W buf 1 = 128 KB), refilled by QUARTERS (the 128 rows one phase reads: a01 =
//     rows 0-63 of both groups' halves, a23 = rows 64-127, w0 / w1 = the first / second 32 rows of every strip; 16 pieces of 1 KB = 2 per
//     wave) exactly two phases after the phase that read them (every wave's reads of the region have been waited for by then: WAR-safe with
//     no extra wait), six phases before the phase that needs them: one quarter per phase, `s_waitcnt vmcnt(10)` (five younger quarters stay in
//     flight) in the memory half of the phase BEFORE the consumer, so the landing is separated from the first read by a barrier both groups
//     have passed.  The last pair of k-tiles is a second copy of the loop body with nothing left to issue and the counts 10, 8, .. 0.
// Same MFMA order per accumulator as gemm_pp.h / gemm.h (k ascending; MX: hi k-step 0, hi k-step 1, the fp6 correction), so results equal the
// other tiles' byte for byte (tests/test_pp_gemm_shim.py runs this kernel on the host shim against the generic kernel).
#pragma once
#include "gemm_pp.h"

namespace p8 {
#ifdef F5_HIPEMU
template <int P>
inline void prio() {}
#else
template <int P>
__device__ __forceinline__ void prio() { __builtin_amdgcn_s_setprio(P); }
#endif
template <int V>
using IC = std::integral_constant<int, V>;
}  // namespace p8

constexpr int P8_OPB = 256 * GEMM_KTB;    // one operand's k-tile: 256 rows x 128 B = 32 KB
constexpr int P8_LDS_BYTES = 4 * P8_OPB;  // A buf 0 | A buf 1 | W buf 0 | W buf 1

// ABL (microbenchmark ablations): bit 0 = no epilogue, bit 3 = no MFMAs
template <int NSPLIT, typename Epi, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_p8_kernel(GemmCore g, Epi epi) {
  using namespace p8;
  static_assert(NSPLIT == 1 || NSPLIT == 2, "plain fp16 rows or MX lines");
  constexpr bool MX = NSPLIT == 2;
  constexpr int NPL = MX ? 2 : 1;
  constexpr int TM = 4, TN = 2, BM = 256, BN = 256, OPB = P8_OPB;
  F5_DYN_LDS(char, smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = pp::uniform(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;
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
  const int kbytes = g.K * 2 * NPL;   // bytes of one operand row; a multiple of 256 (launcher: an even number of k-tiles)
  const int nkt = kbytes / GEMM_KTB;  // k-tiles
  // the tile's rows start at the buffer descriptors' base: rows past the operand's end are out of range (zeros), no per-lane row test
  const int arows = g.a_rows - m0, wrows = g.w_rows - n0;
  const BufRsrc Ar = make_rsrc(reinterpret_cast<const char*>(g.A) + (int64_t)m0 * g.lda * 2, arows > 0 ? (uint32_t)((int64_t)(arows - 1) * g.lda * 2 + kbytes) : 0u);
  const BufRsrc Wr = make_rsrc(reinterpret_cast<const char*>(g.W) + (int64_t)n0 * g.ldw * 2, wrows > 0 ? (uint32_t)((int64_t)(wrows - 1) * g.ldw * 2 + kbytes) : 0u);

  // LDS-DMA: piece P of an operand tile = rows 8P .. 8P+7 -> bytes [1024 P, +1024) of its buffer; lane l brings row 8P + l/8, logical
  // 16-byte chunk (l%8) ^ swz(row) (the swizzle sits on the SOURCE side: the LDS destination of a DMA is lane-linear).  A wave's pieces of an
  // operand are 16, 32 or 64 rows apart — swz(row) = (row >> 1) & 7 is the same for all of them — so ONE per-lane offset serves every piece
  // and the distance rides in the scalar offset next to the k offset (2 long-lived VGPRs instead of 8).
  //   A quarters (a01 = rows 0-63 of both groups' halves, a23 = rows 64-127): this wave brings pieces pa, pa + 2 (+ 8 for a23)
  //   W quarters (w0 / w1 = first / second 32 rows of every 64-channel strip): pieces pw, pw + 2 (+ 4 for w1)
  const int pa = 16 * grp + 4 * (wn >> 1) + (wn & 1);
  const int pw = 8 * (wave >> 1) + (wave & 1);
  uint32_t qa, qw;
  {
    int row = 8 * pa + (lane >> 3), lc = (lane & 7) ^ ((row >> 1) & 7);
    qa = (uint32_t)((int64_t)row * g.lda * 2 + lc * 16);
    row = 8 * pw + (lane >> 3);
    lc = (lane & 7) ^ ((row >> 1) & 7);
    qw = (uint32_t)((int64_t)row * g.ldw * 2 + lc * 16);
  }
  const uint32_t a16 = (uint32_t)(16 * g.lda * 2), w16 = (uint32_t)(16 * g.ldw * 2);  // 16 rows further, in bytes
  char* const a_dst = smem + pa * 1024;
  char* const w_dst = smem + 2 * OPB + pw * 1024;
  auto issue_a = [&](auto H, auto BUF, int kt) {  // quarter a01 (H = 0) / a23 (1) of k-tile kt into buffer BUF
    constexpr int h = decltype(H)::value, buf = decltype(BUF)::value;
    if constexpr (ABL & 4) return;
#pragma unroll
    for (int p = 0; p < 2; ++p) pp::dma_b128(Ar, a_dst + buf * OPB + h * 8192 + p * 2048, qa, (uint32_t)kt * GEMM_KTB + (uint32_t)(4 * h + p) * a16);
  };
  auto issue_w = [&](auto H, auto BUF, int kt) {  // quarter w0 (H = 0) / w1 (1)
    constexpr int h = decltype(H)::value, buf = decltype(BUF)::value;
    if constexpr (ABL & 4) return;
#pragma unroll
    for (int p = 0; p < 2; ++p) pp::dma_b128(Wr, w_dst + buf * OPB + h * 4096 + p * 2048, qw, (uint32_t)kt * GEMM_KTB + (uint32_t)(2 * h + p) * w16);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int j = 0; j < TM; ++j)
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  // fragment addressing: lane (row = lane & 31, half = lane >> 5) reads 16-byte chunks of its row's line.  Plain fp16: chunk 2 ks + half
  // for the 16-wide k-step ks = 0..3.  MX lines: chunks half and 2 + half (the two hi k-steps), 4 + 2 half and 5 + 2 half (the lane's P words).
  const uint32_t lds0 = pp::lds_base(smem);
  uint32_t fa_addr[4];  // + 4096 * (32-row tile) + OPB * buffer as immediates; the W fragments' = + wdelta (wave-uniform), added at the read
  {
    const int r = lane & 31, fswz = (r >> 1) & 7, fh = lane >> 5;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int chunk = MX ? (x < 2 ? 2 * x + fh : 4 + 2 * fh + (x - 2)) : 2 * x + fh;
      const uint32_t o = (uint32_t)(r * GEMM_KTB + ((chunk ^ fswz) << 4));
      fa_addr[x] = lds0 + o + (uint32_t)(grp * 128 * GEMM_KTB);
    }
  }
  const uint32_t wdelta = (uint32_t)(2 * OPB + wn * 64 * GEMM_KTB - grp * 128 * GEMM_KTB);
  Frag fa[2][4], fw[2][4];  // the two row tiles of the current A pair / the two weight tiles x the four 16-byte reads of a line
  auto read_a = [&](auto JP, auto BUF) {  // row tiles 2 JP, 2 JP + 1 of this wave's 128 rows
    constexpr int jp = decltype(JP)::value, buf = decltype(BUF)::value;
    static_for<2>([&](auto JJ) {
      static_for<4>([&](auto X) {
        fa[decltype(JJ)::value][decltype(X)::value].u = pp::lds_read_b128<(2 * jp + decltype(JJ)::value) * 4096 + buf * OPB>(fa_addr[decltype(X)::value]);
      });
    });
  };
  auto read_w = [&](auto I, auto BUF) {  // weight tile I of this wave's 64 channels
    constexpr int i = decltype(I)::value, buf = decltype(BUF)::value;
    static_for<4>([&](auto X) { fw[i][decltype(X)::value].u = pp::lds_read_b128<i * 4096 + buf * OPB>(fa_addr[decltype(X)::value] + wdelta); });
  };
  auto mma_q = [&](auto JP, auto I) {  // quadrant (row tiles 2 JP, 2 JP + 1) x weight tile I over the whole k-tile
    constexpr int jp = decltype(JP)::value, i = decltype(I)::value;
    // every fragment re-defined behind the wait that preceded this call (common.h pin_after_wait: hipcc takes an asm read's destination as
    // written when the read is ISSUED)
    static_for<2>([&](auto JJ) { static_for<4>([&](auto X) { pin_after_wait(fa[decltype(JJ)::value][decltype(X)::value].u); }); });
    static_for<4>([&](auto X) { pin_after_wait(fw[i][decltype(X)::value].u); });
    if constexpr (ABL & 8) {
#ifndef F5_HIPEMU
      asm volatile("" ::"v"(fw[i][0].u.x), "v"(fw[i][3].u.w), "v"(fa[0][0].u.x), "v"(fa[1][3].u.w));
#endif
      return;
    }
    constexpr int NH = MX ? 2 : 4;  // fp16 MFMA k-steps of the line
#pragma unroll
    for (int x = 0; x < NH; ++x)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) Mma32<f16>::mma(acc[2 * jp + jj][i], fw[i][x], fa[jj][x]);
    if constexpr (MX) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) mx_mma(acc[2 * jp + jj][i], fw[i][2].u, fw[i][3].u, fa[jj][2].u, fa[jj][3].u);  // both correction terms of the line
    }
  };
  // end of a memory half: (this wave's pieces of) the next phase's operands have landed, then the barrier that ends the partner's matrix half
  auto mem_end = [&](auto VM) {
    if constexpr (decltype(VM)::value >= 0 && !(ABL & 4)) pp::wait_vmcnt<decltype(VM)::value>();
    pp::wg_barrier();
  };
  auto mat_half = [&](auto JP, auto I, bool closing) {
    pp::lds_wait();  // my fragment reads have landed; nothing is scheduled across
    prio<1>();
    mma_q(JP, I);
    prio<0>();
    pp::pin();
    if (closing) pp::wg_barrier();
  };
  using B0 = IC<0>;
  using B1 = IC<1>;
  using LO = IC<0>;  // a01 / w0 quarter, row tiles 0-1, weight tile 0
  using HI = IC<1>;  // a23 / w1

  // k-tiles t (even: buffer 0) and t + 1 (buffer 1).  MODE 0: steady state (k-tiles t + 2, t + 3 exist and are requested), 1: the last pair.
  auto pair = [&](auto MODE, int t) {
    constexpr bool st = decltype(MODE)::value == 0;
    // E0
    issue_a(HI{}, B1{}, t + 1);
    read_a(LO{}, B0{});
    mem_end(IC<10>{});
    mat_half(LO{}, LO{}, true);
    // E1
    if constexpr (st) issue_w(LO{}, B0{}, t + 2);
    read_w(HI{}, B0{});
    mem_end(IC<st ? 10 : 8>{});
    mat_half(LO{}, HI{}, true);
    // E2
    if constexpr (st) issue_a(LO{}, B0{}, t + 2);
    read_a(HI{}, B0{});
    mem_end(IC<st ? 10 : 6>{});
    mat_half(HI{}, HI{}, true);
    // E3: weight tile 1 is free (its last MFMAs were E2's): k-tile t + 1's is read now
    if constexpr (st) issue_w(HI{}, B0{}, t + 2);
    read_w(HI{}, B1{});
    mem_end(IC<st ? 10 : 4>{});
    mat_half(HI{}, LO{}, true);
    // O0
    if constexpr (st) issue_a(HI{}, B0{}, t + 2);
    read_a(LO{}, B1{});
    mem_end(IC<st ? 10 : 2>{});
    mat_half(LO{}, HI{}, true);
    // O1
    if constexpr (st) issue_w(HI{}, B1{}, t + 3);
    read_w(LO{}, B1{});
    mem_end(IC<st ? 10 : 0>{});
    mat_half(LO{}, LO{}, true);
    // O2
    if constexpr (st) issue_a(LO{}, B1{}, t + 3);
    read_a(HI{}, B1{});
    mem_end(IC<st ? 10 : -1>{});
    mat_half(HI{}, LO{}, true);
    // O3: weight tile 0 is free: k-tile t + 2's is read now
    if constexpr (st) {
      issue_w(LO{}, B1{}, t + 3);
      read_w(LO{}, B0{});
    }
    mem_end(IC<st ? 10 : -1>{});
    mat_half(HI{}, HI{}, st || grp == 0);  // group 1 started one barrier late: it skips the last one
  };

  // prologue: the seven quarters the steady state would have in flight, in its order; the first two landed and visible
  issue_w(LO{}, B0{}, 0);
  issue_a(LO{}, B0{}, 0);
  issue_w(HI{}, B0{}, 0);
  issue_a(HI{}, B0{}, 0);
  issue_w(HI{}, B1{}, 1);
  issue_a(LO{}, B1{}, 1);
  issue_w(LO{}, B1{}, 1);
  if constexpr (!(ABL & 4)) pp::wait_vmcnt<10>();
  pp::wg_barrier();
  if (grp == 1) pp::wg_barrier();  // half a phase behind group 0 from here on
  read_w(LO{}, B0{});
  // Group 1 issues this first weight read one barrier before group 0 starts refilling the same quarter (E1's issue_w(LO, B0, t + 2)) and would
  // otherwise wait for it only in its first matrix half, behind that barrier: in steady state every read of a quarter has been waited for
  // before its refill is issued, here not.  An LDS-DMA round trip is an order of magnitude longer than a ds_read, so this is a window in
  // principle only; one wait per launch closes it.
  if (grp == 1) pp::lds_wait();
  for (int t = 0; t < nkt - 2; t += 2) pair(IC<0>{}, t);
  pair(IC<1>{}, nkt - 2);

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
    // the epilogue's per-lane address arithmetic hangs off an opaque copy of the lane id: loop-invariant code motion would otherwise compute
    // it ahead of the k-loop and carry it through — with MX lines (fragment tuples + copies) that is what tipped the kernel into scratch
    int lane_e = lane;
#ifndef F5_HIPEMU
    asm volatile("" : "+v"(lane_e));
#endif
    pp_finish<TM, TN, Epi, !MX>(acc, g, epi, m0 + grp * 128, n0 + wn * 64, lane_e);
  }
}
