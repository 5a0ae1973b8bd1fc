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
//     have passed.
//   * PERSISTENT TILE LOOP: a launch is at most one workgroup per CU (gridDim = min(tiles, 256)), workgroup b takes the tiles b, b + grid, ..
//     of the XCD-aware tile order, and the k-tile stream never stops at a tile boundary: the last pair of k-tiles of tile i requests the
//     first two k-tiles of tile i + 1 in its usual slots (through the next tile's buffer descriptors — the per-lane offsets are tile
//     independent, a tile's rows beyond M fall outside ITS descriptor), so the next tile starts on landed operands instead of the 112 KB
//     prologue burst that every CU of a round would issue at the same moment (measured ~5 us per round at 90k rows, profiles/r05b_*), and
//     tile i's stores drain under tile i + 1's first phases.  Around the epilogue: `vmcnt(0)` first (everything requested so far has
//     landed — the epilogue's own first wait, for its bias loads, would wait for the same loads), group 0 passes the phase's closing
//     barrier BEFORE its epilogue and group 1 AFTER, so both groups' epilogues run side by side; the first five phases of the next tile
//     need no counted wait (all their operands were requested before the epilogue), the sixth waits with the usual count — by then the
//     stores, which retire in order with the loads, have had five phases to be acknowledged.
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
constexpr int P8_MAX_WGS = 256;           // one workgroup per CU (128 KB of LDS each)

// ABL (microbenchmark ablations): bit 0 = no epilogue, bit 2 = no LDS-DMA, bit 3 = no MFMAs
template <int NSPLIT, typename Epi, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_p8_kernel(GemmCore g, Epi epi, int ntiles) {
  using namespace p8;
  static_assert(NSPLIT == 1 || NSPLIT == 2, "plain fp16 rows or MX lines");
  constexpr bool MX = NSPLIT == 2;
  constexpr int NPL = MX ? 2 : 1;
  constexpr int TM = 4, TN = 2, BM = 256, BN = 256, OPB = P8_OPB;
  F5_DYN_LDS(char, smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = pp::uniform(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;
  const int kbytes = g.K * 2 * NPL;   // bytes of one operand row; a multiple of 256 (launcher: an even number of k-tiles, at least 4)
  const int nkt = kbytes / GEMM_KTB;  // k-tiles
  // tile v of the launch (as gemm_pp_kernel: XCD-contiguous runs — v and v + gridDim sit on the same XCD —, channel tiles fastest, optional
  // groups of row tiles) -> its first row / channel
  auto tile_origin = [&](int v, int& m0, int& n0) {
    const int nt = (g.N + BN - 1) / BN, nwg = ntiles;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = v & 7, slot = v >> 3;
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
  };
  // the buffer descriptors of a tile start at ITS first row: rows past the operand's end are out of range (zeros) with tile-independent lane offsets
  auto desc_a = [&](int m0) {
    const int rows = g.a_rows - m0;
    return make_rsrc(reinterpret_cast<const char*>(g.A) + (int64_t)m0 * g.lda * 2, rows > 0 ? (uint32_t)((int64_t)(rows - 1) * g.lda * 2 + kbytes) : 0u);
  };
  auto desc_w = [&](int n0) {
    const int rows = g.w_rows - n0;
    return make_rsrc(reinterpret_cast<const char*>(g.W) + (int64_t)n0 * g.ldw * 2, rows > 0 ? (uint32_t)((int64_t)(rows - 1) * g.ldw * 2 + kbytes) : 0u);
  };

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
  auto issue_a = [&](BufRsrc R, auto H, auto BUF, int kt) {  // quarter a01 (H = 0) / a23 (1) of k-tile kt into buffer BUF
    constexpr int h = decltype(H)::value, buf = decltype(BUF)::value;
    if constexpr (ABL & 4) return;
#pragma unroll
    for (int p = 0; p < 2; ++p) pp::dma_b128(R, a_dst + buf * OPB + h * 8192 + p * 2048, qa, (uint32_t)kt * GEMM_KTB + (uint32_t)(4 * h + p) * a16);
  };
  auto issue_w = [&](BufRsrc R, auto H, auto BUF, int kt) {  // quarter w0 (H = 0) / w1 (1)
    constexpr int h = decltype(H)::value, buf = decltype(BUF)::value;
    if constexpr (ABL & 4) return;
#pragma unroll
    for (int p = 0; p < 2; ++p) pp::dma_b128(R, w_dst + buf * OPB + h * 4096 + p * 2048, qw, (uint32_t)kt * GEMM_KTB + (uint32_t)(2 * h + p) * w16);
  };

  f32x16 acc[TM][TN];  // (defined at the top of every tile: not carried around the tile loop)
  auto zero_acc = [&] {
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
  };

  // fragment addressing: lane (row = lane & 31, half = lane >> 5) reads 16-byte chunks of its row's line.  Plain fp16: chunk 2 ks + half
  // for the 16-wide k-step ks = 0..3.  MX lines: chunks half and 2 + half (the two hi k-steps), 4 + 2 half and 5 + 2 half (the lane's P words).
  const uint32_t lds0 = pp::lds_base(smem);
  uint32_t fa_addr[4], fw_addr[4];  // + 4096 * (32-row tile) + OPB * buffer as immediates
  {
    const int r = lane & 31, fswz = (r >> 1) & 7, fh = lane >> 5;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int chunk = MX ? (x < 2 ? 2 * x + fh : 4 + 2 * fh + (x - 2)) : 2 * x + fh;
      const uint32_t o = (uint32_t)(r * GEMM_KTB + ((chunk ^ fswz) << 4));
      fa_addr[x] = lds0 + o + (uint32_t)(grp * 128 * GEMM_KTB);
      fw_addr[x] = lds0 + 2 * OPB + o + (uint32_t)(wn * 64 * GEMM_KTB);
    }
  }
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
    static_for<4>([&](auto X) { fw[i][decltype(X)::value].u = pp::lds_read_b128<i * 4096 + buf * OPB>(fw_addr[decltype(X)::value]); });
  };
  // One phase.  ISS: the quarter this phase requests (a callable), RD: its fragment reads, VM: the counted wait that makes the NEXT phase's
  // operands (this wave's pieces) landed before the barrier that ends the memory half (-1: none needed), (JP, I): the quadrant its matrix half
  // multiplies, closing: whether the matrix half ends with a barrier.
  auto phase = [&](auto ISS, auto RD, auto VM, auto JP, auto I, bool closing) {
    constexpr int vm = decltype(VM)::value;
    constexpr int jp = decltype(JP)::value, i = decltype(I)::value;
    // memory half
    ISS();
    RD();
    if constexpr (vm >= 0 && !(ABL & 4)) pp::wait_vmcnt<(vm >= 0 ? vm : 0)>();
    pp::wg_barrier();
    // matrix half
    pp::lds_wait();  // my fragment reads have landed; nothing is scheduled across
    prio<1>();
    // every fragment re-defined behind the wait (common.h pin_after_wait: hipcc takes an asm read's destination as written when the read is ISSUED)
    static_for<2>([&](auto JJ) { static_for<4>([&](auto X) { pin_after_wait(fa[decltype(JJ)::value][decltype(X)::value].u); }); });
    static_for<4>([&](auto X) { pin_after_wait(fw[i][decltype(X)::value].u); });
    if constexpr (ABL & 8) {
#ifndef F5_HIPEMU
      asm volatile("" ::"v"(fw[i][0].u.x), "v"(fw[i][3].u.w), "v"(fa[0][0].u.x), "v"(fa[1][3].u.w));
#endif
    } else {
      constexpr int NH = MX ? 2 : 4;  // fp16 MFMA k-steps of the line
#pragma unroll
      for (int x = 0; x < NH; ++x)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) Mma32<f16>::mma(acc[2 * jp + jj][i], fw[i][x], fa[jj][x]);
      if constexpr (MX) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) mx_mma(acc[2 * jp + jj][i], fw[i][2].u, fw[i][3].u, fa[jj][2].u, fa[jj][3].u);  // both correction terms of the line
      }
    }
    prio<0>();
    pp::pin();
    if (closing) pp::wg_barrier();
  };
  using B0 = IC<0>;
  using B1 = IC<1>;
  using LO = IC<0>;  // a01 / w0 quarter, row tiles 0-1, weight tile 0
  using HI = IC<1>;  // a23 / w1

  // The pair of k-tiles t (even: buffer 0), t + 1 (buffer 1) of the current tile (descriptors Ac, Wc).
  // LAST 0: k-tiles t + 2, t + 3 of this tile are requested;  2: the last pair of a tile — the NEXT tile's k-tiles 0, 1 are requested through
  // An, Wn (a workgroup's last tile has empty ones: its requests read nothing and land as zeros nobody reads — one code path, no second
  // copy of the body whose accumulators would have to be merged with this one's at a join).
  // FIRST 1: the first pair of a tile — phases E0 .. O0 read operands that landed before the tile began (the vmcnt(0) in front of the
  // epilogue / behind the prologue), so they carry no counted wait; O1 is the first phase whose successor reads a quarter requested since.
  auto pair = [&](auto LASTC, auto FIRSTC, int t, BufRsrc Ac, BufRsrc Wc, BufRsrc An, BufRsrc Wn) {
    constexpr int last = decltype(LASTC)::value;
    constexpr bool first = decltype(FIRSTC)::value != 0;
    const int k2 = last == 2 ? 0 : t + 2, k3 = last == 2 ? 1 : t + 3;
    const BufRsrc A2 = last == 2 ? An : Ac, W2 = last == 2 ? Wn : Wc;
    constexpr int V = first ? -1 : 10;
    // E0
    phase([&] { issue_a(Ac, HI{}, B1{}, t + 1); }, [&] { read_a(LO{}, B0{}); }, IC<V>{}, LO{}, LO{}, true);
    // E1
    phase([&] { issue_w(W2, LO{}, B0{}, k2); }, [&] { read_w(HI{}, B0{}); }, IC<V>{}, LO{}, HI{}, true);
    // E2
    phase([&] { issue_a(A2, LO{}, B0{}, k2); }, [&] { read_a(HI{}, B0{}); }, IC<V>{}, HI{}, HI{}, true);
    // E3: weight tile 1 is free (its last MFMAs were E2's): k-tile t + 1's is read now
    phase([&] { issue_w(W2, HI{}, B0{}, k2); }, [&] { read_w(HI{}, B1{}); }, IC<V>{}, HI{}, LO{}, true);
    // O0
    phase([&] { issue_a(A2, HI{}, B0{}, k2); }, [&] { read_a(LO{}, B1{}); }, IC<V>{}, LO{}, HI{}, true);
    // O1
    phase([&] { issue_w(W2, HI{}, B1{}, k3); }, [&] { read_w(LO{}, B1{}); }, IC<10>{}, LO{}, LO{}, true);
    // O2
    phase([&] { issue_a(A2, LO{}, B1{}, k3); }, [&] { read_a(HI{}, B1{}); }, IC<10>{}, HI{}, LO{}, true);
    // O3: weight tile 0 is free: the next k-tile's (k-tile t + 2, or the next tile's first) is read now.  The closing barrier: always inside a
    // tile; at a tile's end group 0 passes it BEFORE its epilogue and group 1 AFTER (tile loop below) or — the workgroup's last tile — never
    // (it started one barrier late)
    phase([&] { issue_w(W2, LO{}, B1{}, k3); }, [&] { read_w(LO{}, B0{}); }, IC<10>{}, HI{}, HI{}, last == 0 || grp == 0);
  };

  int v = blockIdx.x, m0, n0;
  tile_origin(v, m0, n0);
  BufRsrc Ac = desc_a(m0), Wc = desc_w(n0);
  // prologue: the seven quarters a tile finds requested by its predecessor, in the steady state's order — all landed and visible, as behind an epilogue
  issue_w(Wc, LO{}, B0{}, 0);
  issue_a(Ac, LO{}, B0{}, 0);
  issue_w(Wc, HI{}, B0{}, 0);
  issue_a(Ac, HI{}, B0{}, 0);
  issue_w(Wc, HI{}, B1{}, 1);
  issue_a(Ac, LO{}, B1{}, 1);
  issue_w(Wc, LO{}, B1{}, 1);
  if constexpr (!(ABL & 4)) pp::wait_vmcnt<0>();
  pp::wg_barrier();
  if (grp == 1) pp::wg_barrier();  // half a phase behind group 0 from here on
  read_w(LO{}, B0{});
  while (true) {
    zero_acc();
    const int vn = v + (int)gridDim.x;
    const bool has_next = vn < ntiles;
    int mn = 0, nn = 0;
    if (has_next) tile_origin(vn, mn, nn);
    const BufRsrc An = desc_a(has_next ? mn : g.a_rows), Wn = desc_w(has_next ? nn : g.w_rows);  // (no successor: empty descriptors)
    pair(IC<0>{}, IC<1>{}, 0, Ac, Wc, An, Wn);
    for (int t = 2; t < nkt - 2; t += 2) pair(IC<0>{}, IC<0>{}, t, Ac, Wc, An, Wn);
    pair(IC<2>{}, IC<0>{}, nkt - 2, Ac, Wc, An, Wn);
    // everything requested so far (the next tile's first seven quarters) has landed: its first phases need no counted wait
    if constexpr (!(ABL & 4)) pp::wait_vmcnt<0>();
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
      pp_unscale<TM, TN>(acc, g, n0 + wn * 64, lane);
      epi.template tile<TM, TN>(acc, m0 + grp * 128, n0 + wn * 64, lane);
    }
    if (!has_next) break;
    pp::pin();
    if (grp == 1) pp::wg_barrier();  // group 1's closing barrier of the tile's last phase, behind its epilogue (group 0 passed it before its own)
    v = vn; m0 = mn; n0 = nn;
    Ac = An; Wc = Wn;
  }
}
