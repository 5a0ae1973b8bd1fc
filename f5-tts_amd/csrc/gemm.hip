// gemm.hip — instantiations and launch heuristics of the MFMA GEMMs: the generic LDS-tiled kernels of gemm.h (any epilogue, any
// operand mode, batched) and the pipelined kernel of gemm_pp.h for the DiT / UNetT block projections (fp16 / fp16x3, wave-tile epilogues)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "kernels.h"
#include "gemm_pp.h"

// gemm_p8.hip: the ping-pong 256x256 kernel of the many-round launches (variant id 80; gemm_p8.h)
template <int NSPLIT, typename Epi>
hipError_t launch_p8(const GemmCore& g, const Epi& e, int abl, hipStream_t s);
bool p8_applies(int nsplit, const GemmCore& g);
hipError_t init_p8_kernels();

namespace {

// One tile variant: TM x TN 32x32 tiles per wave, WGM x WGN waves.
template <int TM_, int TN_, int WGM_, int WGN_>
struct V {
  static constexpr int TM = TM_, TN = TN_, WGM = WGM_, WGN = WGN_;
  static constexpr int BM = 32 * WGM * TM, BN = 32 * WGN * TN;
};

// variant ids (rows x output channels): 0 = 64x128, 1 = 128x64, 2 = 128x128 (4 waves), 8 = 64x64
// every mode stages one 128-byte line per operand row per k-tile, so LDS per workgroup is the same in all modes
template <typename T, int NSPLIT, int ID>
struct Variant;
#define F5_VARIANT(ID, TM, TN, WGM, WGN)                                                    \
  template <typename T, int NSPLIT>                                                         \
  struct Variant<T, NSPLIT, ID> : V<TM, TN, WGM, WGN> {}
F5_VARIANT(0, 1, 2, 2, 2);
F5_VARIANT(1, 2, 1, 2, 2);
F5_VARIANT(2, 2, 2, 2, 2);
F5_VARIANT(8, 1, 1, 2, 2);  // 64x64
#undef F5_VARIANT

template <typename T, int NSPLIT, int ID, typename Epi>
hipError_t set_attr() {
  using C = Variant<T, NSPLIT, ID>;
  constexpr int lds = gemm_lds_bytes<T, NSPLIT, C::TM, C::TN, C::WGM, C::WGN>();
  return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<T, NSPLIT, C::TM, C::TN, Epi, C::WGM, C::WGN>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

template <typename T, int NSPLIT, typename Epi>
hipError_t set_attrs_op() {
  hipError_t e;
  if ((e = set_attr<T, NSPLIT, 0, Epi>()) != hipSuccess) return e;
  if ((e = set_attr<T, NSPLIT, 1, Epi>()) != hipSuccess) return e;
  if ((e = set_attr<T, NSPLIT, 2, Epi>()) != hipSuccess) return e;
  if ((e = set_attr<T, NSPLIT, 8, Epi>()) != hipSuccess) return e;
  return hipSuccess;
}

template <typename Epi>
hipError_t set_attrs_epi() {
  hipError_t e;
  if ((e = set_attrs_op<float, 1, Epi>()) != hipSuccess) return e;
  if ((e = set_attrs_op<f16, 1, Epi>()) != hipSuccess) return e;
  if ((e = set_attrs_op<f16, 3, Epi>()) != hipSuccess) return e;
  return hipSuccess;
}

template <typename T, int NSPLIT, int ID, typename Epi>
hipError_t launch_one(const GemmCore& g, const Epi& e, int batch, hipStream_t s) {
  using C = Variant<T, NSPLIT, ID>;
  constexpr int lds = gemm_lds_bytes<T, NSPLIT, C::TM, C::TN, C::WGM, C::WGN>();
  auto kern = gemm_kernel<T, NSPLIT, C::TM, C::TN, Epi, C::WGM, C::WGN>;
  // operands are addressed with 32-bit byte offsets through buffer descriptors; offsets >= 2 GiB mean "no such row"
  if ((int64_t)g.a_rows * g.lda * (int64_t)sizeof(T) >= (int64_t)0x7ff00000 || (int64_t)g.w_rows * g.ldw * (int64_t)sizeof(T) >= (int64_t)0x7ff00000)
    return hipErrorInvalidValue;
  dim3 grid(((g.M + C::BM - 1) / C::BM) * ((g.N + C::BN - 1) / C::BN), 1, batch);
  hipLaunchKernelGGL(kern, grid, dim3(64 * C::WGM * C::WGN), lds, s, g, e);
  return hipGetLastError();
}

int pick_variant(const GemmCore& g, int batch) {
  static const int forced = [] { const char* e = getenv("F5HIP_GEMM_VARIANT"); return e ? atoi(e) : -1; }();  // tuning / test knob
  if (forced >= 0) return forced;
  if (g.M <= 64) return 0;
  // 128x64 tiles (3 workgroups per CU) until the grid is several waves deep, then 128x128 (higher FLOP per byte staged)
  const int64_t big = (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128) * batch;
  if (big >= 1024) return 2;
#ifndef F5_HIPEMU
  return (g.N <= 1024 && g.M > 256 && batch == 1) ? 6 : 1;  // small grids, few channel tiles: the direct-to-LDS ring of the same tile
#else
  return 1;
#endif
}

#ifndef F5_HIPEMU
// the dynamic-LDS limit of every gemm_glds_kernel instantiation launch_tiled can reach (variant 6), for the CURRENT device
template <typename T, int NSPLIT, typename Epi>
hipError_t set_glds_attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_glds_kernel<T, NSPLIT, 2, 1, Epi, 2, 2, 3, 0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             gemm_glds_lds_bytes<T, NSPLIT, 2, 1, 2, 2, 3>());
}
template <typename Epi>
hipError_t set_glds_attrs() {
  hipError_t e;
  if ((e = set_glds_attr<float, 1, Epi>()) != hipSuccess) return e;
  if ((e = set_glds_attr<f16, 1, Epi>()) != hipSuccess) return e;
  return set_glds_attr<f16, 3, Epi>();
}
// direct-to-LDS ring variant of the 128x64 tile (variant id 6)
template <typename T, int NSPLIT, int TM, int TN, typename Epi, int WGM = 2, int WGN = 2, int NS = 3, int PRIO = 0>
hipError_t launch_glds(const GemmCore& g, const Epi& e, int batch, hipStream_t s) {
  constexpr int lds = gemm_glds_lds_bytes<T, NSPLIT, TM, TN, WGM, WGN, NS>();
  auto kern = gemm_glds_kernel<T, NSPLIT, TM, TN, Epi, WGM, WGN, NS, PRIO>;  // (dynamic-LDS limit: set_glds_attrs, per device, at context creation)
  if ((int64_t)g.a_rows * g.lda * (int64_t)sizeof(T) >= (int64_t)0x7ff00000 || (int64_t)g.w_rows * g.ldw * (int64_t)sizeof(T) >= (int64_t)0x7ff00000)
    return hipErrorInvalidValue;
  constexpr int BM = 32 * WGM * TM, BN = 32 * WGN * TN;
  dim3 grid(((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN), 1, batch);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WGM * WGN), lds, s, g, e);
  return hipGetLastError();
}
#endif  // F5_HIPEMU

template <typename T, int NSPLIT, typename Epi>
hipError_t launch_tiled(const GemmCore& g, const Epi& e, int batch, int variant, hipStream_t s) {
  if (variant < 0) variant = pick_variant(g, batch);
  switch (variant) {
    case 0: return launch_one<T, NSPLIT, 0, Epi>(g, e, batch, s);
    case 1: return launch_one<T, NSPLIT, 1, Epi>(g, e, batch, s);
    case 2: return launch_one<T, NSPLIT, 2, Epi>(g, e, batch, s);
    case 8: return launch_one<T, NSPLIT, 8, Epi>(g, e, batch, s);
#ifndef F5_HIPEMU
    case 6: return launch_glds<T, NSPLIT, 2, 1, Epi>(g, e, batch, s);
#else  // the host shim runs the register-staged kernel of the same tile at the LDS-DMA variant's call sites
    case 6: return launch_one<T, NSPLIT, 1, Epi>(g, e, batch, s);
#endif
    default: break;
  }
  return hipErrorInvalidValue;
}

// ---- the pipelined kernel (gemm_pp.h) -----------------------------------------------------------------------------------------------
// variant ids 50..: workgroup tile (rows x channels), waves, ring depth
//   50  256x256  8 waves of 128x64   2 stages      51  256x128  8 waves of 64x64   3 stages      52  128x256  8 waves of 64x64  3 stages
//   53  192x64   4 waves of  96x32   3 stages      54  192x128  4 waves of 96x64   3 stages      55  192x128  8 waves of 96x32  3 stages
//   56  192x192  4 waves of  96x96   3 stages      57  128x128  4 waves of 64x64   3 stages      58  128x128  4 waves of 64x64  2 stages (2 / CU)
//   59   96x128  4 waves of  96x32   3 stages      60  256x128  4 waves of 128x64  3 stages
//   61-64  2-stage 4-wave tiles that fit two workgroups per CU (microbenchmark only, see pick_pp_variant)
//   65  192x128  66  96x128  67  192x64: k-split, 2 groups x 4 waves on alternate k-tiles of one output tile
//   68  192x192  69  192x128  70  192x64: k-step split, 2 groups x 4 waves on alternate k-steps of the same k-tiles
template <int ID>
struct PpV;
#define F5_PPV(ID, TM_, TN_, WGM_, WGN_, NS_, JG_, ...)                                   \
  template <>                                                                             \
  struct PpV<ID> {                                                                        \
    static constexpr int TM = TM_, TN = TN_, WGM = WGM_, WGN = WGN_, NS = NS_, JG = JG_;  \
    static constexpr int BM = 32 * WGM * TM, BN = 32 * WGN * TN;                          \
    static constexpr int KSP = (1 __VA_ARGS__) == 2 ? 2 : 1, KSS = (1 __VA_ARGS__) == 3 ? 2 : 1; /* +1: k-split, +2: k-step split */ \
  }
F5_PPV(50, 4, 2, 2, 4, 2, 1);
F5_PPV(51, 2, 2, 4, 2, 3, 2);
F5_PPV(52, 2, 2, 2, 4, 3, 2);
F5_PPV(53, 3, 1, 2, 2, 3, 3);
F5_PPV(54, 3, 2, 2, 2, 3, 3);
F5_PPV(55, 3, 1, 2, 4, 3, 3);
F5_PPV(56, 3, 3, 2, 2, 3, 1);
F5_PPV(57, 2, 2, 2, 2, 3, 2);
F5_PPV(58, 2, 2, 2, 2, 2, 2);
F5_PPV(59, 3, 1, 1, 4, 3, 3);
F5_PPV(60, 4, 2, 2, 2, 3, 1);
F5_PPV(61, 2, 3, 2, 2, 2, 2);  // 128x192, 4 waves of 64x96, 2 stages = 80 KB: two workgroups per CU
F5_PPV(62, 3, 2, 2, 2, 2, 3);  // 192x128, 4 waves of 96x64, 2 stages = 80 KB
F5_PPV(63, 3, 1, 1, 4, 2, 3);  //  96x128, 4 waves of 96x32, 2 stages = 56 KB
F5_PPV(64, 3, 1, 2, 2, 2, 3);  // 192x64,  4 waves of 96x32, 2 stages = 64 KB
// k-split (gemm_pp.h): two groups of 4 waves on alternate k-tiles of one output tile, 2 stages of 2 k-tiles
F5_PPV(65, 3, 2, 2, 2, 2, 3, +1);  // 192x128, 2 x 4 waves of 96x64 = 160 KB
F5_PPV(66, 3, 1, 1, 4, 2, 3, +1);  //  96x128, 2 x 4 waves of 96x32 = 112 KB
F5_PPV(67, 3, 1, 2, 2, 2, 3, +1);  // 192x64,  2 x 4 waves of 96x32 = 128 KB
// k-step split (gemm_pp.h): two groups of 4 waves on alternate 16-wide k-steps of the same k-tiles, one ring filled by all 8 waves
F5_PPV(68, 3, 3, 2, 2, 3, 1, +2);  // 192x192, 2 x 4 waves of 96x96, 3 stages = 144 KB
F5_PPV(69, 3, 2, 2, 2, 3, 3, +2);  // 192x128, 2 x 4 waves of 96x64, 3 stages = 120 KB
F5_PPV(70, 3, 1, 2, 2, 3, 3, +2);  // 192x64,  2 x 4 waves of 96x32, 3 stages =  96 KB
// (round 4, measured and removed: 256x256 as FOUR waves of 128x128 — one wave per SIMD, 16 accumulator tiles in AGPRs, 2/3 of the 8-wave tile's
// fragment reads — for MX lines, where LDS bandwidth bounds the 8-wave tile: 8-10 % SLOWER at M = 11k .. 90k (881 against 816 us on FF1 at
// 90k rows; profiles/r04e_tiles_4wave.log): with one wave per SIMD nothing hides the fragment waits and the barrier)
// (deeper rings — 192x64 k-step split x 5 stages, 192x128 x 4, 96x128 / 4 waves x 5, 192x128 / 8 waves x 4 — measured the same or 1-3 % slower
// than the 3-stage tiles: the one-round k-loops are not waiting for their DMA; profiles/r02e_deep_rings.log)
#undef F5_PPV

// what the pipelined kernel needs from a launch: fp16 operands whose rows are whole 128-byte k-tiles (at least 3 of them), channel
// count a multiple of 32, one batch, every byte offset of the operands within 31 bits
constexpr int pp_planes(int nsplit) { return nsplit == 1 ? 1 : 2; }  // 128-byte lines hold 64 k (plain fp16) or 32 k (hi | lo, hi | MX words)
template <int NSPLIT>
bool pp_applies(const GemmCore& g, int batch) {
  const int64_t kbytes = (int64_t)g.K * 2 * pp_planes(NSPLIT);
  return batch == 1 && g.strideA == 0 && g.strideW == 0 && kbytes % GEMM_KTB == 0 && kbytes / GEMM_KTB >= 3 && g.N % 32 == 0 && g.M >= 1 &&
         (int64_t)g.a_rows * g.lda * 2 < (int64_t)0x7ff00000 && (int64_t)g.w_rows * g.ldw * 2 < (int64_t)0x7ff00000;
}

template <int NSPLIT, int ID>
constexpr int pp_ns() { return PpV<ID>::NS; }

// "this tile does not take this launch" (the caller falls back to the generic kernel): a value no HIP call returns here, so that a real
// launch failure is never mistaken for it
constexpr hipError_t PP_NOT_APPLICABLE = hipErrorNotSupported;

template <int NSPLIT, int ID, typename Epi, int ABL = 0>
hipError_t launch_pp_one(const GemmCore& g, const Epi& e, hipStream_t s) {
  using C = PpV<ID>;
  constexpr int NSX = pp_ns<NSPLIT, ID>();
  constexpr int lds = gemm_pp_lds_bytes<C::TM, C::TN, C::WGM, C::WGN, NSX, C::KSP>();
  static_assert(lds <= 160 * 1024, "ring does not fit the LDS");
  static_assert(C::KSP * C::KSS == 1 || C::WGM * C::WGN * C::TM * C::TN * 4096 <= lds, "the partial-sum exchange of the split tiles reuses the ring");
  if constexpr (C::KSS > 1) {  // even / odd tiles alternate the fragment buffers: an even number of k-tiles, a whole pipeline
    const int64_t kt = (int64_t)g.K * 2 * pp_planes(NSPLIT) / GEMM_KTB;
    if (kt % 2 != 0 || kt < NSX + 1) return PP_NOT_APPLICABLE;
  }
  if constexpr (C::KSP > 1) {  // each group needs its own whole pipeline: k-tiles split evenly, at least NS + 1 per group
    const int64_t kt = (int64_t)g.K * 2 * pp_planes(NSPLIT) / GEMM_KTB;
    if (kt % C::KSP != 0 || kt / C::KSP < NSX + 1) return PP_NOT_APPLICABLE;
  }
  auto kern = gemm_pp_kernel<f16, NSPLIT, C::TM, C::TN, C::WGM, C::WGN, NSX, C::JG, Epi, ABL, C::KSP, C::KSS>;
  if constexpr (ABL != 0) {  // microbenchmark ablations only: the production instantiations get their limit in init_gemm_kernels()
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (err != hipSuccess) return err;
  }
  dim3 grid(((g.M + C::BM - 1) / C::BM) * ((g.N + C::BN - 1) / C::BN), 1, 1);
  static const bool trace = getenv("F5HIP_GEMM_TRACE") != nullptr;  // which kernel ran (tests, tuning)
  if (trace) fprintf(stderr, "gemm_pp variant %d (%dx%d) nsplit %d M=%d N=%d K=%d grid %u\n", ID, C::BM, C::BN, NSPLIT, g.M, g.N, g.K, grid.x);
  hipLaunchKernelGGL(kern, grid, dim3(64 * C::WGM * C::WGN * C::KSP * C::KSS), lds, s, g, e);
  return hipGetLastError();
}

// the tiles instantiated for MX lines (NSPLIT 2): the tiles pick_pp_variant can choose in that mode plus their microbenchmark
// alternatives — every instantiation is a minute of build time
#define F5_MX_TILES(X) X(50) X(54) X(55) X(56) X(59) X(61) X(62) X(63) X(66) X(68) X(69)
template <int NSPLIT, typename Epi>
hipError_t launch_pp(const GemmCore& g, const Epi& e, int variant, hipStream_t s) {
  if (variant == 80) {  // the ping-pong kernel (gemm_p8.h): plain fp16 rows and MX lines
    if constexpr (NSPLIT == 1 || NSPLIT == 2) return launch_p8<NSPLIT, Epi>(g, e, 0, s);
    else return PP_NOT_APPLICABLE;
  }
  if constexpr (NSPLIT == 2) {
    switch (variant) {
#define F5_CASE(ID) case ID: return launch_pp_one<2, ID, Epi>(g, e, s);
      F5_MX_TILES(F5_CASE)
#undef F5_CASE
      default: return PP_NOT_APPLICABLE;
    }
  } else {
  switch (variant) {
    case 50: return launch_pp_one<NSPLIT, 50, Epi>(g, e, s);
    case 51: return launch_pp_one<NSPLIT, 51, Epi>(g, e, s);
    case 52: return launch_pp_one<NSPLIT, 52, Epi>(g, e, s);
    case 53: return launch_pp_one<NSPLIT, 53, Epi>(g, e, s);
    case 54: return launch_pp_one<NSPLIT, 54, Epi>(g, e, s);
    case 55: return launch_pp_one<NSPLIT, 55, Epi>(g, e, s);
    case 56: return launch_pp_one<NSPLIT, 56, Epi>(g, e, s);
    case 57: return launch_pp_one<NSPLIT, 57, Epi>(g, e, s);
    case 58: return launch_pp_one<NSPLIT, 58, Epi>(g, e, s);
    case 59: return launch_pp_one<NSPLIT, 59, Epi>(g, e, s);
    case 60: return launch_pp_one<NSPLIT, 60, Epi>(g, e, s);
    case 61: return launch_pp_one<NSPLIT, 61, Epi>(g, e, s);
    case 62: return launch_pp_one<NSPLIT, 62, Epi>(g, e, s);
    case 63: return launch_pp_one<NSPLIT, 63, Epi>(g, e, s);
    case 64: return launch_pp_one<NSPLIT, 64, Epi>(g, e, s);
    case 65: return launch_pp_one<NSPLIT, 65, Epi>(g, e, s);
    case 66: return launch_pp_one<NSPLIT, 66, Epi>(g, e, s);
    case 67: return launch_pp_one<NSPLIT, 67, Epi>(g, e, s);
    case 68: return launch_pp_one<NSPLIT, 68, Epi>(g, e, s);
    case 69: return launch_pp_one<NSPLIT, 69, Epi>(g, e, s);
    case 70: return launch_pp_one<NSPLIT, 70, Epi>(g, e, s);
    default: return PP_NOT_APPLICABLE;
  }
  }
}

// Tile choice.  A launch costs rounds x (time of one workgroup), so prefer the tile whose workgroup count fills whole rounds of the CUs
// with the largest wave tiles; measured tables: DESIGN.md section 4 (tools/kernel_bench.py, profiles/r02*).
inline int64_t ktiles_of(const GemmCore& g, int nsplit_planes) { return (int64_t)g.K * 2 * nsplit_planes / GEMM_KTB; }
int pick_pp_variant(const GemmCore& g, int nsplit_planes, bool qkv = false, bool mx = false, bool act16 = false) {  // nsplit_planes: 1 plain fp16 rows, 2 packed hi | lo rows
  static const int forced = [] { const char* e = getenv("F5HIP_PP_VARIANT"); return e ? atoi(e) : -1; }();  // tuning knob; 0 = never use the pipelined kernel
  if (forced >= 0) return forced;
  static const int f3072 = [] { const char* e = getenv("F5HIP_PP_VARIANT_N3072"); return e ? atoi(e) : -1; }();  // per-shape tuning knobs (tools/)
  static const int f2048 = [] { const char* e = getenv("F5HIP_PP_VARIANT_N2048"); return e ? atoi(e) : -1; }();
  static const int f1024 = [] { const char* e = getenv("F5HIP_PP_VARIANT_N1024"); return e ? atoi(e) : -1; }();
  if (g.N == 3072 && f3072 >= 0) return f3072;
  if (g.N == 2048 && f2048 >= 0) return f2048;
  if (g.N == 1024 && f1024 >= 0) return f1024;
  // measured on MI355X (profiles/r02b_kernel_bench.md; fp16x3, the DiT Base shapes N = 3072 / 2048 / 1024, K = 1024 / 2048):
  //   M >= 40k   (B = 32): 256x256, 8 waves of 128x64 — 352-377 TF against 322-350 for 128x128 x 2 per CU
  //   4k .. 40k  (B = 2..16): round 2: 256x128, 8 waves of 64x64, 3 stages (320-355 TF at M = 22k); round 3: two workgroups per CU, below
  //   2k .. 4k   (B = 1, one chain of 2 x 1406 rows): one round of 240 workgroups — 192x192 for N = 3072 (62 us against 84 for the
  //              128x64 tiles of gemm.h), 192x128 / 8 waves for N = 2048 (41 against 53), 96x128 for N = 1024 (26 / 45 against 32 / 52)
  //   < 2k       (B = 1, one CFG chain of 1406 rows): 192x128 / 8 waves for N = 3072, 96x128 otherwise
  if (mx) {  // MX lines (fp16m), measured (profiles/r04c_kernel_bench_fp16m_vs_fp16x3.log; us, fp16x3's choice first): small launches
             // stay on the pipelined kernel (the mode has no generic-kernel fallback)
    //   M = 90k: 256x256 everywhere (FF1 1136 -> 776, FF2 1110 -> 812)
    //   M = 11k .. 22k: FF1 (GELU -> MX rows epilogue) 256x256 (150 -> 118 at 11k; two-per-CU 128-130), out / FF2 192x128 two per CU
    //   (22k: 292 -> 210; 256x256 231), q|k|v 128x192 two per CU (260 -> 177)
    //   M = 2812: q|k|v 192x192 (59 -> 48), FF1 192x128 / 8 waves (41 -> 32), out / FF2 96x128 / 4 waves (24 / 40 -> 21 / 34; its k-split 21 / 35)
    // round 5: the ping-pong 256x256 kernel (gemm_p8.h, id 80) from 16k rows — FF1-type launches from 8k, where the lockstep 256x256 tile was
    // the choice — (us, lockstep / two-per-CU choice -> ping-pong; profiles/r05a_p8_kernel_bench.log, r05c_*): M = 90k FF1 818 -> 757, FF2 769 ->
    // 688, q|k|v 1225 -> 1106, out 435 -> 387; M = 45k FF1 425 -> 377; M = 22k FF1 201 -> 184, gate / residual launches 249 -> 233, q|k|v at
    // 45k rows 697 -> 665; at 11k rows the two-per-CU tiles stay ahead on everything but FF1 (123 -> 110)
    if ((g.M >= 16384 || (act16 && g.M >= 8192)) && p8_applies(2, g)) return 80;
    if (g.M >= 40000) return 50;
    if (g.M >= 4096) return qkv ? 61 : (act16 && g.M >= 8192) ? 50 : (g.N >= 2048 || g.M >= 8192) ? 62 : 63;
    constexpr bool kss = true;  // the k-step-split tiles for the one-round launches (q|k|v 44.8 -> 40.7 us, FF1 32 -> 30.4; profiles/r04d_kernel_bench_mx.log)
    const bool even_kt = ktiles_of(g, 2) % 2 == 0 && ktiles_of(g, 2) >= 4;
    if (g.M >= 2048) return g.N >= 3072 ? (kss && even_kt ? 68 : 56) : g.N >= 2048 ? (kss && even_kt ? 69 : 55) : 59;
    return g.N >= 3072 ? 55 : 59;
  }
  if (g.M < 512) return 0;  // a handful of row tiles: the generic small tiles
  // plain fp16 rows: the ping-pong kernel from 16k rows (M = 22k FF1 122 -> 111 us, 45k 252 -> 220, 90k 534 -> 445 = 850 TF against hipBLASLt's
  // 1165 without an epilogue; profiles/r05a_p8_kernel_bench.log, r05c_*); fp16x3 rows have no such kernel
  if (nsplit_planes == 1 && g.M >= 16384 && p8_applies(1, g)) return 80;
  if (g.M >= 40000) return 50;
  // A few rounds (B = 2 .. 16): the 4-wave, 2-stage tiles that fit TWO workgroups per CU - the two run out of phase, one's prologue and
  // epilogue under the other's k-loop.  Kept out of the engine in round 2 because of wrong rope values that came and went with the build;
  // round 3 traced those to a gfx950 packed-fp32 operand fault (Makefile NOPK), not to the tiles.  Measured against
  // the 8-wave tiles they replace (profiles/r03b_kernel_bench_2percu.log; fp16x3, us): M = 5.6k q|k|v 120 -> 109, FF1 95 -> 73, out / FF2
  // 48 / 84 -> 43 / 76 (96x128); M = 11k 269 -> 229, 165 -> 141, 97 / 173 -> 77 / 139; M = 22k 518 -> 431, 334 -> 276, 174 / 306 -> 151 / 282.
  if (g.M >= 4096) return qkv ? 61 : g.N >= 2048 ? 62 : g.M >= 8192 ? 62 : 63;
  // the fused q|k|v projection (rope + scatter epilogue, tools/kernel_bench.py qkv): 192x192 in the one-round regime (63 us against 73
  // for 192x128 / 8 waves), 192x128 / 8 waves for one CFG chain (39 against 43-56) and for a few rounds (230 us at M = 11k against 244)
  if (qkv && g.M < 2048) return 55;
  // narrow outputs (out-proj, FF2; FF1 of one chain): the k-split 96x128 — two waves per SIMD on one output tile, -10 % against the 4-wave
  // 96x128 (M = 2812: 23.9 / 39.9 us against 26.4 / 44.5; M = 1406: 20.9 / 34.0 / 23.6 against 23.5 / 39.6 / 25.8; profiles/r02c_ksplit.log).
  // For 192x128 it measures the same as the 8 waves of 96x32 (41.0 / 41.3), so that one stays.
  const int narrow = (ktiles_of(g, nsplit_planes) % 2 == 0 && ktiles_of(g, nsplit_planes) / 2 >= 3) ? 66 : 59;  // k-split: k-tiles split evenly, a whole pipeline (NS + 1 = 3) per group
  // one round of 240 workgroups (2048 <= M < 4096): the k-step-split 192x192 / 192x128 — 8 waves on the 4-wave tiles' ring, 57.0 against
  // 62.2 us (q|k|v, 192x192 / 4 waves) and 40.0 against 41.2 (FF1, 192x128 / 8 waves of 96x32); profiles/r02e_kss.log
  // k-step split: an even number of k-tiles and a whole pipeline (fp16: K / 64 tiles, fp16x3: K / 32)
  const int64_t ktiles = ktiles_of(g, nsplit_planes);
  const bool even_kt = ktiles % 2 == 0 && ktiles >= 4;
  if (g.M >= 2048) return g.N >= 3072 ? (even_kt ? 68 : 56) : g.N >= 2048 ? (even_kt ? 69 : 55) : narrow;
  return g.N >= 3072 ? 55 : narrow;
}

// EpiStore configurations the block GEMMs use -> wave-tile epilogues of gemm_pp.h; returns PP_NOT_APPLICABLE when the launch is not
// one of them (the caller then runs the generic kernel)
template <int NSPLIT>
hipError_t try_pp_store(const GemmCore& g, const EpiStore& e, int batch, int variant, hipStream_t s) {
  if (!pp_applies<NSPLIT>(g, batch)) return PP_NOT_APPLICABLE;
  if (variant < 0) variant = pick_pp_variant(g, pp_planes(NSPLIT), false, NSPLIT == 2, e.out16 != nullptr);
  if (variant < 50) return PP_NOT_APPLICABLE;
  constexpr bool PK = NSPLIT != 1;
  constexpr int FMT = NSPLIT == 3 ? 1 : NSPLIT == 2 ? 2 : 0;  // the operand format the consumer of out16 reads = this launch's own
  const bool plain_out = e.alpha == 1.f && e.bias && !e.out2 && !e.zdiv;
  if (plain_out && e.out16 && !e.out32 && !e.res && !e.colscale && !e.rowmask && (e.act == ACT_GELU_TANH || e.act == ACT_NONE) &&
      (PK ? (e.pk16 == FMT && (FMT == 3 ? e.ldo16 >= 3 * (int64_t)g.N / 2 : (e.out16_lo == e.out16 + 32 && e.ldo16 >= 2 * (int64_t)g.N))) : (!e.out16_lo && !e.pk16))) {
    const int64_t ld = PK ? e.ldo16 : (e.ldo16 ? e.ldo16 : e.ldo);
    if ((int64_t)(g.M + 512) * ld * 2 >= (int64_t)0x7ff00000) return PP_NOT_APPLICABLE;
    if constexpr (NSPLIT == 3) {  // microbenchmark ablations of three tiles: variant = 1000 * code + id; code 1 no epilogue, 2 epilogue without stores,
                         // 4 no LDS-DMA in the loop, 8 no MFMAs, 12 neither (fragment reads + barriers + epilogue)
      if (variant >= 1000 && e.act == ACT_GELU_TANH) {
        const PpEpiAct16<1, ACT_GELU_TANH> ep{e.bias, e.out16, ld, g.M, g.N};
        const PpEpiAct16<1, ACT_GELU_TANH, 1> ens{e.bias, e.out16, ld, g.M, g.N};
        const PpEpiAct16<1, ACT_GELU_TANH, 2> eds{e.bias, e.out16, ld, g.M, g.N};
        switch (variant) {
#define F5_ABL(ID)                                                          \
  case 1000 + ID: return launch_pp_one<3, ID, decltype(ep), 1>(g, ep, s);   \
  case 2000 + ID: return launch_pp_one<3, ID, decltype(ens), 0>(g, ens, s); \
  case 3000 + ID: return launch_pp_one<3, ID, decltype(eds), 0>(g, eds, s); \
  case 4000 + ID: return launch_pp_one<3, ID, decltype(ep), 4>(g, ep, s);   \
  case 8000 + ID: return launch_pp_one<3, ID, decltype(ep), 8>(g, ep, s);   \
  case 12000 + ID: return launch_pp_one<3, ID, decltype(ep), 12>(g, ep, s); \
  case 13000 + ID: return launch_pp_one<3, ID, decltype(ep), 13>(g, ep, s);
          F5_ABL(50)
          F5_ABL(56)
          F5_ABL(59)
#undef F5_ABL
          default: return PP_NOT_APPLICABLE;
        }
      }
    }
    if constexpr (NSPLIT == 1 || NSPLIT == 2) {  // ablations of the ping-pong kernel: 1000 * code + 80 (code 1 no epilogue, 4 no LDS-DMA, 8 no MFMAs, 9 neither epilogue nor MFMAs)
      if (variant >= 1000 && variant % 1000 == 80 && e.act == ACT_GELU_TANH) return launch_p8<NSPLIT>(g, PpEpiAct16<FMT, ACT_GELU_TANH>{e.bias, e.out16, ld, g.M, g.N}, variant / 1000, s);
    }
    if (e.act == ACT_GELU_TANH) return launch_pp<NSPLIT>(g, PpEpiAct16<FMT, ACT_GELU_TANH>{e.bias, e.out16, ld, g.M, g.N}, variant, s);
    return launch_pp<NSPLIT>(g, PpEpiAct16<FMT, ACT_NONE>{e.bias, e.out16, ld, g.M, g.N}, variant, s);
  }
  if (plain_out && e.act == ACT_NONE && e.out32 && e.res == e.out32 && e.ldres == e.ldo && !e.out16 && (!e.rowmask || (e.mask_mode == 1 && e.smask == 0))) {
    if ((int64_t)(g.M + 512) * e.ldo * 4 >= (int64_t)0x7ff00000) return PP_NOT_APPLICABLE;
    if (e.colscale) return launch_pp<NSPLIT>(g, PpEpiGateRes<true>{e.bias, e.colscale, e.rowmask, e.out32, e.ldo, g.M, g.N}, variant, s);
    return launch_pp<NSPLIT>(g, PpEpiGateRes<false>{e.bias, nullptr, e.rowmask, e.out32, e.ldo, g.M, g.N}, variant, s);
  }
  return PP_NOT_APPLICABLE;
}

// Tile rasterisation (GemmCore::group_m = row tiles per group, row tile fastest inside a group).  An XCD takes a contiguous run of the tile
// order (block b runs on XCD b % 8; gemm_pp.h), and what its L2 has to pull is the row panels + the weight panels its run touches:
//  * many rounds (M >= 8192): groups of 4 row tiles (+2-5 % at M >= 22k, L2-miss traffic / 2);
//  * the one-round launches of a single utterance with a WIDE output (q|k|v, FF1: N >= 2048): channel tiles fastest gave every XCD 2 row
//    panels and ALL of the weights — groups of 5 make the run of 30 tiles a 5 x 6 block: 118 -> 77 MB fetched per q|k|v launch, 85 -> 64 MB
//    per FF1 launch (FETCH_SIZE, profiles/r04l_groupm_fetch.log), 1-2 % of the launch;
//  * narrow outputs (N = 1024: 8 channel tiles): measured the same at every group size (71-76 MB), plain order.
// F5HIP_GEMM_GROUPM overrides (tuning knob).
int default_group_m(const GemmCore& g) {
  static const int gm_env = [] { const char* v = getenv("F5HIP_GEMM_GROUPM"); return v ? atoi(v) : -1; }();
  if (gm_env >= 0) return gm_env;
  return g.M >= 8192 ? 4 : (g.N >= 2048 ? 5 : 1);
}

template <typename Epi>
hipError_t dispatch(int op, const GemmCore& g0, const Epi& e, int batch, int variant, hipStream_t s) {
  GemmCore g = g0;
  if (g.group_m == 0) g.group_m = default_group_m(g);
  if constexpr (std::is_same<Epi, EpiStore>::value) {
    if (op == OP_F16M) {  // MX lines: the pipelined kernel or nothing (the engine checks the shapes before it chooses the mode)
      const hipError_t r = (variant >= 0 && variant < 50) ? PP_NOT_APPLICABLE : try_pp_store<2>(g, e, batch, variant, s);
      return r == PP_NOT_APPLICABLE ? hipErrorInvalidValue : r;
    }
    if ((variant < 0 || variant >= 50) && (op == OP_F16 || op == OP_F16X3)) {
      const hipError_t r = op == OP_F16 ? try_pp_store<1>(g, e, batch, variant, s) : try_pp_store<3>(g, e, batch, variant, s);
      if (r != PP_NOT_APPLICABLE) return r;
      if (variant >= 50) return hipErrorInvalidValue;  // an explicitly requested pipelined tile that does not take this launch
    }
  }
  if (op == OP_F16M) return hipErrorInvalidValue;
  switch (op) {
    case OP_F32: return launch_tiled<float, 1, Epi>(g, e, batch, variant, s);
    case OP_F16: return launch_tiled<f16, 1, Epi>(g, e, batch, variant, s);
    case OP_F16X3: return launch_tiled<f16, 3, Epi>(g, e, batch, variant, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

hipError_t launch_gemm_store(int op, const GemmCore& g, const EpiStore& e, int batch, hipStream_t s) {
  return dispatch<EpiStore>(op, g, e, batch, -1, s);
}
// OP_F16M launches have no generic-kernel fallback, so the tuning knobs that take a launch away from the tiles instantiated for MX lines
// (a forced tile id outside F5_MX_TILES / 0 = "never the pipelined kernel") make the mode unusable: the
// engine asks here before it chooses MX lines for a call and runs the call in fp16x3 otherwise (f5hip_sample)
bool gemm_mx_tiles_usable() {
  auto mx_tile = [](int id) {
    if (id == 80) return true;
#define F5_IS(ID) if (id == ID) return true;
    F5_MX_TILES(F5_IS)
#undef F5_IS
    return false;
  };
  for (const char* name : {"F5HIP_PP_VARIANT", "F5HIP_PP_VARIANT_N3072", "F5HIP_PP_VARIANT_N2048", "F5HIP_PP_VARIANT_N1024"}) {
    const char* v = getenv(name);
    if (v && atoi(v) >= 0 && !mx_tile(atoi(v))) return false;
  }
  return true;
}
hipError_t launch_gemm_store_variant(int op, const GemmCore& g, const EpiStore& e, int batch, int variant, hipStream_t s) {
  return dispatch<EpiStore>(op, g, e, batch, variant, s);
}
hipError_t launch_gemm_qkv(int op, const GemmCore& g0, const EpiQKV& e0, hipStream_t s) { return launch_gemm_qkv_variant(op, g0, e0, -1, s); }
namespace {
// is tile id instantiated for this operand split?  (launch_pp's switch: 80 = the ping-pong kernel, plain rows and MX lines only; MX lines: the
// F5_MX_TILES set; otherwise 50 .. 70)
bool pp_tile_exists(int nsplit, int id) {
  if (id == 80) return nsplit == 1 || nsplit == 2;
  if (nsplit == 2) {
#define F5_IS(ID) if (id == ID) return true;
    F5_MX_TILES(F5_IS)
#undef F5_IS
    return false;
  }
  return id >= 50 && id <= 70;
}
// Does this q|k|v launch go to a pipelined kernel (half-precision outputs of the flash layouts, dim_head 64, no qk_norm detour), and which?
// e: prepared (epi_qkv_prepare).  Fills the tile and the sizes of the q / k and V^T slabs.
bool qkv_pp_plan(int op, const GemmCore& g0, const EpiQKV& e, int want, GemmCore& g, int& variant, int64_t& qkb, int64_t& vtb) {
  if (!((want < 0 || want >= 50) && e.fast && (op == OP_F16 || op == OP_F16X3 || op == OP_F16M) && e.dh == 64 && e.nseq >= 8 && !e.qk_raw && e.q16 && !e.q32 &&
        (op == OP_F16 ? pp_applies<1>(g0, 1) : pp_applies<3>(g0, 1))))
    return false;
  g = g0;
  if (g.group_m == 0) g.group_m = default_group_m(g);
  variant = want >= 50 ? want : pick_pp_variant(g, op == OP_F16 ? 1 : 2, true, op == OP_F16M, false);
  // sequences the slabs hold: all of the padded rows, or (packed rows) what the caller says — M no longer determines it
  const int64_t sn = e.slab_n ? e.slab_n : e.nseq, bpm = e.rowinfo ? e.nslab : (g.M + e.nseq - 1) / e.nseq;
  qkb = bpm * e.heads * sn * 64 * 2; vtb = bpm * e.heads * 64 * e.ldvt * 2;
  // a tile id forced by a tuning knob that this operand split does not instantiate (e.g. 80 in fp16x3) is "not a pipelined launch": the
  // engine then asks for split remainders instead of P words and the generic kernel takes the call
  return variant >= 50 && pp_tile_exists(op == OP_F16 ? 1 : op == OP_F16M ? 2 : 3, variant) && qkb < (int64_t)0x7ff00000 && vtb < (int64_t)0x7ff00000;
}
}  // namespace
bool gemm_qkv_takes_pp(int op, const GemmCore& g0, const EpiQKV& e0) {
  EpiQKV e = e0;
  epi_qkv_prepare(e, g0.M);
  GemmCore g; int variant; int64_t qkb, vtb;
  return qkv_pp_plan(op, g0, e, -1, g, variant, qkb, vtb);
}
hipError_t launch_gemm_qkv_variant(int op, const GemmCore& g0, const EpiQKV& e0, int want, hipStream_t s) {
  EpiQKV e = e0;
  epi_qkv_prepare(e, g0.M);  // fast = 1 when its preconditions hold (else the general division / 64-bit index path)
  GemmCore g; int variant; int64_t qkb, vtb;
  if (qkv_pp_plan(op, g0, e, want, g, variant, qkb, vtb)) {
    if (e.mx_qk && (!e.q16_lo || !e.k16_lo)) return hipErrorInvalidValue;
    PpEpiQKV p{};
    p.bias = e.bias; p.rope_cs = e.rope_cs;
    p.q16 = e.q16; p.k16 = e.k16; p.vt16 = e.vt16; p.q16_lo = e.q16_lo; p.k16_lo = e.k16_lo; p.vt16_lo = e.vt16_lo;
    p.nseq = e.nseq; p.heads = e.heads; p.pe_heads = e.pe_heads; p.slab_n = e.slab_n; p.pos_off = e.pos_off; p.ldvt = (int)e.ldvt;
    p.qscale = e.qscale; p.nseq_magic = e.nseq_magic; p.nseq_shift = e.nseq_shift; p.inner = e.inner_;
    p.M = g.M; p.N = g.N; p.qk_bytes = (uint32_t)qkb; p.vt_bytes = (uint32_t)vtb; p.rowinfo = e.rowinfo; p.mx_qk = e.mx_qk;
    const hipError_t r = op == OP_F16 ? launch_pp<1>(g, p, variant, s) : op == OP_F16M ? launch_pp<2>(g, p, variant, s) : launch_pp<3>(g, p, variant, s);
    if (r != PP_NOT_APPLICABLE) return r;
  }
  if (e.mx_qk) return hipErrorInvalidValue;  // P words: the pipelined kernels' epilogue only
  if (op == OP_F16M) return hipErrorInvalidValue;  // MX lines: no generic-kernel fallback
  const int gv = want >= 0 && want < 50 ? want : -1;
  if (e.fast) {
    static_assert(sizeof(EpiQKVFast) == sizeof(EpiQKV), "same fields");
    EpiQKVFast f;
    memcpy(&f, &e, sizeof(f));
    return dispatch<EpiQKVFast>(op, g0, f, 1, gv, s);
  }
  return dispatch<EpiQKV>(op, g0, e, 1, gv, s);
}

namespace {
template <int NSPLIT, int ID, typename Epi>
hipError_t set_pp_attr() {
  using C = PpV<ID>;
  constexpr int NSX = pp_ns<NSPLIT, ID>();
  constexpr int lds = gemm_pp_lds_bytes<C::TM, C::TN, C::WGM, C::WGN, NSX, C::KSP>();
  return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pp_kernel<f16, NSPLIT, C::TM, C::TN, C::WGM, C::WGN, NSX, C::JG, Epi, 0, C::KSP, C::KSS>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}
template <int NSPLIT, typename Epi, int... IDS>
hipError_t set_pp_attrs_ids(std::integer_sequence<int, IDS...>) {
  hipError_t e = hipSuccess;
  ((e = e == hipSuccess ? set_pp_attr<NSPLIT, 50 + IDS, Epi>() : e), ...);
  return e;
}
template <typename Epi>
hipError_t set_pp_attrs_mx() {
  hipError_t e = hipSuccess;
#define F5_ATTR(ID) e = e == hipSuccess ? set_pp_attr<2, ID, Epi>() : e;
  F5_MX_TILES(F5_ATTR)
#undef F5_ATTR
  return e;
}
template <int NSPLIT>
hipError_t set_pp_attrs() {  // every tile id 50 .. 70 x every wave-tile epilogue launch_pp can be asked for
  hipError_t e;
  if constexpr (NSPLIT == 2) {
    if ((e = set_pp_attrs_mx<PpEpiAct16<2, ACT_GELU_TANH>>()) != hipSuccess) return e;
    if ((e = set_pp_attrs_mx<PpEpiAct16<2, ACT_NONE>>()) != hipSuccess) return e;
    if ((e = set_pp_attrs_mx<PpEpiGateRes<true>>()) != hipSuccess) return e;
    if ((e = set_pp_attrs_mx<PpEpiGateRes<false>>()) != hipSuccess) return e;
    return set_pp_attrs_mx<PpEpiQKV>();
  } else {
    constexpr auto ids = std::make_integer_sequence<int, 21>{};
    if ((e = set_pp_attrs_ids<NSPLIT, PpEpiAct16<NSPLIT == 3 ? 1 : 0, ACT_GELU_TANH>>(ids)) != hipSuccess) return e;
    if ((e = set_pp_attrs_ids<NSPLIT, PpEpiAct16<NSPLIT == 3 ? 1 : 0, ACT_NONE>>(ids)) != hipSuccess) return e;
    if ((e = set_pp_attrs_ids<NSPLIT, PpEpiGateRes<true>>(ids)) != hipSuccess) return e;
    if ((e = set_pp_attrs_ids<NSPLIT, PpEpiGateRes<false>>(ids)) != hipSuccess) return e;
    return set_pp_attrs_ids<NSPLIT, PpEpiQKV>(ids);
  }
}
}  // namespace

// Dynamic-LDS limits of every kernel of this file, for the CURRENT device: called at context creation (hipFuncSetAttribute is per device
// and must not run inside a stream capture — the first pipelined launch of a sample call is already inside one).
hipError_t init_gemm_kernels() {
  hipError_t e = set_attrs_epi<EpiStore>();
  if (e != hipSuccess) return e;
  e = set_attrs_epi<EpiQKV>();
  if (e != hipSuccess) return e;
  e = set_attrs_epi<EpiQKVFast>();
  if (e != hipSuccess) return e;
#ifndef F5_HIPEMU
  if ((e = set_glds_attrs<EpiStore>()) != hipSuccess || (e = set_glds_attrs<EpiQKV>()) != hipSuccess || (e = set_glds_attrs<EpiQKVFast>()) != hipSuccess) return e;
#endif
  if ((e = init_p8_kernels()) != hipSuccess) return e;
  e = set_pp_attrs<1>();
  if (e != hipSuccess) return e;
  e = set_pp_attrs<2>();
  if (e != hipSuccess) return e;
  return set_pp_attrs<3>();
}
