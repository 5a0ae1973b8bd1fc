// gemm.hip — instantiations and launch heuristics of the MFMA GEMM (gemm.h)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "kernels.h"
#ifndef F5_HIPEMU
#include "gemm_sk.h"  // inline asm: not part of the host-shim build (tests/hipemu)
#endif
#include "gemm_skrs.h"

namespace {

// One tile variant: TM x TN 32x32 tiles per wave, WGM x WGN waves.
template <int TM_, int TN_, int WGM_, int WGN_>
struct V {
  static constexpr int TM = TM_, TN = TN_, WGM = WGM_, WGN = WGN_;
  static constexpr int BM = 32 * WGM * TM, BN = 32 * WGN * TN;
};

// variant ids (rows x output channels): 0 = 64x128, 1 = 128x64, 2 = 128x128 (4 waves);
//                                       3 = 256x128, 4 = 128x256 (8 waves), 5 = 256x256 (16 waves)
// every mode stages one 128-byte line per operand row per k-tile, so LDS per workgroup is the same in all modes
template <typename T, int NSPLIT, int ID>
struct Variant;
#define F5_VARIANT(ID, TM, TN, WGM, WGN)                                                    \
  template <typename T, int NSPLIT>                                                         \
  struct Variant<T, NSPLIT, ID> : V<TM, TN, WGM, WGN> {}
F5_VARIANT(0, 1, 2, 2, 2);
F5_VARIANT(1, 2, 1, 2, 2);
F5_VARIANT(2, 2, 2, 2, 2);
F5_VARIANT(3, 2, 2, 4, 2);
F5_VARIANT(4, 2, 2, 2, 4);
F5_VARIANT(5, 2, 2, 4, 4);
F5_VARIANT(8, 1, 1, 2, 2);  // 64x64
F5_VARIANT(10, 2, 3, 2, 2);  // 128x192
#undef F5_VARIANT

template <typename T, int NSPLIT, int ID, typename Epi>
hipError_t set_attr() {
  using C = Variant<T, NSPLIT, ID>;
  constexpr int lds = gemm_lds_bytes<T, NSPLIT, C::TM, C::TN, C::WGM, C::WGN>();
  if (lds > 160 * 1024) return hipSuccess;  // variant does not exist in this mode (launch_one rejects it)
  return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<T, NSPLIT, C::TM, C::TN, Epi, C::WGM, C::WGN>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

template <typename T, int NSPLIT, typename Epi>
hipError_t set_attrs_op() {
  hipError_t e;
  if ((e = set_attr<T, NSPLIT, 0, Epi>()) != hipSuccess) return e;
  if ((e = set_attr<T, NSPLIT, 1, Epi>()) != hipSuccess) return e;
  if ((e = set_attr<T, NSPLIT, 2, Epi>()) != hipSuccess) return e;
  if ((e = set_attr<T, NSPLIT, 3, Epi>()) != hipSuccess) return e;
  if ((e = set_attr<T, NSPLIT, 4, Epi>()) != hipSuccess) return e;
  if ((e = set_attr<T, NSPLIT, 5, Epi>()) != hipSuccess) return e;
  if ((e = set_attr<T, NSPLIT, 8, Epi>()) != hipSuccess) return e;
  if ((e = set_attr<T, NSPLIT, 10, Epi>()) != hipSuccess) return e;
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
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  // operands are addressed with 32-bit byte offsets through buffer descriptors; offsets >= 2 GiB mean "no such row"
  if ((int64_t)g.a_rows * g.lda * (int64_t)sizeof(T) >= (int64_t)0x7ff00000 || (int64_t)g.w_rows * g.ldw * (int64_t)sizeof(T) >= (int64_t)0x7ff00000)
    return hipErrorInvalidValue;
  dim3 grid(((g.M + C::BM - 1) / C::BM) * ((g.N + C::BN - 1) / C::BN), 1, batch);
  hipLaunchKernelGGL(kern, grid, dim3(64 * C::WGM * C::WGN), lds, s, g, e);
  return hipGetLastError();
}

int pick_variant(const GemmCore& g, int batch) {
  static const int forced = [] { const char* e = getenv("F5HIP_GEMM_VARIANT"); return e ? atoi(e) : -1; }();  // tuning / test knobs
  if (forced >= 0) return forced;
  static const int f3072 = [] { const char* e = getenv("F5HIP_GEMM_VARIANT_N3072"); return e ? atoi(e) : -1; }();
  static const int f2048 = [] { const char* e = getenv("F5HIP_GEMM_VARIANT_N2048"); return e ? atoi(e) : -1; }();
  static const int f1024 = [] { const char* e = getenv("F5HIP_GEMM_VARIANT_N1024"); return e ? atoi(e) : -1; }();
  if (g.M > 256 && g.N == 3072 && f3072 >= 0) return f3072;
  if (g.M > 256 && g.N == 2048 && f2048 >= 0) return f2048;
  if (g.M > 256 && g.N == 1024 && f1024 >= 0) return f1024;
  if (g.M <= 64) return 0;
  // 128x64 tiles (3 workgroups per CU) until the grid is several waves deep, then 128x128 (higher FLOP per byte staged):
  // measured crossover between M = 2812 (B=1: 128x64 wins on all four block GEMMs) and M = 22496 (B=8: 128x128 wins).
  const int64_t big = (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128) * batch;
  // many waves of tiles: the 256x256 / 8-wave LDS-DMA tile (128x64 per wave) stages and reads the fewest LDS bytes per MFMA —
  // the LDS pipe, not the matrix pipe, is what the 128x128 tile saturates first (+10-16 % at M >= 22k in both fp16 modes)
  if (big >= 1024 && g.M >= 65536 && g.N >= 1024 && batch == 1 && g.K % 32 == 0) return 21;
  // a few thousand to a few ten thousand rows (B = 2..16): 128x256 with 8 waves of 64x64 and a 3-stage ring (one workgroup per CU, two
  // waves per SIMD sharing one LDS tile: 0.9 KB of LDS traffic per MFMA against 1.5 KB for two independent 4-wave 128x64 tiles) —
  // 10-17 % faster than both the 256x256 tile (too few tiles, one k-tile in flight) and the 128x128 / 128x64 tiles at M = 11k-22k
  if (g.M >= 8192 && g.N >= 1024 && batch == 1 && g.K % 32 == 0) return 31;  // in situ: -6 % end to end at B = 4 and 8, nothing at B = 2
  if (big >= 1024) return 2;
  // small grids: the direct-to-LDS ring (variant 6) wins where the tile count is lowest (N <= 1024: out-projection, FF2: -10 %),
  // the register-staged kernel elsewhere (tools/kernel_bench.py, B=1)
  return (g.N <= 1024 && g.M > 256 && batch == 1) ? 6 : 1;
}

#ifndef F5_HIPEMU
// direct-to-LDS ring variants (variant ids 6 = 128x64, 7 = 128x128, 3-stage ring)
template <typename T, int NSPLIT, int TM, int TN, typename Epi, int WGM = 2, int WGN = 2, int NS = 3, int PRIO = 0>
hipError_t launch_glds(const GemmCore& g, const Epi& e, int batch, hipStream_t s) {
  constexpr int lds = gemm_glds_lds_bytes<T, NSPLIT, TM, TN, WGM, WGN, NS>();
  auto kern = gemm_glds_kernel<T, NSPLIT, TM, TN, Epi, WGM, WGN, NS, PRIO>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (err != hipSuccess) return err;
    attr_done = true;
  }
  if ((int64_t)g.a_rows * g.lda * (int64_t)sizeof(T) >= (int64_t)0x7ff00000 || (int64_t)g.w_rows * g.ldw * (int64_t)sizeof(T) >= (int64_t)0x7ff00000)
    return hipErrorInvalidValue;
  constexpr int BM = 32 * WGM * TM, BN = 32 * WGN * TN;
  dim3 grid(((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN), 1, batch);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WGM * WGN), lds, s, g, e);
  return hipGetLastError();
}

// stream-K launch (gemm_sk.h): 8 waves of 64x64, 256x128 (ROWS256) or 128x256 tile, grid = g.sk_grid resident workgroups
constexpr int64_t SK_SLOT_BYTES = 512 * 4 * 16 * 4;  // 8 waves x 64 lanes x (2x2 tiles x 16 regs) fp32
template <typename T, int NSPLIT, typename Epi, bool ROWS256>
hipError_t launch_sk(const GemmCore& g, const Epi& e, hipStream_t s) {
  constexpr int WGM = ROWS256 ? 4 : 2, WGN = ROWS256 ? 2 : 4, BM = 64 * WGM, BN = 64 * WGN;
  constexpr int lds = gemm_glds_lds_bytes<T, NSPLIT, 2, 2, WGM, WGN, 3>();
  auto kern = gemm_sk_kernel<T, NSPLIT, 2, 2, Epi, WGM, WGN>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (err != hipSuccess) return err;
    attr_done = true;
  }
  if (!g.sk_ws || g.sk_grid < 8 || (g.sk_grid & 7)) return hipErrorInvalidValue;
  if ((int64_t)g.a_rows * g.lda * (int64_t)sizeof(T) >= (int64_t)0x7ff00000 || (int64_t)g.w_rows * g.ldw * (int64_t)sizeof(T) >= (int64_t)0x7ff00000)
    return hipErrorInvalidValue;
  SkArgs sk{};
  sk.ws = reinterpret_cast<float*>(g.sk_ws);
  sk.flags = reinterpret_cast<int*>(reinterpret_cast<char*>(g.sk_ws) + (int64_t)g.sk_grid * SK_SLOT_BYTES);
  sk.err = sk.flags + g.sk_grid;
  static const bool dbg_env = getenv("F5HIP_SK_DEBUG") != nullptr;  // microbenchmark only: the caller sized the workspace for the stamps
  sk.dbg = dbg_env ? reinterpret_cast<long long*>(sk.err + 2) : nullptr;
  sk.tiles_n = (g.N + BN - 1) / BN;
  sk.tiles = ((g.M + BM - 1) / BM) * sk.tiles_n;
  const int kbytes = g.K * (int)sizeof(T) * (NSPLIT == 3 ? 2 : 1);
  sk.kt = (kbytes + GEMM_KTB - 1) / GEMM_KTB;
  hipLaunchKernelGGL(kern, dim3(g.sk_grid), dim3(512), lds, s, g, e, sk);
  return hipGetLastError();
}

#else
constexpr int64_t SK_SLOT_BYTES = 512 * 4 * 16 * 4;
#endif  // F5_HIPEMU

// stream-K with the reduce-scattered epilogue (gemm_skrs.h).  Workspace (caller-owned, private to one stream, flags zeroed once):
// [grid][2] slots of 128 KB, then [grid][2] int flags, then the error word.
template <typename T, int NSPLIT, typename Epi, bool ROWS256>
hipError_t launch_skrs(const GemmCore& g, const Epi& e, hipStream_t s) {
  constexpr int WGM = ROWS256 ? 4 : 2, WGN = ROWS256 ? 2 : 4, BM = 64 * WGM, BN = 64 * WGN;
  constexpr int lds = 3 * (BM + BN) * GEMM_KTB;  // the 3-stage ring of the LDS-DMA variants
  auto kern = gemm_skrs_kernel<T, NSPLIT, Epi, WGM, WGN>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (err != hipSuccess) return err;
    attr_done = true;
  }
  if (!g.sk_ws || g.sk_grid < 8 || (g.sk_grid & 7)) return hipErrorInvalidValue;
  // every workgroup must be able to be resident (one 144 KB workgroup per CU): never more workgroups than the device has CUs
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return n;
  }();
  if (g.sk_grid > cus) return hipErrorInvalidValue;
  if ((int64_t)g.a_rows * g.lda * (int64_t)sizeof(T) >= (int64_t)0x7ff00000 || (int64_t)g.w_rows * g.ldw * (int64_t)sizeof(T) >= (int64_t)0x7ff00000)
    return hipErrorInvalidValue;
  SkrsArgs sk{};
  sk.ws = reinterpret_cast<float*>(g.sk_ws);
  sk.flags = reinterpret_cast<int*>(reinterpret_cast<char*>(g.sk_ws) + (int64_t)g.sk_grid * 2 * SK_SLOT_BYTES);
  sk.err = sk.flags + 2 * g.sk_grid;
  sk.tiles_n = (g.N + BN - 1) / BN;
  sk.tiles = ((g.M + BM - 1) / BM) * sk.tiles_n;
  const int kbytes = g.K * (int)sizeof(T) * (NSPLIT == 3 ? 2 : 1);
  sk.kt = (kbytes + GEMM_KTB - 1) / GEMM_KTB;
  // every share must be non-empty and at least an eighth of a tile long: a tile then has at most 9 contributors (16 units to deal out)
  const int64_t min_class_iters = (int64_t)(sk.tiles / 8) * sk.kt, gx = g.sk_grid >> 3;
  if (min_class_iters < gx || min_class_iters / gx < (sk.kt + 7) / 8) return hipErrorInvalidValue;
#ifdef F5_HIPEMU  // all workgroups alive at once (they talk through flags)
  if (getenv("F5HIP_SK_TRACE")) fprintf(stderr, "skrs M=%d N=%d K=%d grid=%d\n", g.M, g.N, g.K, g.sk_grid);
  hipemu::launch_coop(dim3(g.sk_grid), dim3(512), lds, [=] { kern(g, e, sk); });
#else
  hipLaunchKernelGGL(kern, dim3(g.sk_grid), dim3(512), lds, s, g, e, sk);
#endif
  return hipGetLastError();
}

// microbenchmark ablations of the 128x128 variant (variant id 8 + ABL); EpiStore only
template <typename T, int NSPLIT, int ABL, typename Epi, int VID = 2>
hipError_t launch_abl(const GemmCore& g, const Epi& e, int batch, hipStream_t s) {
  using C = Variant<T, NSPLIT, VID>;
  constexpr int lds = gemm_lds_bytes<T, NSPLIT, C::TM, C::TN, C::WGM, C::WGN>();
  auto kern = gemm_kernel<T, NSPLIT, C::TM, C::TN, Epi, C::WGM, C::WGN, ABL>;
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (err != hipSuccess) return err;
  dim3 grid(((g.M + C::BM - 1) / C::BM) * ((g.N + C::BN - 1) / C::BN), 1, batch);
  hipLaunchKernelGGL(kern, grid, dim3(64 * C::WGM * C::WGN), lds, s, g, e);
  return hipGetLastError();
}

template <typename T, int NSPLIT, typename Epi>
hipError_t launch_tiled(const GemmCore& g, const Epi& e, int batch, int variant, hipStream_t s) {
  if (variant < 0) variant = pick_variant(g, batch);
  switch (variant) {
    case 0: return launch_one<T, NSPLIT, 0, Epi>(g, e, batch, s);
    case 1: return launch_one<T, NSPLIT, 1, Epi>(g, e, batch, s);
    case 2: return launch_one<T, NSPLIT, 2, Epi>(g, e, batch, s);
    case 3: return launch_one<T, NSPLIT, 3, Epi>(g, e, batch, s);
    case 4: return launch_one<T, NSPLIT, 4, Epi>(g, e, batch, s);
    case 5: return launch_one<T, NSPLIT, 5, Epi>(g, e, batch, s);
#ifndef F5_HIPEMU
    case 6: return launch_glds<T, NSPLIT, 2, 1, Epi>(g, e, batch, s);
    case 7: return launch_glds<T, NSPLIT, 2, 2, Epi>(g, e, batch, s);
    case 13: return launch_glds<T, NSPLIT, 4, 2, Epi>(g, e, batch, s);  // 256x128, 4 waves of 128x64
    case 14: return launch_glds<T, NSPLIT, 2, 4, Epi>(g, e, batch, s);  // 128x256, 4 waves of 64x128
    case 21: return launch_glds<T, NSPLIT, 4, 2, Epi, 2, 4, 2>(g, e, batch, s);  // 256x256, 8 waves of 128x64, 2-stage ring
    case 22: return launch_glds<T, NSPLIT, 2, 4, Epi, 4, 2, 2>(g, e, batch, s);  // 256x256, 8 waves of 64x128, 2-stage ring
    // wave tile 64x64 (2/3 of the LDS fragment bytes per MFMA of the 64x32 wave tile of variant 6) at unchanged workgroup tile counts
    case 26: return launch_glds<T, NSPLIT, 2, 2, Epi, 2, 1, 3>(g, e, batch, s);  // 128x64, 2 waves of 64x64, 3-stage ring (72 KB: 2 WG/CU)
    case 27: return launch_glds<T, NSPLIT, 2, 2, Epi, 2, 1, 2>(g, e, batch, s);  // 128x64, 2 waves of 64x64, 2-stage ring (48 KB: 3 WG/CU)
    case 28: return launch_glds<T, NSPLIT, 2, 2, Epi, 2, 2, 2>(g, e, batch, s);  // 128x128, 4 waves of 64x64, 2-stage ring (64 KB: 2 WG/CU)
    case 29: return launch_glds<T, NSPLIT, 2, 2, Epi, 1, 2, 3>(g, e, batch, s);  // 64x128, 2 waves of 64x64, 3-stage ring
    case 30: return launch_glds<T, NSPLIT, 2, 2, Epi, 4, 2, 3>(g, e, batch, s);  // 256x128, 8 waves of 64x64, 3-stage ring (144 KB, 1 WG/CU)
    case 31: return launch_glds<T, NSPLIT, 2, 2, Epi, 2, 4, 3>(g, e, batch, s);  // 128x256, 8 waves of 64x64, 3-stage ring
    case 40: return launch_sk<T, NSPLIT, Epi, true>(g, e, s);   // stream-K, 256x128 tiles
    case 41: return launch_sk<T, NSPLIT, Epi, false>(g, e, s);  // stream-K, 128x256 tiles
#else  // the host shim runs the register-staged kernel at the LDS-DMA variants' call sites (same tile or the nearest one)
    case 6: case 26: case 27: case 24: return launch_one<T, NSPLIT, 1, Epi>(g, e, batch, s);
    case 7: case 13: case 14: case 21: case 22: case 23: case 25: case 28: case 29: case 30: case 31: return launch_one<T, NSPLIT, 2, Epi>(g, e, batch, s);
#endif
    case 42: return launch_skrs<T, NSPLIT, Epi, true>(g, e, s);   // stream-K, reduce-scattered epilogue, 256x128 tiles
    case 43: return launch_skrs<T, NSPLIT, Epi, false>(g, e, s);  // stream-K, reduce-scattered epilogue, 128x256 tiles
#ifndef F5_HIPEMU
    case 24: return launch_glds<T, NSPLIT, 2, 1, Epi, 2, 2, 3, 2>(g, e, batch, s);     // ablation: variant 6 with 2 of the 3 fp16x3 products
    case 25: return launch_glds<T, NSPLIT, 4, 2, Epi, 2, 4, 2, 2>(g, e, batch, s);     // ablation: variant 21 with 2 of the 3 fp16x3 products
    case 23: return launch_glds<T, NSPLIT, 4, 2, Epi, 2, 4, 2, 1>(g, e, batch, s);  // variant 21 + s_setprio around the MFMA clusters
#endif
    case 8: return launch_one<T, NSPLIT, 8, Epi>(g, e, batch, s);
    case 10: return launch_one<T, NSPLIT, 10, Epi>(g, e, batch, s);
    default: break;
  }
  if constexpr (std::is_same<Epi, EpiStore>::value && !std::is_same<T, float>::value) {
    switch (variant) {
      case 9: return launch_abl<T, NSPLIT, 1, Epi>(g, e, batch, s);
      case 10: return launch_abl<T, NSPLIT, 2, Epi>(g, e, batch, s);
      case 11: return launch_abl<T, NSPLIT, 3, Epi>(g, e, batch, s);
      case 12: return launch_abl<T, NSPLIT, 4, Epi>(g, e, batch, s);
      case 15: return launch_abl<T, NSPLIT, 7, Epi>(g, e, batch, s);
      case 17: return launch_abl<T, NSPLIT, 8, Epi, 1>(g, e, batch, s);   // 128x64: no epilogue
      case 18: return launch_abl<T, NSPLIT, 4, Epi, 1>(g, e, batch, s);   // 128x64: no MFMA
      case 19: return launch_abl<T, NSPLIT, 15, Epi, 1>(g, e, batch, s);  // 128x64: prologue + reads + barriers only
      case 20: return launch_abl<T, NSPLIT, 3, Epi, 1>(g, e, batch, s);   // 128x64: no loads / LDS stores in the loop
      default: break;
    }
  }
  return hipErrorInvalidValue;
  switch (0) {
    default: return hipErrorInvalidValue;
  }
}

template <typename Epi>
hipError_t dispatch(int op, const GemmCore& g0, const Epi& e, int batch, int variant, hipStream_t s) {
  GemmCore g = g0;
  static const int gm_env = [] { const char* v = getenv("F5HIP_GEMM_GROUPM"); return v ? atoi(v) : -1; }();  // tuning knob
  // default: groups of 4 row-tiles once the grid is many waves deep (+2-5 % at M >= 22k, L2-miss traffic / 2), plain order otherwise
  if (g.group_m == 0) g.group_m = gm_env >= 0 ? gm_env : (g.M >= 8192 ? 4 : 1);
  // stream-K on request of the caller (GemmCore.sk_variant + workspace): only where the schedule applies (launch_skrs checks the share
  // sizes), otherwise the plain heuristic
  if (variant < 0 && g.sk_ws && g.sk_variant && batch == 1 && g.M > 256 && op != OP_F32) {  // launch_skrs decides whether the shape suits the grid
    hipError_t r = hipErrorInvalidValue;
    bool done = false;
    if constexpr (std::is_same<Epi, EpiStore>::value) {  // the two DiT shapes have branch-free epilogues (gemm.h EpiFF1 / EpiGateRes)
      static const bool generic = getenv("F5HIP_SK_GENERIC_EPI") != nullptr;  // A/B switch
      const bool rows256 = g.sk_variant == 42;
      if (!generic && op == OP_F16X3 && e.act == ACT_GELU_TANH && e.alpha == 1.f && e.bias && e.out16 && e.out16_lo == e.out16 + 32 && e.pk16 && e.ldo16 &&
          !e.out32 && !e.res && !e.colscale && !e.rowmask && !e.out2 && !e.zdiv) {
        const EpiFF1 f{e.bias, e.out16, e.ldo16};
        r = rows256 ? launch_skrs<f16, 3, EpiFF1, true>(g, f, s) : launch_skrs<f16, 3, EpiFF1, false>(g, f, s);
        done = true;
      } else if (!generic && e.act == ACT_NONE && e.alpha == 1.f && e.bias && e.colscale && e.out32 && e.res == e.out32 && e.ldres == e.ldo && !e.out16 &&
                 !e.out2 && !e.zdiv && (!e.rowmask || (e.mask_mode == 1 && e.smask == 0))) {
        const EpiGateRes f{e.bias, e.colscale, e.rowmask, e.out32, e.ldo};
        if (op == OP_F16) r = rows256 ? launch_skrs<f16, 1, EpiGateRes, true>(g, f, s) : launch_skrs<f16, 1, EpiGateRes, false>(g, f, s);
        else r = rows256 ? launch_skrs<f16, 3, EpiGateRes, true>(g, f, s) : launch_skrs<f16, 3, EpiGateRes, false>(g, f, s);
        done = true;
      }
    }
    if (!done) r = op == OP_F16 ? launch_tiled<f16, 1, Epi>(g, e, batch, g.sk_variant, s) : launch_tiled<f16, 3, Epi>(g, e, batch, g.sk_variant, s);
    if (r != hipErrorInvalidValue) return r;
  }
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
hipError_t launch_gemm_store_variant(int op, const GemmCore& g, const EpiStore& e, int batch, int variant, hipStream_t s) {
  return dispatch<EpiStore>(op, g, e, batch, variant, s);
}
hipError_t launch_gemm_qkv(int op, const GemmCore& g, const EpiQKV& e0, hipStream_t s) {
  static const bool generic = getenv("F5HIP_QKV_EPI_GENERIC") != nullptr;  // A/B switch: the general (division / 64-bit) index path
  EpiQKV e = e0;
  e.fast = 0;
  if (!generic) epi_qkv_prepare(e, g.M);  // fast = 1 when its preconditions hold
  if (e.fast) {
    static_assert(sizeof(EpiQKVFast) == sizeof(EpiQKV), "same fields");
    EpiQKVFast f;
    memcpy(&f, &e, sizeof(f));
    return dispatch<EpiQKVFast>(op, g, f, 1, -1, s);
  }
  return dispatch<EpiQKV>(op, g, e, 1, -1, s);
}

hipError_t init_gemm_kernels() {
  hipError_t e = set_attrs_epi<EpiStore>();
  if (e != hipSuccess) return e;
  e = set_attrs_epi<EpiQKV>();
  if (e != hipSuccess) return e;
  return set_attrs_epi<EpiQKVFast>();
}
