// qkv_index_check.cpp — EpiQKV's fast index path against its general path (csrc/gemm.h), on the host, store for store.
// Built by tests/test_hipemu.py with the ROCm host clang++ (-DF5_HIPEMU): every (row, 4-channel unit) of a QKV GEMM's output goes
// through EpiQKV::operator() twice — fast = 0 (integer divisions, 64-bit offsets: the path every GPU parity test of round 1 ran) and
// fast = 1 as epi_qkv_prepare() sets it up — into two sets of slabs that must end up byte-identical.  Full-size shapes (the tiny shapes of
// the shim parity tests cannot reach the large offsets), all three output modes, MMDiT slab offsets, partial rope, ragged last sequence.
// Also: the invariant-multiplier division for every m < 2^22 and many divisors, and the refusal of the fast path when an offset could
// exceed 31 bits.
#include <cstdio>
#include <cstring>
#include <vector>

#include "gemm.h"

static int check_shape(const char* name, int M, int nseq, int heads, int dh, int pe_heads, int slab_n, int pos_off, int mode, int qk_raw) {
  // mode 0: fp16 hi only, 1: fp16 hi + lo (q, k, v), 2: fp32 slabs
  const int inner = heads * dh, bp = (M + nseq - 1) / nseq, sn = slab_n ? slab_n : nseq;
  const int64_t ldvt = mode == 2 ? ((sn + 3) & ~3) : ((sn + 7) & ~7);
  const int64_t qk_elems = (int64_t)bp * heads * sn * dh, vt_elems = (int64_t)bp * heads * dh * ldvt;
  std::vector<float> bias(3 * inner), rope((size_t)nseq * dh);
  for (size_t i = 0; i < bias.size(); ++i) bias[i] = 0.001f * (float)(i % 977);
  for (size_t i = 0; i < rope.size(); ++i) rope[i] = 0.5f + 0.0001f * (float)(i % 4099);
  std::vector<std::vector<char>> buf[2];
  EpiQKV e[2];
  for (int path = 0; path < 2; ++path) {
    EpiQKV& x = e[path];
    memset(&x, 0, sizeof(x));
    x.bias = bias.data(); x.rope_cs = rope.data(); x.nseq = nseq; x.heads = heads; x.dh = dh; x.pe_heads = pe_heads; x.qscale = 0.125f;
    x.slab_n = slab_n; x.pos_off = pos_off; x.qk_raw = qk_raw; x.ldvt = ldvt;
    auto mk = [&](int64_t elems, int bytes) { buf[path].emplace_back((size_t)(elems * bytes), (char)0x5a); return buf[path].back().data(); };
    if (mode == 2 || qk_raw) { x.q32 = (float*)mk(qk_elems, 4); x.k32 = (float*)mk(qk_elems, 4); }
    if (mode == 2) x.vt32 = (float*)mk(vt_elems, 4);
    else {
      x.q16 = (f16*)mk(qk_elems, 2); x.k16 = (f16*)mk(qk_elems, 2); x.vt16 = (f16*)mk(vt_elems, 2);
      if (mode == 1) { x.q16_lo = (f16*)mk(qk_elems, 2); x.k16_lo = (f16*)mk(qk_elems, 2); x.vt16_lo = (f16*)mk(vt_elems, 2); }
    }
  }
  epi_qkv_prepare(e[1], M);
  if (!e[1].fast) { printf("FAIL %s: fast path refused\n", name); return 1; }
  EpiQKVFast f;
  static_assert(sizeof(f) == sizeof(e[1]), "same fields");
  memcpy(&f, &e[1], sizeof(f));  // what launch_gemm_qkv does
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < 3 * inner; n += 4) {
      const float s = (float)((m * 31 + n) % 1013) * 0.01f - 5.f;
      e[0](m, n, make_float4(s, s + 0.25f, -s, s * 0.5f), 0);
      f(m, n, make_float4(s, s + 0.25f, -s, s * 0.5f), 0);
    }
  for (size_t b = 0; b < buf[0].size(); ++b)
    if (buf[0][b].size() != buf[1][b].size() || memcmp(buf[0][b].data(), buf[1][b].data(), buf[0][b].size())) {
      printf("FAIL %s: slab %zu differs\n", name, b);
      return 1;
    }
  printf("ok   %s: M=%d nseq=%d heads=%d dh=%d mode=%d (%zu slabs byte-identical)\n", name, M, nseq, heads, dh, mode, buf[0].size());
  return 0;
}

int main(int argc, char** argv) {
  const bool big = argc > 1 && !strcmp(argv[1], "big");
  int bad = 0;
  // invariant-multiplier division: every m < 2^22 (and a band below 2^31) for a spread of divisors
  for (int d : {2, 3, 4, 5, 7, 8, 30, 31, 32, 33, 120, 255, 256, 257, 468, 469, 937, 938, 1023, 1024, 1025, 1406, 1407, 1626, 2812, 4095, 4096, 4097, 65535, 65536, 65537, 1000003}) {
    EpiQKV e;
    memset(&e, 0, sizeof(e));
    e.nseq = d; e.heads = 1; e.dh = 2; e.ldvt = 8;
    epi_qkv_prepare(e, 1);
    if (!e.fast) { printf("FAIL magic: refused for d=%d\n", d); ++bad; continue; }
    auto q = [&](uint32_t m) { return (uint32_t)(((uint64_t)m * e.nseq_magic) >> 32) >> e.nseq_shift; };
    for (uint32_t m = 0; m < (1u << 22); ++m)
      if (q(m) != m / (uint32_t)d) { printf("FAIL magic: d=%d m=%u\n", d, m); ++bad; break; }
    for (uint32_t m = 0x7fffffffu; m > 0x7fffffffu - (1u << 16); --m)
      if (q(m) != m / (uint32_t)d) { printf("FAIL magic: d=%d m=%u\n", d, m); ++bad; break; }
  }
  {  // offsets that do not fit 31 bits: the fast path must refuse
    EpiQKV e;
    memset(&e, 0, sizeof(e));
    e.nseq = 1406; e.heads = 16; e.dh = 64; e.ldvt = 1408;
    epi_qkv_prepare(e, 1406 * 2048);  // 2.9e9 elements per slab
    if (e.fast) { printf("FAIL: fast path accepted a slab of more than 2^31 elements\n"); ++bad; }
    e.dh = 48;                        // not a power of two
    epi_qkv_prepare(e, 1406);
    if (e.fast) { printf("FAIL: fast path accepted dh = 48\n"); ++bad; }
  }
  bad += check_shape("F5-TTS Base B=1 (packed cond+uncond), fp16x3", 2812, 1406, 16, 64, -1, 0, 0, 1, 0);
  bad += check_shape("F5-TTS Base v0 (rope on head 0 only), fp16", 2812, 1406, 16, 64, 1, 0, 0, 0, 0);
  bad += check_shape("one CFG chain, fp32 slabs", 1406, 1406, 16, 64, -1, 0, 0, 2, 0);
  bad += check_shape("E2-TTS Base B=2 (time token: 1407 rows per sequence)", 4 * 1407, 1407, 16, 64, 1, 0, 0, 1, 0);
  bad += check_shape("Small (12 heads), ragged last sequence", 2 * 700 + 333, 700, 12, 64, -1, 0, 0, 1, 0);
  bad += check_shape("MMDiT text stream into joint slabs", 2 * 220, 220, 8, 64, -1, 1406 + 220, 1406, 1, 0);
  bad += check_shape("qk_norm (raw fp32 q / k + fp16 v)", 2812, 1406, 16, 64, -1, 0, 0, 0, 1);
  bad += check_shape("dim_head 32, power-of-two sequence", 4 * 1024, 1024, 8, 32, -1, 0, 0, 1, 0);
  if (big) bad += check_shape("F5-TTS Base B=32 (configs[2]: 89 984 rows), fp16", 64 * 1406, 1406, 16, 64, -1, 0, 0, 0, 0);
  printf(bad ? "FAILED (%d)\n" : "all ok\n", bad);
  return bad ? 1 : 0;
}
