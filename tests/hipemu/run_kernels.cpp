// run_kernels.cpp — runs kernels of f5-tts_amd/csrc on the CPU through hipemu.h (TEST INFRASTRUCTURE, driven by tests/test_hipemu.py).
// Tensors travel as raw little-endian files in the directory given as argv[1]; argv[2] is the mode, the rest its integer arguments.
#include "hipemu.h"
// kernel sources, exactly as hipcc sees them
#include "bigvgan_kernels.h"
#include "attention_kernel.h"
#include "conv_gemm.h"

#include <string>

static std::string g_dir;
template <typename T>
static std::vector<T> rd(const char* name, bool optional = false) {
  std::vector<T> v;
  FILE* f = fopen((g_dir + "/" + name).c_str(), "rb");
  if (!f) {
    if (optional) return v;
    fprintf(stderr, "missing %s\n", name);
    exit(2);
  }
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  v.resize(n / sizeof(T));
  if (fread(v.data(), 1, n, f) != (size_t)n) exit(2);
  fclose(f);
  return v;
}
template <typename T>
static void wr(const char* name, const std::vector<T>& v) {
  FILE* f = fopen((g_dir + "/" + name).c_str(), "wb");
  fwrite(v.data(), sizeof(T), v.size(), f);
  fclose(f);
}

template <typename T, int NSPLIT, int TM, int TN>
static void run_gemm(bool conv, GemmCore g, ConvTaps tp, EpiStore e, int batch) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  const int lds = gemm_lds_bytes<T, NSPLIT, TM, TN, 2, 2>();
  dim3 grid(((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN), 1, batch);
  if (conv) hipemu::launch(grid, dim3(256), lds, [&] { conv_gemm_kernel<T, NSPLIT, TM, TN, EpiStore, 2, 2>(g, tp, e); });
  else hipemu::launch(grid, dim3(256), lds, [&] { gemm_kernel<T, NSPLIT, TM, TN, EpiStore, 2, 2>(g, e); });
}
template <typename T, int NSPLIT>
static void run_gemm_tile(bool conv, int tile, GemmCore g, ConvTaps tp, EpiStore e, int batch) {
  if (tile == 1) run_gemm<T, NSPLIT, 2, 1>(conv, g, tp, e, batch);
  else run_gemm<T, NSPLIT, 2, 2>(conv, g, tp, e, batch);
}

int main(int argc, char** argv) {
  if (argc < 3) return 1;
  g_dir = argv[1];
  const std::string mode = argv[2];
  auto A = [&](int i) { return atoi(argv[3 + i]); };
  if (mode == "aa") {  // B L C logscale op(-1 = fp32 [B,L,C] out) cpad
    const int B = A(0), L = A(1), C = A(2), ls = A(3), op = A(4), cpad = A(5);
    auto x = rd<float>("x.bin"), al = rd<float>("alpha.bin"), be = rd<float>("beta.bin"), f = rd<float>("f.bin");
    std::vector<float> y((size_t)B * L * C, -777.f);
    std::vector<char> oper(op >= 0 ? (size_t)B * L * cpad * (op == OP_F16 ? 2 : 4) : 0, (char)0x5a);
    AaArgs a{};
    a.x = x.data(); a.y = y.data(); a.alpha = al.data(); a.beta = be.data(); a.L = L; a.C = C; a.logscale = ls;
    a.oper = op >= 0 ? oper.data() : nullptr; a.op = op >= 0 ? op : OP_F32; a.cpad = op >= 0 ? cpad : C;
    for (int j = 0; j < 12; ++j) a.f[j] = f[j];
    constexpr int TL = 32;
    const int chunks = (L + TL - 1) / TL;
    dim3 grid(((op >= 0 ? cpad : C) + 63) / 64, (chunks + 3) / 4, B);
    hipemu::launch(grid, dim3(64, 4), 0, [&] { aa_snake_kernel<TL>(a); });
    if (op >= 0) wr("out.bin", oper); else wr("out.bin", y);
  } else if (mode == "im2col") {  // B L C ntaps shift0 dstep cpad op sb sl sc
    const int B = A(0), L = A(1), C = A(2), nt = A(3), s0 = A(4), ds = A(5), cpad = A(6), op = A(7);
    auto src = rd<float>("src.bin");
    const int64_t K = (int64_t)nt * cpad, ld = K * (op == OP_F16X3 ? 2 : 1);
    std::vector<char> out((size_t)B * L * ld * (op == OP_F32 ? 4 : 2), (char)0x5a);
    ColArgs a{};
    a.src = src.data(); a.sb = A(8); a.sl = A(9); a.sc = A(10); a.L = L; a.C = C; a.ntaps = nt; a.shift0 = s0; a.dstep = ds; a.cpad = cpad;
    a.op = op; a.out = out.data(); a.ldo = ld; a.ob = (int64_t)L * ld;
    const int64_t n = (int64_t)L * (nt * cpad / 4);
    hipemu::launch(dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, [&] { im2col_kernel(a); });
    wr("out.bin", out);
  } else if (mode == "mean") {  // nk n
    const int nk = A(0);
    const int64_t n = A(1);
    std::vector<float> r[4], out(n);
    for (int j = 0; j < nk; ++j) r[j] = rd<float>(("r" + std::to_string(j) + ".bin").c_str());
    hipemu::launch(dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, [&] {
      mean_kernel(r[0].data(), nk > 1 ? r[1].data() : nullptr, nk > 2 ? r[2].data() : nullptr, nk > 3 ? r[3].data() : nullptr, nk, (float)nk, n / 4, out.data());
    });
    wr("out.bin", out);
  } else if (mode == "post") {  // B L C use_tanh has_bias
    const int B = A(0), L = A(1), C = A(2), ut = A(3), hb = A(4);
    auto y = rd<float>("y.bin"), w7 = rd<float>("w7.bin"), bias = rd<float>("bias.bin", true);
    std::vector<float> out((size_t)B * L);
    hipemu::launch(dim3((L + 255) / 256, B), dim3(256), 7 * C * sizeof(float),
                   [&] { conv_post_kernel(y.data(), w7.data(), hb ? bias.data() : nullptr, L, C, ut, out.data()); });
    wr("out.bin", out);
  } else if (mode == "gemm" || mode == "conv") {  // op M N K batch tile has_res [ntaps shift0 dstep cpad]
    const bool conv = mode == "conv";
    const int op = A(0), M = A(1), N = A(2), K = A(3), batch = A(4), tile = A(5), has_res = A(6);
    auto Ab = rd<char>("A.bin"), Wb = rd<char>("W.bin");
    auto bias = rd<float>("bias.bin"), res = rd<float>("res.bin", true);
    std::vector<float> out((size_t)batch * M * N, -777.f);
    const int mul = op == OP_F16X3 ? 2 : 1;
    GemmCore g{};
    ConvTaps tp{};
    g.A = Ab.data(); g.W = Wb.data();
    g.ldw = (int64_t)K * mul; g.strideW = 0; g.M = M; g.N = N; g.K = K; g.a_rows = M; g.w_rows = N; g.group_m = 1;
    if (conv) {
      const int cpad = A(10);
      tp = ConvTaps{A(7), A(8), A(9), cpad * (op == OP_F16 ? 2 : 4) / GEMM_KTB};
      g.lda = (int64_t)cpad * mul;
    } else {
      g.lda = (int64_t)K * mul;
    }
    g.strideA = (int64_t)M * g.lda;
    EpiStore e{};
    e.alpha = 1.f; e.act = ACT_NONE; e.bias = bias.data(); e.out32 = out.data(); e.ldo = N; e.ldres = N; e.res = has_res ? res.data() : nullptr;
    e.zdiv = 1; e.so1 = (int64_t)M * N; e.so2 = 0;
    switch (op) {
      case OP_F32: run_gemm_tile<float, 1>(conv, tile, g, tp, e, batch); break;
      case OP_F16: run_gemm_tile<f16, 1>(conv, tile, g, tp, e, batch); break;
      default: run_gemm_tile<f16, 3>(conv, tile, g, tp, e, batch); break;
    }
    wr("out.bin", out);
  } else if (mode == "attn") {  // nsplit(1|2|3; 4 = MX-corrected scores) Bp heads n kv_split o_packed has_kvlen [pipe: 0 | 4 | 6 waves, +10 = row sums on the VALU; with nsplit 4: 40 / 60 = the lazy 4- / 6-wave forms, 42 / 43 / 62 = V / V and P as hi + lo halves]
    const int nsplit = A(0), Bp = A(1), heads = A(2), n = A(3), kvs = A(4), o_packed = A(5), has_kvlen = A(6), pipe = argc > 10 ? A(7) : 0;
    const int bh = Bp * heads, ldv = (n + 7) & ~7;
    auto q = rd<f16>("q.bin"), ql = rd<f16>("q_lo.bin", true), k = rd<f16>("k.bin"), kl = rd<f16>("k_lo.bin", true);
    auto vt = rd<f16>("vt.bin"), vtl = rd<f16>("vt_lo.bin", true);
    auto kvlen = rd<int32_t>("kvlen.bin", true);
    const int64_t ow = (int64_t)heads * 64 * (o_packed ? 2 : 1);
    std::vector<f16> o((size_t)Bp * n * ow, (f16)-7.f), olo(o_packed ? 0 : o.size(), (f16)-7.f);
    std::vector<float> part_o((size_t)bh * n * kvs * 64, -777.f), part_ml((size_t)bh * n * kvs * 2, -777.f);
    FlashArgs a{};
    a.q = q.data(); a.q_lo = ql.empty() ? nullptr : ql.data(); a.k = k.data(); a.k_lo = kl.empty() ? nullptr : kl.data();
    a.vt = vt.data(); a.vt_lo = vtl.empty() ? nullptr : vtl.data();
    a.o = o.data(); a.o_lo = o_packed ? o.data() + 32 : olo.data();
    a.kvlen = has_kvlen ? kvlen.data() : nullptr;
    a.n = n; a.ldv = ldv; a.heads = heads; a.o_packed = o_packed;
    a.kv_split = kvs; a.part_o = part_o.data(); a.part_ml = part_ml.data();
    a.nqb = (n + QB - 1) / QB; a.nwg = bh * a.nqb * (kvs > 1 ? kvs : 1);
    auto go = [&](auto ns, auto pv) {
      constexpr int NS = decltype(ns)::value, PV = decltype(pv)::value;
      const int lds = flash_lds_bytes<NS, PV>();
      if (kvs > 1) {
        hipemu::launch(dim3(a.nwg), dim3(256), lds, [&] { flash_attn_kernel<NS, PV, 4, true>(a); });
        const int64_t rows = (int64_t)bh * n;
        hipemu::launch(dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, [&] { flash_combine_kernel(a, rows); });
      } else {
        hipemu::launch(dim3(a.nwg), dim3(256), lds, [&] { flash_attn_kernel<NS, PV, 4, false>(a); });
      }
    };
    using I1 = std::integral_constant<int, 1>;
    using I3 = std::integral_constant<int, 3>;
    if (pipe && nsplit != 4) {  // the software-pipelined kernel: plain fp16 operands, q carries log2(e)
      const int nw = pipe % 10, lds = flash_lds_bytes<1, 1>();
      a.log2q = 1;
      a.nqb = (n + 32 * nw - 1) / (32 * nw); a.nwg = bh * a.nqb;
      if (pipe == 4) hipemu::launch(dim3(a.nwg), dim3(256), lds, [&] { flash_pipe_kernel<4, false>(a); });
      else if (pipe == 14) hipemu::launch(dim3(a.nwg), dim3(256), lds, [&] { flash_pipe_kernel<4, true>(a); });
      else if (pipe == 6) hipemu::launch(dim3(a.nwg), dim3(384), lds, [&] { flash_pipe_kernel<6, false>(a); });
      else hipemu::launch(dim3(a.nwg), dim3(384), lds, [&] { flash_pipe_kernel<6, true>(a); });
    } else if (nsplit == 4) {
      // MX-corrected scores (flash_attn_kernel<2, 1>): the fp16 remainders of q_lo / k_lo become the P words PpEpiQKV::mx_qk writes — per row
      // [block][half-wave] x 32 bytes, the half-wave's 16 channels 32 blk + 8 (t / 4) + 4 h + t % 4, q as the activation, k as the weight
      auto pack = [&](const std::vector<f16>& hi, std::vector<f16>& lo, bool weight) {
        std::vector<f16> P(lo.size());
        for (size_t r = 0; r < hi.size() / 64; ++r)
          for (int blk = 0; blk < 2; ++blk)
            for (int h = 0; h < 2; ++h) {
              float v[16];
              for (int t = 0; t < 16; ++t) {
                const size_t c = r * 64 + blk * 32 + 8 * (t / 4) + 4 * h + (t % 4);
                v[t] = (float)hi[c] + (float)lo[c];
              }
              uint32_t hv[8], pw[8];
              if (weight) mx_pack16<true>(v, hv, pw); else mx_pack16<false>(v, hv, pw);
              memcpy(reinterpret_cast<char*>(&P[r * 64]) + blk * 64 + h * 32, pw, 32);
            }
        lo.swap(P);
      };
      pack(q, ql, false);
      pack(k, kl, true);
      a.q_lo = ql.data(); a.k_lo = kl.data();
      const int lds = flash_lds_bytes<2, 1>();
      if (pipe == 60) {  // the one-round launch of a single utterance: 192-row blocks, row sums on the VALU, lazy reference maximum
        a.log2q = 1;
        a.nqb = (n + 191) / 192; a.nwg = bh * a.nqb;
        hipemu::launch(dim3(a.nwg), dim3(384), lds, [&] { flash_attn_kernel<2, 1, 6, false, true, true>(a); });
      } else if (pipe == 42 || pipe == 43 || pipe == 62) {  // round 6: the same scores with V (42, 62) / V and P (43) as hi + lo halves (vt_lo.bin)
        if (!a.vt_lo) return 2;
        a.log2q = 1;
        if (pipe == 42) hipemu::launch(dim3(a.nwg), dim3(256), flash_lds_bytes<2, 2>(), [&] { flash_attn_kernel<2, 2, 4, false, false, true>(a); });
        else if (pipe == 43) hipemu::launch(dim3(a.nwg), dim3(256), flash_lds_bytes<2, 3>(), [&] { flash_attn_kernel<2, 3, 4, false, false, true>(a); });
        else {
          a.nqb = (n + 191) / 192; a.nwg = bh * a.nqb;
          hipemu::launch(dim3(a.nwg), dim3(384), flash_lds_bytes<2, 2>(), [&] { flash_attn_kernel<2, 2, 6, false, true, true>(a); });
        }
      } else if (pipe == 40) {  // many workgroups per CU: 128-row blocks, row sums on the matrix pipe, lazy reference maximum
        a.log2q = 1;
        hipemu::launch(dim3(a.nwg), dim3(256), lds, [&] { flash_attn_kernel<2, 1, 4, false, false, true>(a); });
      } else if (kvs > 1) {
        hipemu::launch(dim3(a.nwg), dim3(256), lds, [&] { flash_attn_kernel<2, 1, 4, true>(a); });
        const int64_t rows = (int64_t)bh * n;
        hipemu::launch(dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, [&] { flash_combine_kernel(a, rows); });
      } else {
        hipemu::launch(dim3(a.nwg), dim3(256), lds, [&] { flash_attn_kernel<2, 1, 4, false>(a); });
      }
    } else if (nsplit == 3) go(I3{}, I3{});
    else if (nsplit == 2) go(I3{}, I1{});
    else go(I1{}, I1{});
    wr("out.bin", o);
    if (!o_packed) wr("out_lo.bin", olo);
  } else {
    return 1;
  }
  return 0;
}
