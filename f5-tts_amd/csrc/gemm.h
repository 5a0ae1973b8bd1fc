// gemm.h — LDS-tiled MFMA GEMM for gfx950:  out[M,N] = epilogue( A[M,K] . W[N,K]^T )
//
// Both operands are K-contiguous (torch nn.Linear weight layout [out, in]); tiles of 128 bytes of K
// per row (64 halves / 32 floats) are staged global -> registers -> LDS (row stride 144 B: the +16 B
// pad makes the 32-row ds_read_b128 fragment reads conflict-free), double-buffered in LDS with the
// next tile's global loads in flight during the MFMAs of the current tile.  256 threads = 4 waves as
// 2(m) x 2(n); each wave owns TM x TN tiles of 32x32 (v_mfma_f32_32x32x16_f16 or 4x
// v_mfma_f32_32x32x2_f32).  The weight axis is the MFMA "A" operand (accumulator rows) so a lane
// owns 4 consecutive output channels of one activation row -> 16-byte epilogue accesses.
//
// NSPLIT == 3: fp16 hi/lo split operands, acc += A_hi.W_hi + A_lo.W_hi + A_hi.W_lo (~fp32 accuracy).
#pragma once
#include "common.h"

struct GemmCore {
  const void* A;      // [M, K] activations (hi plane)
  const void* A_lo;   // lo plane (NSPLIT == 3)
  const void* W;      // [N, K] weights (hi plane)
  const void* W_lo;
  int64_t lda, ldw;   // row strides in elements
  int64_t strideA, strideW;      // blockIdx.z batch strides in elements
  int M, N, K;
  int a_rows;         // rows of A that exist (<= M): rows beyond are read as zero
  int w_rows;         // rows of W that exist (<= N)
};

// Generic store epilogue:
//   v = alpha * acc + bias[n];  v = act(v);  v *= colscale[n];
//   if (rowmask && !rowmask[m] && mask_mode == 1) v = 0;
//   if (out2) out2[m,n] = v + res2[m,n];
//   if (res) v += res[m,n];
//   if (rowmask && !rowmask[m] && mask_mode == 2) v = 0;
//   out32[m,n] = v;  out16(_lo)[m,n] = split(v)
struct EpiStore {
  float alpha;
  int act;
  const float* bias;
  const float* colscale;
  const float* res;
  int64_t ldres;
  const float* res2;
  float* out2;
  const uint8_t* rowmask;
  int mask_mode;
  float* out32;
  f16* out16;
  f16* out16_lo;
  int64_t ldo;
  // batch (blockIdx.z) addressing of the outputs/residual: off = (z / zdiv) * so1 + (z % zdiv) * so2
  int zdiv;
  int64_t so1, so2;

  __device__ __forceinline__ void operator()(int m, int n, float4 v, int z) const {
    float x[4] = {v.x, v.y, v.z, v.w};
    if (bias) {
      const float4 b = *reinterpret_cast<const float4*>(bias + n);
      x[0] = alpha * x[0] + b.x; x[1] = alpha * x[1] + b.y; x[2] = alpha * x[2] + b.z; x[3] = alpha * x[3] + b.w;
    } else {
      x[0] *= alpha; x[1] *= alpha; x[2] *= alpha; x[3] *= alpha;
    }
    if (act != ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = apply_act(act, x[e]);
    }
    if (colscale) {
      const float4 c = *reinterpret_cast<const float4*>(colscale + n);
      x[0] *= c.x; x[1] *= c.y; x[2] *= c.z; x[3] *= c.w;
    }
    const bool dead = rowmask && !rowmask[m];
    if (dead && mask_mode == 1) { x[0] = x[1] = x[2] = x[3] = 0.f; }
    const int64_t zoff = zdiv ? (int64_t)(z / zdiv) * so1 + (int64_t)(z % zdiv) * so2 : 0;
    const int64_t o = zoff + (int64_t)m * ldo + n;
    if (out2) {
      const float4 r2 = *reinterpret_cast<const float4*>(res2 + o);
      *reinterpret_cast<float4*>(out2 + o) = make_float4(x[0] + r2.x, x[1] + r2.y, x[2] + r2.z, x[3] + r2.w);
    }
    if (res) {
      const float4 r = *reinterpret_cast<const float4*>(res + zoff + (int64_t)m * ldres + n);
      x[0] += r.x; x[1] += r.y; x[2] += r.z; x[3] += r.w;
    }
    if (dead && mask_mode == 2) { x[0] = x[1] = x[2] = x[3] = 0.f; }
    if (out32) *reinterpret_cast<float4*>(out32 + o) = make_float4(x[0], x[1], x[2], x[3]);
    if (out16) {
      f16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) { f16 h, l; split_f16(x[e], h, l); hi[e] = h; lo[e] = l; }
      *reinterpret_cast<f16x4*>(out16 + o) = hi;
      if (out16_lo) *reinterpret_cast<f16x4*>(out16_lo + o) = lo;
    }
  }
};

// QKV epilogue: bias, rotary embedding on (2i,2i+1) pairs of q and k (x_transformers
// apply_rotary_pos_emb, reference call sites model/modules.py:503-509), q * scale, scatter to
// per-head layouts.  n in [0, 3*inner): n / inner selects q|k|v.
//   half mode  (q16 != null): q16/k16 [B'*H, nseq, dh] f16, vt16 [B'*H, dh, ldvt] f16 (V transposed for the
//                              flash kernel's V^T tiles); optional *_lo planes (fp16 hi/lo split)
//   float mode (q32 != null): q32/k32 [B'*H, nseq, dh] f32, vt32 [B'*H, dh, ldvt] f32 (V transposed)
struct EpiQKV {
  const float* bias;    // [3*inner]
  const float* rope_cs; // [nseq, dh/2, 2] (cos, sin) fp32
  int nseq, heads, dh;
  int pe_heads;         // -1 = all
  float qscale;
  f16 *q16, *k16, *vt16;
  f16 *q16_lo, *k16_lo, *vt16_lo;
  float *q32, *k32, *vt32;
  int64_t ldvt;

  __device__ __forceinline__ void operator()(int m, int n, float4 v, int /*z*/) const {
    const int inner = heads * dh;
    const int which = n / inner;
    const int c = n - which * inner;
    const int hh = c / dh, d = c - hh * dh;
    const int bp = m / nseq, pos = m - bp * nseq;
    const float4 b = *reinterpret_cast<const float4*>(bias + n);
    float x[4] = {v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w};
    if (which < 2 && (pe_heads < 0 || hh < pe_heads)) {
      const float4 cs = *reinterpret_cast<const float4*>(rope_cs + ((int64_t)pos * (dh / 2) + d / 2) * 2);
      const float a0 = x[0] * cs.x - x[1] * cs.y, a1 = x[1] * cs.x + x[0] * cs.y;
      const float a2 = x[2] * cs.z - x[3] * cs.w, a3 = x[3] * cs.z + x[2] * cs.w;
      x[0] = a0; x[1] = a1; x[2] = a2; x[3] = a3;
    }
    if (which == 0) { x[0] *= qscale; x[1] *= qscale; x[2] *= qscale; x[3] *= qscale; }
    const int64_t bh = (int64_t)bp * heads + hh;
    if (q16) {
      f16x4 hv, lv;
#pragma unroll
      for (int e = 0; e < 4; ++e) { f16 h, l; split_f16(x[e], h, l); hv[e] = h; lv[e] = l; }
      if (which < 2) {
        const int64_t off = (bh * nseq + pos) * dh + d;
        *reinterpret_cast<f16x4*>((which == 0 ? q16 : k16) + off) = hv;
        if (q16_lo) *reinterpret_cast<f16x4*>((which == 0 ? q16_lo : k16_lo) + off) = lv;
      } else {
        const int64_t off = (bh * dh + d) * ldvt + pos;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vt16[off + e * ldvt] = hv[e];
          if (vt16_lo) vt16_lo[off + e * ldvt] = lv[e];
        }
      }
    } else {
      if (which < 2) {
        float* dst = (which == 0 ? q32 : k32) + (bh * nseq + pos) * dh + d;
        *reinterpret_cast<float4*>(dst) = make_float4(x[0], x[1], x[2], x[3]);
      } else {
        float* dst = vt32 + (bh * dh + d) * ldvt + pos;
        dst[0] = x[0]; dst[ldvt] = x[1]; dst[2 * ldvt] = x[2]; dst[3 * ldvt] = x[3];
      }
    }
  }
};

constexpr int GEMM_ROWB = 144;  // LDS bytes per tile row (128 data + 16 pad)

template <typename T, int NSPLIT, int TM, int TN>
constexpr int gemm_lds_bytes() {
  return 2 * (64 * TM + 64 * TN) * GEMM_ROWB * (NSPLIT == 3 ? 2 : 1);
}

template <typename T, int NSPLIT, int TM, int TN, typename Epi>
__global__ __launch_bounds__(256) void gemm_kernel(GemmCore g, Epi epi) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int KT = 128 / (int)sizeof(T);   // k elements per tile
  constexpr int KC = 16 / (int)sizeof(T);    // k elements per 16-byte chunk
  constexpr int NPL = (NSPLIT == 3) ? 2 : 1;
  constexpr int CA = BM * 8 / 256, CW = BN * 8 / 256;  // chunks per thread per plane
  constexpr int PLANE_A = BM * GEMM_ROWB, PLANE_W = BN * GEMM_ROWB;
  constexpr int STAGE = NPL * (PLANE_A + PLANE_W);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, z = blockIdx.z;

  const T* Ap[NPL];
  const T* Wp[NPL];
  Ap[0] = reinterpret_cast<const T*>(g.A) + (int64_t)z * g.strideA;
  Wp[0] = reinterpret_cast<const T*>(g.W) + (int64_t)z * g.strideW;
  if constexpr (NPL == 2) {
    Ap[1] = reinterpret_cast<const T*>(g.A_lo) + (int64_t)z * g.strideA;
    Wp[1] = reinterpret_cast<const T*>(g.W_lo) + (int64_t)z * g.strideW;
  }

  uint4 ra[NPL][CA], rw[NPL][CW];

  auto load_global = [&](int kt) {
    const int kbase = kt * KT;
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int c = tid + i * 256, row = c >> 3, col = (c & 7) * KC;
      const int gm = m0 + row, gk = kbase + col;
      const bool ok = gm < g.a_rows && gk < g.K;
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        ra[p][i] = ok ? *reinterpret_cast<const uint4*>(Ap[p] + (int64_t)gm * g.lda + gk) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < CW; ++i) {
      const int c = tid + i * 256, row = c >> 3, col = (c & 7) * KC;
      const int gn = n0 + row, gk = kbase + col;
      const bool ok = gn < g.w_rows && gk < g.K;
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        rw[p][i] = ok ? *reinterpret_cast<const uint4*>(Wp[p] + (int64_t)gn * g.ldw + gk) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_lds = [&](int stage) {
    char* base = smem + stage * STAGE;
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
#pragma unroll
      for (int i = 0; i < CA; ++i) {
        const int c = tid + i * 256, row = c >> 3, col = c & 7;
        *reinterpret_cast<uint4*>(base + p * PLANE_A + row * GEMM_ROWB + col * 16) = ra[p][i];
      }
#pragma unroll
      for (int i = 0; i < CW; ++i) {
        const int c = tid + i * 256, row = c >> 3, col = c & 7;
        *reinterpret_cast<uint4*>(base + NPL * PLANE_A + p * PLANE_W + row * GEMM_ROWB + col * 16) = rw[p][i];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int j = 0; j < TM; ++j)
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  const int nkt = (g.K + KT - 1) / KT;
  load_global(0);
  store_lds(0);
  __syncthreads();

  const int frag_off = (lane & 31) * GEMM_ROWB + (lane >> 5) * 16;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) load_global(kt + 1);
    const char* base = smem + (kt & 1) * STAGE;
    const char* sA = base + (wm * 32 * TM) * GEMM_ROWB + frag_off;
    const char* sW = base + NPL * PLANE_A + (wn * 32 * TN) * GEMM_ROWB + frag_off;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      Frag fa[NPL][TM], fw[NPL][TN];
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
#pragma unroll
        for (int j = 0; j < TM; ++j)
          fa[p][j].u = *reinterpret_cast<const uint4*>(sA + p * PLANE_A + j * 32 * GEMM_ROWB + ks * 32);
#pragma unroll
        for (int i = 0; i < TN; ++i)
          fw[p][i].u = *reinterpret_cast<const uint4*>(sW + p * PLANE_W + i * 32 * GEMM_ROWB + ks * 32);
      }
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          Mma32<T>::mma(acc[j][i], fw[0][i], fa[0][j]);
          if constexpr (NPL == 2) {
            Mma32<T>::mma(acc[j][i], fw[0][i], fa[1][j]);  // W_hi . A_lo
            Mma32<T>::mma(acc[j][i], fw[1][i], fa[0][j]);  // W_lo . A_hi
          }
        }
    }
    if (kt + 1 < nkt) store_lds((kt + 1) & 1);
    __syncthreads();
  }

  // epilogue: lane owns row m = .. + (lane & 31), channels n = .. + 8q + 4*(lane>>5) + 0..3
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + wm * 32 * TM + j * 32 + (lane & 31);
    if (m >= g.M) continue;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 32 * TN + i * 32 + 8 * q + 4 * (lane >> 5);
        if (n < g.N) epi(m, n, make_float4(acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3]), z);
      }
    }
  }
}
