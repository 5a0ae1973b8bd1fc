/*
 * f5hip_bench.h — kernel microbenchmarks, format checks and fault reproducers of libf5hip (TOOLS and TESTS only; no reference counterpart).
 *
 * Implemented by f5-tts_amd/csrc/libf5hip_bench.so (microbench.cpp + race_probe.hip), which links against libf5hip.so and reaches its
 * internal launchers; nothing here is needed to run the product, and libf5hip.so exports none of it (the reproducer
 * of the gfx950 packed-fp32 operand fault is compiled WITH the packed instructions, so it must not sit inside the engine's library).
 * Users: tools/kernel_bench.py, tests/test_pp_gemm_shim.py, tests/test_gpu_parity.py (tile checks), tests/test_gpu_race_probe.py.
 */
#ifndef F5HIP_BENCH_H
#define F5HIP_BENCH_H
#include "f5hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel microbenchmarks (tools/kernel_bench.py; no reference counterpart): average milliseconds of ONE launch of a
 * hot-path kernel on full-range random synthetic operands, HIP events on the launch stream, 2 warm-up launches.
 *   gemm:      a DiT block GEMM A[M,K] W[N,K]^T; epilogue 0 = +bias -> operand planes, 1 = FF1 (tanh-GELU -> operand planes),
 *              2 = out-proj/FF2 (fp32 residual += gate * (acc + bias)); variant -1 = launch heuristic, 0..5 = 64x128, 128x64,
 *              128x128, 256x128, 128x256, 256x256 tiles (rows x output channels)
 *   attention: the flash kernel over [batch2*heads, n, 64] (precision FP16 or FP16X3) */
int f5hip_bench_gemm(f5hip_ctx* ctx, int precision, int variant, int epilogue, int M, int N, int K, int iters, double* avg_ms);
int f5hip_bench_mx_pack(f5hip_ctx* ctx, int rows, int K, double* out9);
int f5hip_bench_qkv(f5hip_ctx* ctx, int precision, int variant, int seqs, int nseq, int K, int iters, int check, double* avg_ms, int64_t* diff);
int f5hip_bench_attention(f5hip_ctx* ctx, int precision, int batch2, int heads, int n, int iters, double* avg_ms);
/* Reproducer of a co-residency fault found in round 2 (csrc/race_probe.hip; no reference counterpart): `reps`
 * launches of the fused q|k|v GEMM on tile `variant` with epilogue form `expt`, each checked against the generic kernel; bad[r] = wrong
 * outputs of launch r; dump_path (or NULL) receives the wrong outputs as binary records. */
int f5hip_bench_qkv_probe(f5hip_ctx* ctx, int variant, int expt, int abl, int lds_pad, int noise, int seqs, int nseq, int reps, int64_t* bad,
                          const char* dump_path);

#ifdef __cplusplus
}
#endif
#endif
