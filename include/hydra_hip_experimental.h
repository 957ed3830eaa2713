/*
 * hydra_hip_experimental.h — entry points that exist only in a library built with
 * `make -C hydrainfer_amd/csrc EXPERIMENTS=1`: kernels that were built, tested, measured and REJECTED
 * (DESIGN.md §6c), and the read-stream microbenchmarks behind tools/bench_stream.py /
 * tools/bench_attn_ceiling.py.  None of this is on the product path; the default libhydra_hip.so does
 * not export these symbols and hydrainfer_amd binds them only when they are present.
 *   "decode_hpw4" option      — four heads per decode-attention workgroup (attn_decode4.hip: -4 % standalone,
 *                              +-0 in the decode step)
 *   hx_debug_stream_read / hx_debug_paged_read — access-shape probes
 *   hx_debug_fwd_stamps       — in-kernel time stamps of the persistent prefill attention kernel (tools/fwd_timeline.py)
 */
#ifndef HYDRA_HIP_EXPERIMENTAL_H
#define HYDRA_HIP_EXPERIMENTAL_H

#include "hydra_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Debug / tooling (not on the product path): read-streaming microbenchmark used by
 * tools/bench_stream.py to choose load shapes.  variant 0: contiguous 1 KiB per wave instruction;
 * 1..4: 8x128 B, 4x256 B, 2x512 B, 1x1024 B (rows x bytes per instruction) of a row-major matrix
 * with row pitch `pitch` bytes.  unroll = loads in flight per wave (4, 8, 16, 32); policy 1 =
 * non-temporal loads.  Reads `bytes` bytes once. */
int hx_debug_stream_read(const void* p, int64_t bytes, int variant, int64_t pitch, int unroll,
                         int policy, int wgs, float* sink, hx_stream stream);
/* measurement aid: the decode attention kernel's read pattern (paged, one head's 256 B of every key row) with
 * no arithmetic — the ceiling that kernel can reach (tools/bench_attn_ceiling.py) */
int hx_debug_paged_read(const void* kbase, const void* vbase, const int32_t* table, int n_seq, int n_heads,
                        int tiles, int64_t page_bytes, int row_bytes, int heads_per_wg, int waves, int depth,
                        int n_splits, float* sink, hx_stream stream);

/* measurement aid: while `buf` is non-null, bf16 / head_dim 128 / paged launches of the persistent prefill attention
 * kernel write (100 MHz time << 8 | event) words, 512 per workgroup, into it (events: attn_fwd.hip, STAMPS) */
int hx_debug_fwd_stamps(void* buf);

#ifdef __cplusplus
}
#endif
#endif /* HYDRA_HIP_EXPERIMENTAL_H */
