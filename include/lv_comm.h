/*
 * lv_comm.h — C ABI of the direct small-message all-reduce of liblvattn.so (MI355X, gfx950).
 *
 * It replaces, for decode-sized messages, the sum all-reduce the tensor-parallel layers of
 * light-vllm issue through torch.distributed/NCCL:
 *
 *   lv_comm_all_reduce  <- dist.all_reduce(y) in RowParallelLinear.forward   layers/linear.py:388-389
 *                          dist.all_reduce(y) in VocabParallelEmbedding.forward   layers/embed_head.py:112
 *
 * A decode step issues 2 * layers + 1 of them on [batch, hidden] bf16 activations (8 KiB per
 * sequence): latency-bound messages. MI355X GPUs of a node are fully connected by xGMI links, so every
 * rank can read every peer's copy directly: one kernel per rank, no ring, no host involvement.
 *
 * Model
 *   - one process per GPU (or several processes on one GPU: the buffers are exchanged as HIP IPC
 *     handles either way); every rank creates a communicator, the host exchanges the handle blobs
 *     by any means it has (torch.distributed.all_gather_object, a file, a pipe) and connects;
 *   - lv_comm_all_reduce sums `count` bf16 elements in place over all ranks: fp32 accumulation in
 *     rank order 0..world-1 and one rounding to bf16, so every rank ends with the same bits;
 *   - each call only enqueues one kernel on `stream`: no allocation, no host synchronisation, the
 *     sequence numbers live in device memory, so a call can be captured into a hipGraph and
 *     replayed (every rank must then replay the same sequence of calls);
 *   - all ranks must issue the same calls (same count) in the same order, as with any collective;
 *   - a rank that waits longer than the communicator's timeout for a peer gives up: its kernel
 *     ends (its output is then undefined), the communicator is marked failed, and this and every
 *     later call on it returns LV_ERR_COMM. Kernels never spin without a bound.
 */
#ifndef LV_COMM_H
#define LV_COMM_H

#include <stddef.h>
#include <stdint.h>

#include "lv_attention.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LV_ERR_COMM (-5)

#define LV_COMM_HANDLE_BYTES 64           /* one HIP IPC memory handle */
#define LV_COMM_MAX_RANKS 8
#define LV_COMM_MAX_BYTES (8u << 20)      /* largest message lv_comm_all_reduce takes: [512, 8192] bf16 */

/* algorithm of lv_comm_all_reduce */
#define LV_COMM_ALGO_AUTO 0               /* by message size and rank count */
#define LV_COMM_ALGO_ONE_SHOT 1           /* every rank reads every rank's whole message: one exchange */
#define LV_COMM_ALGO_TWO_SHOT 2           /* rank r sums 1/world of it and writes that to all: two exchanges */

typedef struct lv_comm lv_comm_t;

/*
 * Create this rank's end of a communicator on the current device. Allocates (uncached device
 * memory, visible to peers): two staging buffers of LV_COMM_MAX_BYTES and the flag words.
 * timeout_ms: how long a kernel waits for a peer before giving up (0 = 10000).
 */
int lv_comm_create(int rank, int world_size, int timeout_ms, lv_comm_t** comm);

/* This rank's handle blob (LV_COMM_HANDLE_BYTES bytes) to hand to every peer. */
int lv_comm_get_handle(lv_comm_t* comm, void* handle_out);

/*
 * Open every peer's buffers. all_handles: world_size blobs of LV_COMM_HANDLE_BYTES in rank order
 * (this rank's own entry is ignored). Call once, before the first all-reduce, on every rank.
 */
int lv_comm_connect(lv_comm_t* comm, const void* all_handles);

/*
 * data[i] = sum over ranks of data[i], i in [0, count), bf16, in place. count * 2 must be a multiple
 * of 16 bytes and at most LV_COMM_MAX_BYTES, data 16-byte aligned device memory of this rank.
 * algorithm: LV_COMM_ALGO_*; both give the same bits (fp32 sums in rank order, one rounding).
 * Every rank must pass the same algorithm for the same call.
 * Returns LV_ERR_COMM if an earlier call on this communicator timed out.
 */
int lv_comm_all_reduce(lv_comm_t* comm, void* data, size_t count, int algorithm, lv_stream_t stream);

/*
 * The all-reduce followed by the residual add + RMSNorm the tensor-parallel layers run on its result
 * (dist.all_reduce in RowParallelLinear.forward, layers/linear.py:388-389, then
 * RMSNorm.add_rms_forward, layers/layernorm.py:52-80), in one kernel:
 *   y = bf16(sum over ranks of x)   s = float(y) + float(residual)   residual = bf16(s)
 *   x = bf16(s * rsqrt(mean(s^2) + eps)) * weight
 * x, residual: [rows, hidden] bf16, contiguous, in place; weight: [hidden] bf16. Same bits as
 * lv_comm_all_reduce(x) followed by lv_add_rms_norm(x, residual, weight). One exchange (the one-shot
 * form): meant for the message sizes lv_comm_all_reduce serves in one shot (at the default threshold and
 * hidden 4096: up to 256 rows at 2 ranks, 85 at 4, 36 at 8; callers send larger messages through
 * lv_comm_all_reduce, which goes two shot there, and lv_add_rms_norm). hidden: a multiple of 8
 * up to 8192; rows * hidden * 2 <= LV_COMM_MAX_BYTES. The usual collective rules apply.
 */
int lv_comm_all_reduce_add_rms_norm(lv_comm_t* comm, void* x, void* residual, const void* weight, int64_t rows,
                                    int hidden, float eps, lv_stream_t stream);

/*
 * LV_COMM_ALGO_AUTO serves a message in one shot while (world_size - 1) * bytes, what one shot moves
 * into every rank, stays within this many bytes, and in two shots beyond. Default 2 MiB (or the
 * environment's LV_COMM_ONE_SHOT_INFLOW_BYTES when the communicator is created); a figure derived from
 * byte counts that the first run across GPUs should replace (bench.py --gpus N reports both
 * algorithms from 64 KiB to 8 MiB). Must be set to the same value on every rank, between calls.
 */
int lv_comm_set_one_shot_inflow(lv_comm_t* comm, size_t bytes);
size_t lv_comm_one_shot_inflow(lv_comm_t* comm);

/* 0 while healthy; non-zero once a kernel of this rank has given up waiting (host-visible word,
 * no device synchronisation). */
int lv_comm_failed(lv_comm_t* comm);

/* Close the peers' mappings and free this rank's buffers. Synchronises the device. */
int lv_comm_destroy(lv_comm_t* comm);

#ifdef __cplusplus
}
#endif
#endif /* LV_COMM_H */
