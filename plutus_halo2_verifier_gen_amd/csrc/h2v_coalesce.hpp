// Gathering the proofs of several small device-resident calls into one staging batch (h2v_capi.hip: coalesce_call), so that ONE
// launch per kernel serves them all: a stream of 64- or 128-proof calls (the per-GPU shares of a batch cut over eight GPUs) is a
// stream of lone-wave chains otherwise.  Two small kernels per call, on the lane's stream, at the time of the call:
//   k_coalesce_offsets   the call's n proof offsets re-based behind what the group already holds.  A proof keeps its length
//                        when it is short (< plan.proof_len: the kernels reject it by that predicate alone) and is cut to
//                        proof_len otherwise (the kernels never read past it: trailing bytes are ignored, as by the reference's
//                        reader) - so the group's bytes fit a buffer of capacity x proof_len whatever the caller's offsets say.
//   k_coalesce_copy      the bytes, one block per proof.
// Instances and committed instances have a fixed stride and travel by hipMemcpyAsync; accept / status come back the same way.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// dst_off[0] holds the group's current end (0 for an empty group: memset); writes dst_off[1 .. n].  One block of 256 threads.
extern "C" __global__ void __launch_bounds__(256)
k_coalesce_offsets(const uint64_t *__restrict__ src_off, uint32_t n, uint32_t proof_len, uint64_t *__restrict__ dst_off) {
    __shared__ uint64_t part[256];
    const uint32_t t = threadIdx.x, per = (n + 255) / 256, lo = t * per, hi = lo + per < n ? lo + per : n;
    uint64_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) {
        const uint64_t len = src_off[i + 1] - src_off[i];
        sum += len < proof_len ? len : proof_len;
    }
    part[t] = sum;
    __syncthreads();
    for (uint32_t s = 1; s < 256; s <<= 1) {            // inclusive scan of the 256 segment sums
        const uint64_t v = t >= s ? part[t - s] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t run = dst_off[0] + (t ? part[t - 1] : 0);
    __syncthreads();                                     // (every thread has read dst_off[0] before thread 0 could reach dst_off[1] ... fine either way: [0] is not written)
    for (uint32_t i = lo; i < hi; i++) {
        const uint64_t len = src_off[i + 1] - src_off[i];
        run += len < proof_len ? len : proof_len;
        dst_off[i + 1] = run;
    }
}
// block i copies proof i of the call: src_proofs + src_off[i] -> dst_proofs + dst_off[i], dst_off[i + 1] - dst_off[i] bytes
extern "C" __global__ void __launch_bounds__(256)
k_coalesce_copy(const uint8_t *__restrict__ src_proofs, const uint64_t *__restrict__ src_off, uint8_t *__restrict__ dst_proofs,
                const uint64_t *__restrict__ dst_off, uint32_t n) {
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const uint8_t *src = src_proofs + src_off[i];
    uint8_t *dst = dst_proofs + dst_off[i];
    const uint64_t len = dst_off[i + 1] - dst_off[i];
    if ((((uintptr_t)src | (uintptr_t)dst) & 3u) == 0) {
        const uint32_t words = (uint32_t)(len >> 2);
        for (uint32_t k = threadIdx.x; k < words; k += 256) ((uint32_t *)dst)[k] = ((const uint32_t *)src)[k];
        for (uint64_t k = (uint64_t)words * 4 + threadIdx.x; k < len; k += 256) dst[k] = src[k];
    } else {
        for (uint64_t k = threadIdx.x; k < len; k += 256) dst[k] = src[k];
    }
}
