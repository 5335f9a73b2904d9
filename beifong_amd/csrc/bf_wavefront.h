// Wavefront workspace: path state streamed through HBM in queue order.
//
// All per-path arrays are SoA float4/uint4 so that a wave reads and writes
// 1 KiB contiguous per array (16 B per lane, coalesced); the two state buffers
// ping-pong between bounces while survivors are compacted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bfd {

constexpr uint32_t kWfMaxIter = 4096;   // ring of per-bounce queue counters
constexpr uint32_t kSpillDepth = 16;    // stack entries beyond the 16 kept in LDS (tree depth <= 31)
constexpr uint32_t kTraceBlocksPerCU = 8;
constexpr uint32_t kMaxShadeWaves = 1u << 14;

struct WF {
    // path state, double buffered [2][capacity]
    float4 *ray0[2];    // o.xyz, mint
    float4 *ray1[2];    // d.xyz, maxt
    float4 *sa[2];      // throughput, eta, emission_weight, result
    float4 *sb[2];      // aux, bs_pdf, prev_p.x, prev_p.y
    uint4 *sc[2];       // prev_p.z, depth|flags, n_rays, -
    uint4 *sd[2];       // rng state lo/hi, path index lo/hi
    float4 *se[2];      // receive mode only: ray.time, t_rx, lambda0, -
    float4 *hit;        // t, u, v, slot      [capacity]
    // shadow-ray queue [capacity]
    float4 *sh0;        // o.xyz, mint
    float4 *sh1;        // d.xyz, maxt
    uint2 *sh2;         // slot in the next state buffer, NEE contribution bits
    int *spill;         // traversal-stack overflow: [kSpillDepth][max trace threads]
    // per-bounce counters [kWfMaxIter + 2]
    uint32_t *n_q;          // live slots entering bounce `it`
    uint32_t *n_sh;         // shadow rays produced by bounce `it`
    uint32_t *head_shade;   // work-queue heads of the persistent kernels
    uint32_t *head_trace;
    unsigned long long *counters;   // CTR_* (bf_device.h)
    unsigned long long *pool;       // per shade-wave pool of path indices {next, end} [kMaxShadeWaves]
    uint32_t capacity;
};

}  // namespace bfd
