// Per-slot path state of the wavefront pipeline and its (de)serialisation;
// shared by bf_wavefront.hip and the tail kernel in bf_kernels.hip.
#pragma once
#include "bf_device_math.h"
#include "bf_wavefront.h"

namespace bfd {

constexpr uint32_t kFlagValid = 1u << 24, kFlagFilmOk = 1u << 25, kFlagTermPending = 1u << 26, kDepthMask = 0xffffffu;

struct PathState {
    V3 ro, rd;
    float rmint, rmaxt;
    float throughput, eta, emission_weight, result;
    float aux, bs_pdf;
    V3 prev_p;
    uint32_t flags;      // depth | kFlag*
    uint32_t n_rays;
    Rng rng;
    uint64_t path_i;
};

BF_DEV void load_state(const WF &wf, int b, uint32_t i, PathState &s) {
    float4 r0 = wf.ray0[b][i], r1 = wf.ray1[b][i], a = wf.sa[b][i], bb = wf.sb[b][i];
    uint4 c = wf.sc[b][i], d = wf.sd[b][i];
    s.ro = mk(r0.x, r0.y, r0.z);
    s.rmint = r0.w;
    s.rd = mk(r1.x, r1.y, r1.z);
    s.rmaxt = r1.w;
    s.throughput = a.x;
    s.eta = a.y;
    s.emission_weight = a.z;
    s.result = a.w;
    s.aux = bb.x;
    s.bs_pdf = bb.y;
    s.prev_p = mk(bb.z, bb.w, __uint_as_float(c.x));
    s.flags = c.y;
    s.n_rays = c.z;
    s.rng.state = ((uint64_t) d.y << 32) | d.x;
    s.path_i = ((uint64_t) d.w << 32) | d.z;
}
BF_DEV void store_state(const WF &wf, int b, uint32_t j, const PathState &s) {
    wf.ray0[b][j] = make_float4(s.ro.x, s.ro.y, s.ro.z, s.rmint);
    wf.ray1[b][j] = make_float4(s.rd.x, s.rd.y, s.rd.z, s.rmaxt);
    wf.sa[b][j] = make_float4(s.throughput, s.eta, s.emission_weight, s.result);
    wf.sb[b][j] = make_float4(s.aux, s.bs_pdf, s.prev_p.x, s.prev_p.y);
    wf.sc[b][j] = make_uint4(__float_as_uint(s.prev_p.z), s.flags, s.n_rays, 0u);
    wf.sd[b][j] = make_uint4((uint32_t) s.rng.state, (uint32_t) (s.rng.state >> 32), (uint32_t) s.path_i,
                             (uint32_t) (s.path_i >> 32));
}

}  // namespace bfd
