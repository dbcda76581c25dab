// pt_traverse_sched.h -- wave-scheduled persistent traversal.
//
// Measured on incoherent bounces with one-ray-per-lane-until-the-batch-ends traversal of binary
// nodes: only 14-19 % of the vector lanes do useful work (SQ_THREAD_CYCLES_VALU / 64 / SQ_ACTIVE_INST_VALU), because
// (a) the 64 rays of a batch need very different numbers of steps and the wave runs as long as its
//     longest ray, and
// (b) lanes sit in different phases (box step, triangle test, instance entry) and every phase present
//     in the wave is executed on every iteration, each at its own low lane count.
// This kernel removes both:
// (a) ray replacement: a lane whose ray has finished takes the next ray of the wave's private chunk
//     (a shard's fair share, at most 256 slots, claimed with one atomic), so lanes stay populated
//     until the queue is empty;
// (b) phase voting: the wave executes ONE phase at a time -- the one its lanes vote for, for a few
//     steps in a row -- and the other lanes wait, so the long phases (triangle test, instance entry)
//     only run once enough lanes have accumulated in them.
// Registers decide how many waves hide each other's memory latency, and a fifth wave per SIMD is worth
// 7 % here, so per-lane state that is touched rarely lives in LDS, [word][thread]:
//   ws    the world-space origin, direction and reciprocals while a lane is inside an instance (leaving
//         the last instance with an empty stack ends the ray without reading them; otherwise the few
//         lanes that leave read them back, which costs the wave nine LDS loads and three products
//         instead of three divisions);
//   cold  the ids and barycentrics of the closest hit so far (read only on an exact distance tie and
//         when the ray ends), the ray's queue slot and the instance's delta.
// In registers remain the current ray space, the closest distance and six scalars; the triangle test
// rebuilds its axis permutation from the ray and handles one triangle per step (two at once would
// need four registers too many).  That is 92 registers: five waves per SIMD without spills.
// The per-lane arithmetic (slab test, gated watertight triangle test, hit ordering) is the one in
// pt_device.h, so results are bit-identical to the plain traversal.
#pragma once
#include "pt_device.h"

#ifndef PT_SCHED_CHUNK
#define PT_SCHED_CHUNK 256u        // most queue slots a wave claims per atomic (fewer when the queue is short)
#endif
#ifndef PT_SCHED_REFILL
#define PT_SCHED_REFILL 24         // refill when at least this many lanes are idle (five waves per SIMD cover for the idle lanes)
#endif
#ifndef PT_SCHED_LEAF_VOTES
#define PT_SCHED_LEAF_VOTES 20     // run the triangle phase once this many lanes wait in it ...
#endif
#ifndef PT_SCHED_ENTER_VOTES
#define PT_SCHED_ENTER_VOTES 16    // ... the instance-entry phase at this many ...
#endif
#ifndef PT_SCHED_NODE_MIN
#define PT_SCHED_NODE_MIN 24       // ... or when fewer than this many lanes want a box step
#endif
#ifndef PT_SCHED_NODE_STEPS
#define PT_SCHED_NODE_STEPS 10     // box steps per vote while enough lanes take part
#endif
#ifndef PT_SCHED_NODE_STAY
#define PT_SCHED_NODE_STAY 24      // ... that is, while this many lanes still do
#endif
#ifndef PT_SCHED_LEAF_STEPS
#define PT_SCHED_LEAF_STEPS 4      // triangles per vote while enough lanes take part
#endif

#ifdef PT_SCHED_STATS
// development build: [0..2] iterations of the node / triangle / entry phase, [3..5] active lanes in them,
// [6] outer loop iterations, [7] refill rounds
__device__ unsigned long long pt_sched_stats[8];
#define PT_STAT(i, v) do { if (lane == 0) atomicAdd(&pt_sched_stats[i], (unsigned long long)(v)); } while (0)
#else
#define PT_STAT(i, v) do { } while (0)
#endif

struct PtSchedQueue {              // the sharded queue of one bounce
    uint32_t* counters;
    uint32_t shard_cap, n_shards, ctr_pad;
    size_t count_base, work_base;  // word offsets of the size / work counters of shard 0 at this depth
};

template <bool BY_ID>
__device__ __forceinline__ void pt_traverse_sched(const PtTlasView& tl, const PtSchedQueue& Q, PtStack& stk, pt_lds_float* ws, pt_lds_int* cold,
                                                  const float4* __restrict__ q0, const float4* __restrict__ q1,
                                                  float tmin, float tmax, PtHit* __restrict__ hits, float* __restrict__ t_out)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;

    // ---- wave-uniform work cursor.  Lane s looks at shard s once (the queue only shrinks while the
    // kernel runs); shards with work left are the set bits of `avail`, visited round-robin from the
    // wave's own shard, so a drained level costs one round trip instead of one per shard.
    const uint32_t ns_lane = lane < Q.n_shards ? Q.counters[Q.count_base + (size_t)lane * Q.ctr_pad] : 0u;
    const uint32_t wk_lane = lane < Q.n_shards
        ? __hip_atomic_load(&Q.counters[Q.work_base + (size_t)lane * Q.ctr_pad], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    unsigned long long avail = __ballot(wk_lane < ns_lane);
    const uint32_t start = wave & 63u;
    uint32_t wcur = 0, wend = 0;
    bool drained = avail == 0ull;
    // a claim is a shard's fair share per wave, in whole batches: long queues amortise the atomic,
    // short ones (deep bounces) still spread over all waves
    const uint32_t waves_per_shard = max(1u, (gridDim.x * (PT_TRAVERSE_BLOCK / 64u)) / Q.n_shards);

    // ---- per-lane traversal state
    PtRaySpace cur;
    float best_t = tmax;           // distance of the closest hit so far; its ids and barycentrics are in `cold`
    float tlo = tmin, thi = tmax;
    uint32_t cur_inst = PT_NO_INSTANCE;
#define PT_COLD(k) cold[(k) * PT_TRAVERSE_BLOCK]      // 0 instance, 1 primitive, 2 u, 3 v, 4 queue slot, 5 delta
    int node = PT_NODE_DONE;       // PT_NODE_DONE: lane is idle
    stk.sp = 0;
    cur.ox = cur.oy = cur.oz = cur.dx = cur.dy = cur.dz = cur.ix = cur.iy = cur.iz = cur.oox = cur.ooy = cur.ooz = 0.0f;

    // A lane that has come to the end of an instance's BLAS pops the sentinel pushed when it entered and keeps it as
    // its item: a few lanes at a time get there and the whole wave would pay for what they do, so the lane waits for
    // the instance phase, which takes it back to world space together with the lanes that enter or leave an
    // instance then.  An instance entered with nothing else on the stack pushes no sentinel: when its BLAS is
    // exhausted the ray is finished (the pop says so) and its world-space form is not needed again.

    // A finished ray keeps its closest hit in `cold` until its lane is refilled (or the wave ends): rays
    // finish one or two lanes at a time, and a record written where the ray ends costs the whole wave
    // a pass per ray (10 % of the kernel); written here, one pass serves every lane that waits.
    PT_COLD(4) = -1;               // queue slot of the ray whose record is still to be written, or -1 (a flag in a register was measured 1 % slower)
    // Hands the lane the ray of queue slot `tag`.
    auto start_ray = [&](const float4 o, const float4 d, int tag) {
        PT_COLD(4) = tag; PT_COLD(0) = (int)PT_NO_INSTANCE; PT_COLD(1) = -1; PT_COLD(2) = 0; PT_COLD(3) = 0;
        pt_ray_space_init(cur, o.x, o.y, o.z, d.x, d.y, d.z);
        ws[0 * PT_TRAVERSE_BLOCK] = cur.ox; ws[1 * PT_TRAVERSE_BLOCK] = cur.oy; ws[2 * PT_TRAVERSE_BLOCK] = cur.oz;
        ws[3 * PT_TRAVERSE_BLOCK] = cur.dx; ws[4 * PT_TRAVERSE_BLOCK] = cur.dy; ws[5 * PT_TRAVERSE_BLOCK] = cur.dz;
        ws[6 * PT_TRAVERSE_BLOCK] = cur.ix; ws[7 * PT_TRAVERSE_BLOCK] = cur.iy; ws[8 * PT_TRAVERSE_BLOCK] = cur.iz;
        cur_inst = PT_NO_INSTANCE; tlo = tmin; thi = tmax;
        best_t = tmax;
        node = pt_ray_start(tl, stk);
    };
    auto flush = [&]() {
        if (node == PT_NODE_DONE && PT_COLD(4) != -1) {
            const uint32_t qi = (uint32_t)PT_COLD(4);
            PtHit out; out.instance = (uint32_t)PT_COLD(0); out.primitive = (uint32_t)PT_COLD(1); out.u = __int_as_float(PT_COLD(2)); out.v = __int_as_float(PT_COLD(3));
            const uint32_t dst = BY_ID ? __float_as_uint(q1[qi].w) : qi;
            if (BY_ID) hits[dst] = out;     // the caller's array: no alignment beyond the record's own
            else pt_stream_store((float4*)&hits[dst], make_float4(__uint_as_float(out.instance), __uint_as_float(out.primitive), out.u, out.v));
            if (BY_ID && t_out) t_out[dst] = best_t;
            PT_COLD(4) = -1;
        }
    };
    for (;;) {
        // ------------------------------------------------------------------ refill idle lanes
        unsigned long long idle = __ballot(node == PT_NODE_DONE);
        const bool refill = idle == ~0ull || __popcll(idle) >= PT_SCHED_REFILL;
        if (refill) flush();
        if (!drained && refill) {
            while (idle) {
                PT_STAT(7, 1);
                if (wcur == wend) {
                    bool got = false;
                    while (avail) {
                        // next shard with work at or after the wave's own
                        const unsigned long long rot = start ? ((avail >> start) | (avail << (64u - start))) : avail;
                        const uint32_t s = ((uint32_t)__builtin_ctzll(rot) + start) & 63u;
                        const uint32_t n_s = __builtin_amdgcn_readlane(ns_lane, s);
                        uint32_t* work = &Q.counters[Q.work_base + (size_t)s * Q.ctr_pad];
                        const uint32_t chunk = min(PT_SCHED_CHUNK, max(64u, (n_s / waves_per_shard + 63u) & ~63u));
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(work, chunk);
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (base < n_s) {
                            wcur = s * Q.shard_cap + base;
                            wend = s * Q.shard_cap + (base + chunk < n_s ? base + chunk : n_s);
                            got = true;
                            break;
                        }
                        avail &= ~(1ull << s);
                    }
                    if (!got) { drained = true; break; }
                }
                const uint32_t n_idle = __popcll(idle), left = wend - wcur;
                const uint32_t take = n_idle < left ? n_idle : left;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));   // set bits of idle below this lane
                if (node == PT_NODE_DONE && rank < take) {
                    const uint32_t slot = wcur + rank;
                    const float4 o = pt_stream_load(&q0[slot]), d = pt_stream_load(&q1[slot]);
                    start_ray(o, d, (int)slot);
                }
                wcur += take;
                idle = __ballot(node == PT_NODE_DONE);
            }
        }
        bool want_node = (unsigned)node < (unsigned)PT_NODE_DONE;
        bool want_leaf = node < 0 && cur_inst != PT_NO_INSTANCE;
        const bool want_enter = (node < 0 && cur_inst == PT_NO_INSTANCE) || node == PT_NODE_SENTINEL;   // enters or leaves an instance
        int n_node = __popcll(__ballot(want_node));
        int n_leaf = __popcll(__ballot(want_leaf));
        const int n_enter = __popcll(__ballot(want_enter));
        if (n_node + n_leaf + n_enter == 0) {
            if (drained) { flush(); break; }
            continue;
        }
        // ------------------------------------------------------------------ vote for one phase
        int phase;   // 0 box step, 1 triangle, 2 instance entry
        if (n_leaf >= PT_SCHED_LEAF_VOTES || (n_leaf > 0 && n_node < PT_SCHED_NODE_MIN && n_leaf >= n_enter)) phase = 1;
        else if (n_enter >= PT_SCHED_ENTER_VOTES || (n_enter > 0 && n_node < PT_SCHED_NODE_MIN)) phase = 2;
        else if (n_node > 0) phase = 0;
        else phase = n_leaf >= n_enter ? 1 : 2;

        PT_STAT(6, 1);
        if (phase == 0) {
            int steps = PT_SCHED_NODE_STEPS;
            do {
                { const int act_ = __popcll(__ballot(want_node)); (void)act_; PT_STAT(0, 1); PT_STAT(3, act_); }
                if (want_node) {
                    node = pt_wide_step<false>(cur, tl.nodes, node, tlo, fminf(thi, best_t), stk);
                    want_node = (unsigned)node < (unsigned)PT_NODE_DONE;
                }
                n_node = __popcll(__ballot(want_node));
            } while (--steps > 0 && n_node >= PT_SCHED_NODE_STAY);
        } else if (phase == 1) {
            int steps = PT_SCHED_LEAF_STEPS;
            do {
                { const int act_ = __popcll(__ballot(want_leaf)); (void)act_; PT_STAT(1, 1); PT_STAT(4, act_); }
                if (want_leaf) {
                    const uint32_t code = ~(uint32_t)node;
                    const uint32_t pos = code >> 3;
                    const float4 a = tl.tris[3 * (size_t)pos + 0], b = tl.tris[3 * (size_t)pos + 1], c = tl.tris[3 * (size_t)pos + 2];
                    const float delta = __int_as_float(PT_COLD(5));
                    float t, u, v;
                    // inside an instance the direction's registers hold the triangle test's shear and axes (set on entry)
                    PtTriSetup S;
                    S.Sx = cur.dx; S.Sy = cur.dy;
                    const uint32_t kk = __float_as_uint(cur.dz);
                    S.kz = (int)(kk & 3u); S.kx = (int)((kk >> 2) & 3u); S.ky = (int)(kk >> 4);
                    S.Sz = pt_sel3(S.kz, cur.ix, cur.iy, cur.iz);
                    if (pt_tri_test_with(S, cur, delta, a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, t, u, v)) {
                        const uint32_t prim = __float_as_uint(c.y);
                        if (t > tmin && t < tmax && t >= tlo && t <= thi) {
                            // the order of pt_hit_update; the ids are only looked at on an exact tie
                            bool closer = t < best_t;
                            if (t == best_t) {
                                const unsigned long long id = ((unsigned long long)cur_inst << 32) | prim;
                                const unsigned long long best_id = ((unsigned long long)(uint32_t)PT_COLD(0) << 32) | (uint32_t)PT_COLD(1);
                                closer = id < best_id;
                            }
                            if (closer) { best_t = t; PT_COLD(0) = (int)cur_inst; PT_COLD(1) = (int)prim; PT_COLD(2) = __float_as_int(u); PT_COLD(3) = __float_as_int(v); }
                        }
                    }
                    // one triangle per step: first + 1, count - 1
                    if (code & 7u) node = (int)~(code + 7u);
                    else node = stk.pop();
                    want_leaf = node < 0 && cur_inst != PT_NO_INSTANCE;
                }
                n_leaf = __popcll(__ballot(want_leaf));
            } while (--steps > 0 && n_leaf >= PT_SCHED_LEAF_VOTES / 2);
        } else {
            PT_STAT(2, 1); PT_STAT(5, n_enter);
            if (node == PT_NODE_SENTINEL) {        // back to world space; something is left on the stack
                cur.ox = ws[0 * PT_TRAVERSE_BLOCK]; cur.oy = ws[1 * PT_TRAVERSE_BLOCK]; cur.oz = ws[2 * PT_TRAVERSE_BLOCK];
                cur.dx = ws[3 * PT_TRAVERSE_BLOCK]; cur.dy = ws[4 * PT_TRAVERSE_BLOCK]; cur.dz = ws[5 * PT_TRAVERSE_BLOCK];
                cur.ix = ws[6 * PT_TRAVERSE_BLOCK]; cur.iy = ws[7 * PT_TRAVERSE_BLOCK]; cur.iz = ws[8 * PT_TRAVERSE_BLOCK];
                cur.oox = cur.ox * cur.ix; cur.ooy = cur.oy * cur.iy; cur.ooz = cur.oz * cur.iz;
                cur_inst = PT_NO_INSTANCE; tlo = tmin; thi = tmax;
                node = stk.pop();
            }
            if (node < 0 && cur_inst == PT_NO_INSTANCE) {
                // the whole record is requested at once: one round trip, not one per use
                const float4* q = tl.inst[(~(uint32_t)node) >> 3].q;
                const float4 r0 = q[0], r1 = q[1], r2 = q[2], blo = q[3], bhi = q[4], misc = q[5];
                float a, b;
                pt_slab(cur, blo.x, blo.y, blo.z, bhi.x, bhi.y, bhi.z, a, b);
                bool enter = (a <= b) && !(b < tmin) && !(a > tmax) && !(a > best_t);
                if (enter) {
                    float oox, ooy, ooz, odx, ody, odz;
                    pt_to_object_space(r0, r1, r2, cur, oox, ooy, ooz, odx, ody, odz);
                    enter = pt_ray_valid(odx, ody, odz);
                    if (enter) {
                        pt_ray_space_init(cur, oox, ooy, ooz, odx, ody, odz);
                        // The box steps inside the instance read the reciprocals only, and the direction comes back
                        // from LDS when the lane leaves: until then its three registers carry what the triangle
                        // test derives from the ray alone, worked out once per entry instead of once per test.
                        const PtTriSetup S = pt_tri_setup(odx, ody, odz, cur.ix, cur.iy, cur.iz);
                        cur.dx = S.Sx; cur.dy = S.Sy; cur.dz = __uint_as_float((uint32_t)S.kz | ((uint32_t)S.kx << 2) | ((uint32_t)S.ky << 4));
                        PT_COLD(5) = __float_as_int(blo.w);
                        cur_inst = __float_as_uint(misc.y);
                        tlo = fmaxf(tmin, a); thi = fminf(tmax, b);
                        if (stk.sp > 0) stk.push(PT_NODE_SENTINEL);
                        node = __float_as_int(misc.x);
                    }
                }
                if (!enter) node = stk.pop();
            }
        }
    }
}
