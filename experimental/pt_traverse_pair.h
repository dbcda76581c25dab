// pt_traverse_pair.h -- wave-scheduled persistent traversal with two rays per lane.
//
// The wave-scheduled traversal of pt_traverse_sched.h executes one phase at a time (box step, triangle test,
// instance entry / exit) and a lane takes part when its ray is in that phase: measured on the incoherent
// bounces of the dragon frame, 38 of 64 lanes in box steps, 16 in triangle tests, 24 in the instance phase.
// The idle lanes are lanes whose one ray wants another phase.  Here every lane holds TWO rays, both in
// registers, and takes part in a phase when either of them wants it: the one that does is brought into
// the working set by exchanging the two register sets (twenty v_swap_b32 for the lanes that need it), so
// with independent phases a lane that was busy with probability p is busy with 1 - (1 - p)^2.
// What pays for it: the second register set (about 112 registers: four waves per SIMD instead of five)
// and LDS for two rays per lane, which is cut to twenty words per ray to keep four blocks per CU:
//   stack   PT_PAIR_STACK entries (deeper entries go to a per-ray slice of a global buffer)
//   world   origin and direction of the ray while it is inside an instance; the reciprocals are divided
//           out again when it leaves
//   cold    ids and barycentrics of the closest hit so far
// all [word][thread] behind one base pointer per ray, so that exchanging the rays exchanges one pointer.
// The per-lane arithmetic (slab test, gated watertight triangle test, hit ordering) is the one in
// pt_device.h, so results are bit-identical to the other schedules.
//
// Measured (dragon 1920x1080, 64 spp per step, bounces after the first): lanes active per wave iteration
// 49.8 / 25.8 / 31.6 in box / triangle / instance iterations against 41.0 / 22.2 / 24.3 with one ray per
// lane, 16 % fewer box iterations and 12 % fewer triangle iterations per ray -- and a launch that takes
// 10.4 ms instead of 8.1 ms at 256 spp per step: an iteration costs half as much again (the exchange,
// twice the votes and refill passes, one wave fewer per SIMD, and a ten-entry LDS stack whose overflow
// path some lane takes in most iterations).  It is therefore not the default: PT_HIP_TRAVERSAL=pair
// selects it, and the parity tests cover it like the other schedules.
#pragma once
#include "pt_traverse_sched.h"

#ifndef PT_PAIR_STACK
#define PT_PAIR_STACK 10
#endif
#define PT_PAIR_WS (PT_PAIR_STACK)            // word index of the world-space ray (6 words)
#define PT_PAIR_COLD (PT_PAIR_STACK + 6)      // ... of the closest hit's instance, primitive, u, v
#define PT_PAIR_WORDS (PT_PAIR_STACK + 10)    // LDS words per ray
#ifndef PT_PAIR_REFILL
#define PT_PAIR_REFILL 40                     // refill when at least this many of the wave's 128 ray slots are idle
#endif
#ifndef PT_PAIR_LEAF_VOTES
#define PT_PAIR_LEAF_VOTES 32
#endif
#ifndef PT_PAIR_ENTER_VOTES
#define PT_PAIR_ENTER_VOTES 20
#endif
#ifndef PT_PAIR_NODE_MIN
#define PT_PAIR_NODE_MIN 36
#endif
#ifndef PT_PAIR_NODE_STEPS
#define PT_PAIR_NODE_STEPS 6
#endif
#ifndef PT_PAIR_NODE_STAY
#define PT_PAIR_NODE_STAY 44
#endif
#ifndef PT_PAIR_LEAF_STEPS
#define PT_PAIR_LEAF_STEPS 4
#endif

typedef PtStackT<PT_PAIR_STACK, 2> PtPairStack;

struct PtPairRay {                 // what lives in registers of one ray
    float ox, oy, oz, dx, dy, dz, ix, iy, iz;   // the current ray space; o * (1 / d) is multiplied out where a step needs it
    float best_t, tlo, thi, delta;
    uint32_t cur_inst;
    int node;                      // PT_NODE_DONE: the slot holds no ray in flight
    int sp;
    int qslot;                     // queue slot of a ray whose hit record is still to be written, or -1
};

template <bool BY_ID>
__device__ __forceinline__ void pt_traverse_pair(const PtTlasView& tl, const PtSchedQueue& Q, pt_lds_int* lds_base, int* spill,
                                                 const float4* __restrict__ q0, const float4* __restrict__ q1,
                                                 float tmin, float tmax, PtHit* __restrict__ hits, float* __restrict__ t_out)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    constexpr int kRegion = PT_PAIR_WORDS * PT_TRAVERSE_BLOCK;      // words between the two rays' LDS regions

    // ---- wave-uniform work cursor (as in pt_traverse_sched)
    const uint32_t ns_lane = lane < Q.n_shards ? Q.counters[Q.count_base + (size_t)lane * Q.ctr_pad] : 0u;
    const uint32_t wk_lane = lane < Q.n_shards
        ? __hip_atomic_load(&Q.counters[Q.work_base + (size_t)lane * Q.ctr_pad], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    unsigned long long avail = __ballot(wk_lane < ns_lane);
    const uint32_t start = wave & 63u;
    uint32_t wcur = 0, wend = 0;
    bool drained = avail == 0ull;
    const uint32_t waves_per_shard = max(1u, (gridDim.x * (PT_TRAVERSE_BLOCK / 64u)) / Q.n_shards);

    // ---- two rays per lane: A is the one in hand, B the other
    PtPairRay A, B;
    A.ox = A.oy = A.oz = A.dx = A.dy = A.dz = A.ix = A.iy = A.iz = 0.0f;
    A.best_t = tmax; A.tlo = tmin; A.thi = tmax; A.delta = 0.0f; A.cur_inst = PT_NO_INSTANCE; A.node = PT_NODE_DONE; A.sp = 0; A.qslot = -1;
    B = A;
    PtPairStack stk;               // stack of the ray in hand; its sp is A.sp (copied in and out around the steps)
    stk.lds = lds_base; stk.spill_base = spill; stk.sp = 0; stk.slice = 0;
#define PT_PW(k) stk.lds[(k) * PT_TRAVERSE_BLOCK]                     // LDS word k of the ray in hand
#define PT_PWF(k) ((pt_lds_float*)stk.lds)[(k) * PT_TRAVERSE_BLOCK]

    // exchanges the two register sets in the lanes that ask for it: one v_swap_b32 per register
#define PT_SWAPF(a, b) asm volatile("v_swap_b32 %0, %1" : "+v"(a), "+v"(b))
    auto swap_if = [&](bool c) {
        if (c) {
            PT_SWAPF(A.ox, B.ox); PT_SWAPF(A.oy, B.oy); PT_SWAPF(A.oz, B.oz); PT_SWAPF(A.dx, B.dx); PT_SWAPF(A.dy, B.dy); PT_SWAPF(A.dz, B.dz);
            PT_SWAPF(A.ix, B.ix); PT_SWAPF(A.iy, B.iy); PT_SWAPF(A.iz, B.iz);
            PT_SWAPF(A.best_t, B.best_t); PT_SWAPF(A.tlo, B.tlo); PT_SWAPF(A.thi, B.thi); PT_SWAPF(A.delta, B.delta);
            PT_SWAPF(A.cur_inst, B.cur_inst); PT_SWAPF(A.node, B.node); PT_SWAPF(A.sp, B.sp); PT_SWAPF(A.qslot, B.qslot);
            stk.lds = stk.slice ? stk.lds - kRegion : stk.lds + kRegion;
            stk.slice ^= 1;
        }
    };
    // the ray in hand in the form the steps of pt_device.h take
    auto space = [&]() {
        PtRaySpace r;
        r.ox = A.ox; r.oy = A.oy; r.oz = A.oz; r.dx = A.dx; r.dy = A.dy; r.dz = A.dz; r.ix = A.ix; r.iy = A.iy; r.iz = A.iz;
        r.oox = A.ox * A.ix; r.ooy = A.oy * A.iy; r.ooz = A.oz * A.iz;
        return r;
    };
    auto set_space = [&](float ox, float oy, float oz, float dx, float dy, float dz) {
        A.ox = ox; A.oy = oy; A.oz = oz; A.dx = dx; A.dy = dy; A.dz = dz;
        A.ix = pt_safe_rcp(dx); A.iy = pt_safe_rcp(dy); A.iz = pt_safe_rcp(dz);
    };
    auto want_node = [&](const PtPairRay& r) { return (unsigned)r.node < (unsigned)PT_NODE_DONE; };
    auto want_leaf = [&](const PtPairRay& r) { return r.node < 0 && r.cur_inst != PT_NO_INSTANCE; };
    auto want_enter = [&](const PtPairRay& r) { return (r.node < 0 && r.cur_inst == PT_NO_INSTANCE) || r.node == PT_NODE_SENTINEL; };
    // the next item of the ray in hand after the current one is exhausted
    auto advance = [&](int v) {
        if (v == PT_NODE_SENTINEL && stk.sp == 0) v = PT_NODE_DONE;   // nothing left above the instance: the ray is finished
        A.node = v;
    };
    // writes the hit record of a finished ray in hand (one pass serves every lane that waits)
    auto flush = [&]() {
        if (A.node == PT_NODE_DONE && A.qslot != -1) {
            const uint32_t qi = (uint32_t)A.qslot;
            PtHit out; out.instance = (uint32_t)PT_PW(PT_PAIR_COLD + 0); out.primitive = (uint32_t)PT_PW(PT_PAIR_COLD + 1);
            out.u = __int_as_float(PT_PW(PT_PAIR_COLD + 2)); out.v = __int_as_float(PT_PW(PT_PAIR_COLD + 3));
            const uint32_t dst = BY_ID ? __float_as_uint(q1[qi].w) : qi;
            if (BY_ID) hits[dst] = out;
            else pt_stream_store((float4*)&hits[dst], make_float4(__uint_as_float(out.instance), __uint_as_float(out.primitive), out.u, out.v));
            if (BY_ID && t_out) t_out[dst] = A.best_t;
            A.qslot = -1;
        }
    };
    // gives the idle rays in hand the next rays of the wave's chunk
    auto refill = [&]() {
        unsigned long long idle = __ballot(A.node == PT_NODE_DONE);
        while (idle && !drained) {
            if (wcur == wend) {
                bool got = false;
                while (avail) {
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
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            if (A.node == PT_NODE_DONE && rank < take) {
                const uint32_t slot = wcur + rank;
                A.qslot = (int)slot;
                PT_PW(PT_PAIR_COLD + 0) = (int)PT_NO_INSTANCE; PT_PW(PT_PAIR_COLD + 1) = -1; PT_PW(PT_PAIR_COLD + 2) = 0; PT_PW(PT_PAIR_COLD + 3) = 0;
                const float4 o = pt_stream_load(&q0[slot]), d = pt_stream_load(&q1[slot]);
                set_space(o.x, o.y, o.z, d.x, d.y, d.z);
                PT_PWF(PT_PAIR_WS + 0) = o.x; PT_PWF(PT_PAIR_WS + 1) = o.y; PT_PWF(PT_PAIR_WS + 2) = o.z;
                PT_PWF(PT_PAIR_WS + 3) = d.x; PT_PWF(PT_PAIR_WS + 4) = d.y; PT_PWF(PT_PAIR_WS + 5) = d.z;
                A.cur_inst = PT_NO_INSTANCE; A.tlo = tmin; A.thi = tmax; A.best_t = tmax; A.delta = 0.0f;
                A.sp = 0;
                A.node = 0;
            }
            wcur += take;
            idle = __ballot(A.node == PT_NODE_DONE);
        }
    };

    for (;;) {
        // ------------------------------------------------------------------ refill idle ray slots
        {
            const int n_idle = __popcll(__ballot(A.node == PT_NODE_DONE)) + __popcll(__ballot(B.node == PT_NODE_DONE));
            if (n_idle == 128 || n_idle >= PT_PAIR_REFILL) {
                flush();
                if (!drained) refill();
                // the other ray's turn: an idle or finished B comes into hand
                swap_if(B.node == PT_NODE_DONE);
                flush();
                if (!drained) refill();
            }
        }
        const bool nA = want_node(A), nB = want_node(B), lA = want_leaf(A), lB = want_leaf(B), eA = want_enter(A), eB = want_enter(B);
        int n_node = __popcll(__ballot(nA | nB));
        int n_leaf = __popcll(__ballot(lA | lB));
        const int n_enter = __popcll(__ballot(eA | eB));
        if (n_node + n_leaf + n_enter == 0) {
            if (drained) { flush(); swap_if(true); flush(); break; }
            continue;
        }
        // ------------------------------------------------------------------ vote for one phase
        int phase;   // 0 box step, 1 triangle, 2 instance entry / exit
        if (n_leaf >= PT_PAIR_LEAF_VOTES || (n_leaf > 0 && n_node < PT_PAIR_NODE_MIN && n_leaf >= n_enter)) phase = 1;
        else if (n_enter >= PT_PAIR_ENTER_VOTES || (n_enter > 0 && n_node < PT_PAIR_NODE_MIN)) phase = 2;
        else if (n_node > 0) phase = 0;
        else phase = n_leaf >= n_enter ? 1 : 2;

        PT_STAT(6, 1);
        if (phase == 0) {
            int steps = PT_PAIR_NODE_STEPS;
            do {
                swap_if(!want_node(A) && want_node(B));
                const bool act = want_node(A);
                { const int act_ = __popcll(__ballot(act)); (void)act_; PT_STAT(0, 1); PT_STAT(3, act_); }
                if (act) {
                    stk.sp = A.sp;
                    const int next = pt_wide_step<false>(space(), tl.nodes, A.node, A.tlo, fminf(A.thi, A.best_t), stk);
                    if ((unsigned)next < (unsigned)PT_NODE_DONE) A.node = next; else advance(next);
                    A.sp = stk.sp;
                }
                n_node = __popcll(__ballot(want_node(A) || want_node(B)));
            } while (--steps > 0 && n_node >= PT_PAIR_NODE_STAY);
        } else if (phase == 1) {
            int steps = PT_PAIR_LEAF_STEPS;
            do {
                swap_if(!want_leaf(A) && want_leaf(B));
                const bool act = want_leaf(A);
                { const int act_ = __popcll(__ballot(act)); (void)act_; PT_STAT(1, 1); PT_STAT(4, act_); }
                if (act) {
                    const uint32_t code = ~(uint32_t)A.node;
                    const uint32_t pos = code >> 3;
                    const float4 a = tl.tris[3 * (size_t)pos + 0], b = tl.tris[3 * (size_t)pos + 1], c = tl.tris[3 * (size_t)pos + 2];
                    float t, u, v;
                    if (pt_tri_test(space(), A.delta, a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, t, u, v)) {
                        const uint32_t prim = __float_as_uint(c.y);
                        if (t > tmin && t < tmax && t >= A.tlo && t <= A.thi) {
                            // the order of pt_hit_update; the ids are only looked at on an exact tie
                            bool closer = t < A.best_t;
                            if (t == A.best_t) {
                                const unsigned long long id = ((unsigned long long)A.cur_inst << 32) | prim;
                                const unsigned long long best_id = ((unsigned long long)(uint32_t)PT_PW(PT_PAIR_COLD + 0) << 32) | (uint32_t)PT_PW(PT_PAIR_COLD + 1);
                                closer = id < best_id;
                            }
                            if (closer) {
                                A.best_t = t;
                                PT_PW(PT_PAIR_COLD + 0) = (int)A.cur_inst; PT_PW(PT_PAIR_COLD + 1) = (int)prim;
                                PT_PW(PT_PAIR_COLD + 2) = __float_as_int(u); PT_PW(PT_PAIR_COLD + 3) = __float_as_int(v);
                            }
                        }
                    }
                    // one triangle per step: first + 1, count - 1
                    if (code & 7u) A.node = (int)~(code + 7u);
                    else { stk.sp = A.sp; advance(stk.pop()); A.sp = stk.sp; }
                }
                n_leaf = __popcll(__ballot(want_leaf(A) || want_leaf(B)));
            } while (--steps > 0 && n_leaf >= PT_PAIR_LEAF_VOTES / 2);
        } else {
            swap_if(!want_enter(A) && want_enter(B));
            { const int act_ = __popcll(__ballot(want_enter(A))); (void)act_; PT_STAT(2, 1); PT_STAT(5, act_); }
            stk.sp = A.sp;
            if (A.node == PT_NODE_SENTINEL) {        // back to world space; something is left on the stack
                const float ox = PT_PWF(PT_PAIR_WS + 0), oy = PT_PWF(PT_PAIR_WS + 1), oz = PT_PWF(PT_PAIR_WS + 2);
                const float dx = PT_PWF(PT_PAIR_WS + 3), dy = PT_PWF(PT_PAIR_WS + 4), dz = PT_PWF(PT_PAIR_WS + 5);
                set_space(ox, oy, oz, dx, dy, dz);
                A.cur_inst = PT_NO_INSTANCE; A.tlo = tmin; A.thi = tmax;
                A.node = stk.pop();
            }
            if (A.node < 0 && A.cur_inst == PT_NO_INSTANCE) {
                // the whole record is requested at once: one round trip, not one per use
                const float4* q = tl.inst[(~(uint32_t)A.node) >> 3].q;
                const float4 r0 = q[0], r1 = q[1], r2 = q[2], blo = q[3], bhi = q[4], misc = q[5];
                float a, b;
                const PtRaySpace ws = space();
                pt_slab(ws, blo.x, blo.y, blo.z, bhi.x, bhi.y, bhi.z, a, b);
                bool enter = (a <= b) && !(b < tmin) && !(a > tmax) && !(a > A.best_t);
                if (enter) {
                    float oox, ooy, ooz, odx, ody, odz;
                    pt_to_object_space(r0, r1, r2, ws, oox, ooy, ooz, odx, ody, odz);
                    enter = pt_ray_valid(odx, ody, odz);
                    if (enter) {
                        set_space(oox, ooy, ooz, odx, ody, odz);
                        A.delta = blo.w;
                        A.cur_inst = __float_as_uint(misc.y);
                        A.tlo = fmaxf(tmin, a); A.thi = fminf(tmax, b);
                        stk.push(PT_NODE_SENTINEL);
                        A.node = __float_as_int(misc.x);
                    }
                }
                if (!enter) advance(stk.pop());
            }
            A.sp = stk.sp;
        }
    }
#undef PT_PW
#undef PT_PWF
#undef PT_SWAPF
}
