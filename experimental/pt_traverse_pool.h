// pt_traverse_pool.h -- wave-scheduled persistent traversal over a pool of rays kept in LDS.
//
// The wave-scheduled traversal of pt_traverse_sched.h keeps one ray per lane in registers and runs one phase at
// a time (box step, triangle test, instance entry) for the lanes that want it: measured on incoherent bounces,
// 37 of 64 lanes take part in a box step, 24 in a triangle test and 27 in an instance entry, because a wave's 64
// rays are never all in the same phase.  Here a wave owns 32 * PT_POOL_K ray slots (several per lane) whose whole
// state lives in LDS, [word][slot], and a lane is a worker without a ray of its own.  Every iteration the wave
//   1. picks the phase most of its slots wait for (refilling idle slots from the queue is one of the phases; the
//      counts are kept in scalar registers and updated from ballots),
//   2. every lane looks for a slot of that phase among the PT_POOL_K slots of its bank class,
//   3. loads what the phase needs of that slot, executes one step, stores what changed.
// Bank classes: slot s lives in LDS bank s mod 32 (the word stride is a multiple of 32), and lanes l and l + 32
// serve the slots with s mod 32 == l mod 32, the lower lane taking the first candidate and the upper the second.
// A 32-lane half therefore touches 32 different banks in every access: no bank conflicts by construction.  (A
// first version that dealt any slot to any lane through a list spent more LDS cycles in conflicts than in
// accesses.)  The phases of a class's slots sit in one word, four one-hot bits each, so a lane finds its slot
// with one read and changes a phase with one atomic xor.
// The closest hit's record is written to memory whenever it improves (a ray finds one or two improvements on
// average) instead of being kept in LDS: only the distance stays, and the rare exact tie reads the ids back.
// A ray that leaves an instance reads its world-space form back from the queue.
// The per-lane arithmetic (slab test, gated watertight triangle test, hit ordering, ray transformation) is the
// one in pt_device.h, applied to the same values in the same order, so results are bit-identical to the other
// schedules.
#pragma once
#include "pt_traverse_sched.h"

#ifndef PT_POOL_K
#define PT_POOL_K 5                // slots per bank class: a wave holds 32 * PT_POOL_K rays (2.5 per lane)
#endif
#ifndef PT_POOL_STACK
#define PT_POOL_STACK 8            // stack entries per slot in LDS; deeper entries go to the slot's slice of a global buffer
#endif
#ifndef PT_POOL_WAVES
#define PT_POOL_WAVES 2            // waves per SIMD the kernel is compiled for (LDS allows two blocks per CU at the defaults)
#endif
#define PT_POOL_RAYS (32 * PT_POOL_K)
#define PT_POOL_COLS (4 * PT_POOL_RAYS)         // slots per block: wave w of the block owns columns [w * PT_POOL_RAYS, (w + 1) * PT_POOL_RAYS)

enum {
    PW_OX = 0, PW_OY, PW_OZ, PW_DX, PW_DY, PW_DZ, PW_IX, PW_IY, PW_IZ,      // ray in the current space: origin, direction, clamped reciprocals
    PW_BEST, PW_TLO, PW_THI,                                             // closest distance so far; interval handed to the box tests
    PW_NODE, PW_SP, PW_INST, PW_SLOT, PW_DELTA,                          // current item, stack height, instance (or PT_NO_INSTANCE), queue slot, instance delta
    PW_STACK                                                             // first of PT_POOL_STACK stack entries
};
#define PT_POOL_WORDS (PW_STACK + PT_POOL_STACK)
static_assert(PT_POOL_WORDS * PT_POOL_COLS * 4 < 65536, "LDS offsets of a slot's words must fit the 16-bit immediate");
static_assert(PT_POOL_K >= 2 && PT_POOL_K <= 8, "a class word holds eight four-bit phases");

// one-hot phase of a slot inside its class word
enum { PT_PH_BOX = 1, PT_PH_TRI = 2, PT_PH_INST = 4, PT_PH_IDLE = 8 };

// The stack of the slot a lane works on: PT_POOL_STACK entries in the slot's LDS column, then the slot's global slice.
struct PtPoolStack {
    pt_lds_int* lds;   // entry 0 of the slot
    int* spill;        // the slot's slice of the overflow buffer
    int sp;
    __device__ __forceinline__ void push(int v)
    {
        if (sp < PT_POOL_STACK) lds[sp * PT_POOL_COLS] = v;
        else spill[sp - PT_POOL_STACK] = v;
        ++sp;
    }
    // Pushes r[m], ..., r[1] in this order (m in 0..3) so that r[1] is popped first.
    __device__ __forceinline__ void push_far_to_near(int r1, int r2, int r3, int m)
    {
        if (sp + 3 <= PT_POOL_STACK) {
            lds[(sp + max(m - 3, 0)) * PT_POOL_COLS] = r3;
            lds[(sp + max(m - 2, 0)) * PT_POOL_COLS] = r2;
            lds[(sp + max(m - 1, 0)) * PT_POOL_COLS] = r1;
            sp += m;
        } else {
            if (m > 2) push(r3);
            if (m > 1) push(r2);
            if (m > 0) push(r1);
        }
    }
    __device__ __forceinline__ int pop()
    {
        if (sp == 0) return PT_NODE_DONE;
        --sp;
        return sp < PT_POOL_STACK ? lds[sp * PT_POOL_COLS] : spill[sp - PT_POOL_STACK];
    }
};

__device__ __forceinline__ uint32_t pt_pool_phase_of(int node, uint32_t inst)
{
    if ((unsigned)node < (unsigned)PT_NODE_DONE) return PT_PH_BOX;
    if (node == PT_NODE_DONE) return PT_PH_IDLE;
    if (node == PT_NODE_SENTINEL) return PT_PH_INST;
    return inst == PT_NO_INSTANCE ? PT_PH_INST : PT_PH_TRI;
}

__device__ __forceinline__ void pt_pool_wave_sync()
{
    // what one lane wrote to LDS is read by another lane of the same wave afterwards
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// wpool: column of slot 0 of this wave inside the block's pool; cls_words: this wave's 32 class words;
// spill: slice of slot 0 of this wave (PT_SPILL_STACK ints per slot).
template <bool BY_ID>
__device__ __forceinline__ void pt_traverse_pool(const PtTlasView& tl, const PtSchedQueue& Q, pt_lds_int* wpool, pt_lds_int* cls_words, int* spill,
                                                 const float4* __restrict__ q0, const float4* __restrict__ q1,
                                                 float tmin, float tmax, PtHit* __restrict__ hits, float* __restrict__ t_out)
{
    constexpr int K = PT_POOL_K;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t cls = lane & 31u;
    const bool upper = lane >= 32u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;

    // ---- wave-uniform work cursor over the sharded queue (as in pt_traverse_sched)
    const uint32_t ns_lane = lane < Q.n_shards ? Q.counters[Q.count_base + (size_t)lane * Q.ctr_pad] : 0u;
    const uint32_t wk_lane = lane < Q.n_shards
        ? __hip_atomic_load(&Q.counters[Q.work_base + (size_t)lane * Q.ctr_pad], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    unsigned long long avail = __ballot(wk_lane < ns_lane);
    const uint32_t start = wave & 63u;
    uint32_t wcur = 0, wend = 0;
    bool drained = avail == 0ull;
    const uint32_t waves_per_shard = max(1u, (gridDim.x * (PT_TRAVERSE_BLOCK / 64u)) / Q.n_shards);

#define PT_PW(slot, w) wpool[(w) * PT_POOL_COLS + (slot)]
#define PT_PWF(slot, w) (((pt_lds_float*)wpool)[(w) * PT_POOL_COLS + (slot)])

    // What a lane continues with after its item is exhausted: leaving an instance with nothing above it ends the ray.
    auto advance = [](int v, int sp) { return (v == PT_NODE_SENTINEL && sp == 0) ? PT_NODE_DONE : v; };
    // The record of slot s's ray where the next kernel (or the caller) looks for it.
    // (caller-ordered results: the ray's index travels in the spare word of its direction)
    auto hit_index = [&](int s) -> uint32_t { const uint32_t qs = (uint32_t)PT_PW(s, PW_SLOT); return BY_ID ? __float_as_uint(q1[qs].w) : qs; };
    auto store_hit = [&](uint32_t dst, uint32_t inst, uint32_t prim, float u, float v, float t) {
        PtHit out; out.instance = inst; out.primitive = prim; out.u = u; out.v = v;
        hits[dst] = out;
        if (BY_ID && t_out) t_out[dst] = t;
    };

    // every slot idle
    if (!upper) {
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) w |= (uint32_t)PT_PH_IDLE << (4 * k);
        cls_words[cls] = (int)w;
    }
    int nB = 0, nT = 0, nI = 0, nD = PT_POOL_RAYS;      // slots per phase (wave-uniform)
    bool flip = false;
    for (;;) {
        pt_pool_wave_sync();
        if (nB + nT + nI == 0 && (drained || nD == 0)) break;
        // ------------------------------------------------------------------ the phase most slots wait for (a wave serves at most 64 of them)
        uint32_t phase = PT_PH_BOX; int most = min(nB, 64);
        if (min(nT, 64) > most) { phase = PT_PH_TRI; most = min(nT, 64); }
        if (min(nI, 64) > most) { phase = PT_PH_INST; most = min(nI, 64); }
        if (!drained && min(nD, 64) > most) { phase = PT_PH_IDLE; most = min(nD, 64); }
        // ------------------------------------------------------------------ a slot of that phase in this lane's class
        const uint32_t cw = (uint32_t)cls_words[cls];
        uint32_t m = cw & (phase * 0x11111111u);          // bit 4k (+ log2 phase) set: slot k waits for this phase
        // the lower lane takes the first candidate and the upper lane the second; which end counts as first alternates,
        // so that no slot waits for ever behind its neighbours
        int bit;
        if (!flip) { if (upper) m &= m - 1u; bit = m ? __builtin_ctz(m) : 0; }
        else { if (upper && m) m &= ~(0x80000000u >> __builtin_clz(m)); bit = m ? 31 - __builtin_clz(m) : 0; }
        flip = !flip;
        const bool active = m != 0u;
        const int k_slot = bit >> 2;
        const int s = (int)cls + 32 * k_slot;
        const int n_sel = __popcll(__ballot(active));
        uint32_t new_phase = phase;       // of this lane's slot after the step
        PT_STAT(6, 1);

        if (phase == PT_PH_BOX) {
            PT_STAT(0, 1); PT_STAT(3, n_sel);
            if (active) {
                PtRaySpace cur;
                cur.ox = PT_PWF(s, PW_OX); cur.oy = PT_PWF(s, PW_OY); cur.oz = PT_PWF(s, PW_OZ);
                cur.ix = PT_PWF(s, PW_IX); cur.iy = PT_PWF(s, PW_IY); cur.iz = PT_PWF(s, PW_IZ);
                cur.oox = cur.ox * cur.ix; cur.ooy = cur.oy * cur.iy; cur.ooz = cur.oz * cur.iz;
                const float best_t = PT_PWF(s, PW_BEST), tlo = PT_PWF(s, PW_TLO), thi = PT_PWF(s, PW_THI);
                const int node = PT_PW(s, PW_NODE);
                const uint32_t inst = (uint32_t)PT_PW(s, PW_INST);
                PtPoolStack stk;
                stk.lds = &PT_PW(s, PW_STACK); stk.spill = spill + (size_t)s * PT_SPILL_STACK; stk.sp = PT_PW(s, PW_SP);
                int next = pt_wide_step<false>(cur, tl.nodes, node, tlo, fminf(thi, best_t), stk);
                if (!((unsigned)next < (unsigned)PT_NODE_DONE)) next = advance(next, stk.sp);
                PT_PW(s, PW_NODE) = next; PT_PW(s, PW_SP) = stk.sp;
                new_phase = pt_pool_phase_of(next, inst);
            }
        } else if (phase == PT_PH_TRI) {
            PT_STAT(1, 1); PT_STAT(4, n_sel);
            if (active) {
                PtRaySpace cur;
                cur.ox = PT_PWF(s, PW_OX); cur.oy = PT_PWF(s, PW_OY); cur.oz = PT_PWF(s, PW_OZ);
                cur.dx = PT_PWF(s, PW_DX); cur.dy = PT_PWF(s, PW_DY); cur.dz = PT_PWF(s, PW_DZ);
                cur.ix = PT_PWF(s, PW_IX); cur.iy = PT_PWF(s, PW_IY); cur.iz = PT_PWF(s, PW_IZ);
                cur.oox = cur.ox * cur.ix; cur.ooy = cur.oy * cur.iy; cur.ooz = cur.oz * cur.iz;
                const float best_t = PT_PWF(s, PW_BEST), tlo = PT_PWF(s, PW_TLO), thi = PT_PWF(s, PW_THI), delta = PT_PWF(s, PW_DELTA);
                int node = PT_PW(s, PW_NODE);
                const uint32_t inst = (uint32_t)PT_PW(s, PW_INST);
                const uint32_t code = ~(uint32_t)node;
                const uint32_t pos = code >> 3;
                const float4 a = tl.tris[3 * (size_t)pos + 0], b = tl.tris[3 * (size_t)pos + 1], c = tl.tris[3 * (size_t)pos + 2];
                float t, u, v;
                if (pt_tri_test(cur, delta, a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, t, u, v)) {
                    const uint32_t prim = __float_as_uint(c.y);
                    if (t > tmin && t < tmax && t >= tlo && t <= thi) {
                        // the order of pt_hit_update; the ids are only looked at on an exact tie
                        bool closer = t < best_t;
                        const uint32_t dst = hit_index(s);
                        if (t == best_t) {
                            // the record of the hit so far was stored by this wave: read it back once the store has landed
                            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
                            const uint32_t bi = __hip_atomic_load(&hits[dst].instance, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const uint32_t bp = __hip_atomic_load(&hits[dst].primitive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const unsigned long long id = ((unsigned long long)inst << 32) | prim;
                            const unsigned long long best_id = ((unsigned long long)bi << 32) | bp;
                            closer = id < best_id;
                        }
                        if (closer) { PT_PWF(s, PW_BEST) = t; store_hit(dst, inst, prim, u, v, t); }
                    }
                }
                // one triangle per step: first + 1, count - 1
                if (code & 7u) node = (int)~(code + 7u);
                else {
                    PtPoolStack stk;
                    stk.lds = &PT_PW(s, PW_STACK); stk.spill = spill + (size_t)s * PT_SPILL_STACK; stk.sp = PT_PW(s, PW_SP);
                    const int v2 = stk.pop();
                    node = advance(v2, stk.sp);
                    PT_PW(s, PW_SP) = stk.sp;
                }
                PT_PW(s, PW_NODE) = node;
                new_phase = pt_pool_phase_of(node, inst);
            }
        } else if (phase == PT_PH_INST) {
            PT_STAT(2, 1); PT_STAT(5, n_sel);
            if (active) {
                int node = PT_PW(s, PW_NODE);
                PtPoolStack stk;
                stk.lds = &PT_PW(s, PW_STACK); stk.spill = spill + (size_t)s * PT_SPILL_STACK; stk.sp = PT_PW(s, PW_SP);
                PtRaySpace cur;
                uint32_t inst = PT_NO_INSTANCE;
                if (node == PT_NODE_SENTINEL) {        // back to world space (something is left on the stack): the ray as the queue holds it
                    const uint32_t qs = (uint32_t)PT_PW(s, PW_SLOT);
                    const float4 o = q0[qs], d = q1[qs];
                    pt_ray_space_init(cur, o.x, o.y, o.z, d.x, d.y, d.z);
                    PT_PWF(s, PW_OX) = cur.ox; PT_PWF(s, PW_OY) = cur.oy; PT_PWF(s, PW_OZ) = cur.oz;
                    PT_PWF(s, PW_DX) = cur.dx; PT_PWF(s, PW_DY) = cur.dy; PT_PWF(s, PW_DZ) = cur.dz;
                    PT_PWF(s, PW_IX) = cur.ix; PT_PWF(s, PW_IY) = cur.iy; PT_PWF(s, PW_IZ) = cur.iz;
                    PT_PW(s, PW_INST) = (int)PT_NO_INSTANCE; PT_PWF(s, PW_TLO) = tmin; PT_PWF(s, PW_THI) = tmax;
                    node = stk.pop();
                } else {
                    cur.ox = PT_PWF(s, PW_OX); cur.oy = PT_PWF(s, PW_OY); cur.oz = PT_PWF(s, PW_OZ);
                    cur.dx = PT_PWF(s, PW_DX); cur.dy = PT_PWF(s, PW_DY); cur.dz = PT_PWF(s, PW_DZ);
                    cur.ix = PT_PWF(s, PW_IX); cur.iy = PT_PWF(s, PW_IY); cur.iz = PT_PWF(s, PW_IZ);
                    cur.oox = cur.ox * cur.ix; cur.ooy = cur.oy * cur.iy; cur.ooz = cur.oz * cur.iz;
                }
                if (node < 0) {      // an instance leaf of the TLAS
                    const float best_t = PT_PWF(s, PW_BEST);
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
                            PtRaySpace obj;
                            pt_ray_space_init(obj, oox, ooy, ooz, odx, ody, odz);
                            PT_PWF(s, PW_OX) = obj.ox; PT_PWF(s, PW_OY) = obj.oy; PT_PWF(s, PW_OZ) = obj.oz;
                            PT_PWF(s, PW_DX) = obj.dx; PT_PWF(s, PW_DY) = obj.dy; PT_PWF(s, PW_DZ) = obj.dz;
                            PT_PWF(s, PW_IX) = obj.ix; PT_PWF(s, PW_IY) = obj.iy; PT_PWF(s, PW_IZ) = obj.iz;
                            PT_PWF(s, PW_DELTA) = blo.w;
                            inst = __float_as_uint(misc.y);
                            PT_PW(s, PW_INST) = (int)inst;
                            PT_PWF(s, PW_TLO) = fmaxf(tmin, a); PT_PWF(s, PW_THI) = fminf(tmax, b);
                            stk.push(PT_NODE_SENTINEL);
                            node = __float_as_int(misc.x);
                        }
                    }
                    if (!enter) node = advance(stk.pop(), stk.sp);
                }
                PT_PW(s, PW_NODE) = node; PT_PW(s, PW_SP) = stk.sp;
                new_phase = pt_pool_phase_of(node, inst);
            }
        } else {
            // ------------------------------------------------------------------ refill: new rays move into idle slots
            PT_STAT(7, 1);
            const unsigned long long act = __ballot(active);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));   // among the lanes with a slot
            uint32_t given = 0;
            while (given < (uint32_t)n_sel) {
                if (wcur == wend) {
                    bool got = false;
                    while (avail) {
                        // next shard with work at or after the wave's own
                        const unsigned long long rot = start ? ((avail >> start) | (avail << (64u - start))) : avail;
                        const uint32_t sh = ((uint32_t)__builtin_ctzll(rot) + start) & 63u;
                        const uint32_t n_s = __builtin_amdgcn_readlane(ns_lane, sh);
                        uint32_t* work = &Q.counters[Q.work_base + (size_t)sh * Q.ctr_pad];
                        const uint32_t chunk = min(PT_SCHED_CHUNK, max(64u, (n_s / waves_per_shard + 63u) & ~63u));
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(work, chunk);
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (base < n_s) {
                            wcur = sh * Q.shard_cap + base;
                            wend = sh * Q.shard_cap + (base + chunk < n_s ? base + chunk : n_s);
                            got = true;
                            break;
                        }
                        avail &= ~(1ull << sh);
                    }
                    if (!got) { drained = true; break; }
                }
                const uint32_t take = min((uint32_t)n_sel - given, wend - wcur);
                if (active && rank >= given && rank < given + take) {
                    const uint32_t slot = wcur + (rank - given);
                    const float4 o = pt_stream_load(&q0[slot]), d = pt_stream_load(&q1[slot]);
                    PtRaySpace cur;
                    pt_ray_space_init(cur, o.x, o.y, o.z, d.x, d.y, d.z);
                    PT_PWF(s, PW_OX) = cur.ox; PT_PWF(s, PW_OY) = cur.oy; PT_PWF(s, PW_OZ) = cur.oz;
                    PT_PWF(s, PW_DX) = cur.dx; PT_PWF(s, PW_DY) = cur.dy; PT_PWF(s, PW_DZ) = cur.dz;
                    PT_PWF(s, PW_IX) = cur.ix; PT_PWF(s, PW_IY) = cur.iy; PT_PWF(s, PW_IZ) = cur.iz;
                    PT_PWF(s, PW_BEST) = tmax; PT_PWF(s, PW_TLO) = tmin; PT_PWF(s, PW_THI) = tmax;
                    PT_PW(s, PW_INST) = (int)PT_NO_INSTANCE; PT_PW(s, PW_SLOT) = (int)slot;
                    const uint32_t dst = BY_ID ? __float_as_uint(d.w) : slot;
                    store_hit(dst, PT_NO_INSTANCE, 0xFFFFFFFFu, 0.0f, 0.0f, tmax);      // a miss until a triangle is hit
                    PtPoolStack stk;
                    stk.lds = &PT_PW(s, PW_STACK); stk.spill = spill + (size_t)s * PT_SPILL_STACK;
                    const int node = pt_ray_start(tl, stk);
                    PT_PW(s, PW_NODE) = node; PT_PW(s, PW_SP) = stk.sp;
                    new_phase = pt_pool_phase_of(node, PT_NO_INSTANCE);
                }
                wcur += take; given += take;
            }
        }
        // ------------------------------------------------------------------ the slots' new phases: class words and counts
        if (active && new_phase != phase) __hip_atomic_fetch_xor(&cls_words[cls], (int)((phase | new_phase) << (4 * k_slot)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int toB = __popcll(__ballot(active && new_phase == PT_PH_BOX)), toT = __popcll(__ballot(active && new_phase == PT_PH_TRI));
        const int toI = __popcll(__ballot(active && new_phase == PT_PH_INST)), toD = __popcll(__ballot(active && new_phase == PT_PH_IDLE));
        nB += toB - (phase == PT_PH_BOX ? n_sel : 0); nT += toT - (phase == PT_PH_TRI ? n_sel : 0);
        nI += toI - (phase == PT_PH_INST ? n_sel : 0); nD += toD - (phase == PT_PH_IDLE ? n_sel : 0);
    }
#undef PT_PW
#undef PT_PWF
}
