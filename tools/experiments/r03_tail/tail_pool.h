// tail_pool.h -- device side of the dynamic tail of a persistent register-kernel launch (host side: tail_pool_for_launch in
// epilogue.hip, struct TailPool in spectro_internal.h).
//
// Why: every wave of the persistent grid starts within 0.3 us, but they do not finish together -- the oldest wave of a SIMD wins
// issue arbitration, XCDs and CUs differ -- the slowest wave needs ~13 % longer per frame than the median one, so with an even
// static split the launch lasts 8-14 % longer than its average wave (in-kernel stamps, profiles/r03_limiter.txt).
//
// What: work stealing at the END of the runs.  Wave w still owns one contiguous run [g0, g1) of the flattened (clip, frame) space.
// Its last `pieces * chunk` frames are a zone of `pieces` pieces behind one 32-bit word: low half = pieces the owner has claimed
// (from the front, in walking order, so its sample window keeps sliding and its rows keep streaming), high half = pieces thieves
// have claimed (from the back).  Every claim is one atomic add whose return value shows both halves as they were: a claim holds iff
// their sum is below `pieces`, so owner and thieves never get the same piece and every piece is claimed exactly once.  A wave whose
// own run is used up turns thief: it walks other waves' words in an order of its own and stops after `give_up` misses in a row.
// (A first version dealt the last 12 % of ALL frames from shared pools: balanced, but every piece reloaded its whole window --
// +50 % traffic in that phase -- and the launch did not get shorter; profiles/r03_ab_tail.txt.)
//
// Costs (tools/ubench/satomic.hip, profiles/r03_ubench_satomic.txt): one word takes ~87 atomics per us, so shared counters
// serialise the 3 072 waves (a single pool counter: +100 us); one word per wave has no contention.  Claims are s_atomic_add: scalar
// memory operations count in lgkmcnt, so a claim does not wait for the rows the wave has just stored, which a vector atomic does
// (vmcnt is one in-order queue of loads, stores and atomics: ~3 us per claim inside the frame loop).  Tickets are unique across
// XCDs (checked in the micro-benchmark).  Two sets of words alternate per pooled launch of a stream: every wave zeroes its share
// of the other set when it starts; nothing is reset at exit.
#pragma once
#include "spectro_internal.h"

namespace sg {

struct TailCursor { int victim, step, misses, own_left; };      // wave-uniform

__device__ __forceinline__ unsigned tail_add(unsigned* word, unsigned v) {
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(word) : "memory");
    return v;
}

__device__ __forceinline__ TailCursor tail_cursor(const TailPool& tp, int lw, int n_waves, int lane) {
    TailCursor c;
    c.step = __builtin_amdgcn_readfirstlane(static_cast<int>(((static_cast<unsigned>(lw) * 0x9E3779B1u) >> 8) % static_cast<unsigned>(n_waves > 1 ? n_waves - 1 : 1)) + 1);
    c.victim = lw;
    c.misses = 0;
    c.own_left = tp.pieces;
    if (tp.pieces > 0) {       // the other set, all kTailMaxWaves words of it (the next launch may have more waves than this one)
        const int word = lw + lane * n_waves;
        if (word < kTailMaxWaves) __hip_atomic_store(tp.words_next + static_cast<size_t>(word) * kTailStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return c;
}

// The owner asks for the next piece of its own zone.  -> true: the run may go on for `chunk` more frames.
__device__ __forceinline__ bool tail_claim_own(const TailPool& tp, TailCursor& c, int lw) {
    if (c.own_left <= 0) return false;
    const unsigned old = tail_add(tp.words + static_cast<size_t>(lw) * kTailStride, 1u);
    if ((old & 0xffffu) + (old >> 16) < static_cast<unsigned>(tp.pieces)) { --c.own_left; return true; }
    c.own_left = 0;                                   // thieves have taken the rest
    return false;
}

// A wave without work of its own looks for a piece elsewhere.  -> true: *victim_wave's piece number *piece_from_back (0 = its last
// `chunk` frames) is this wave's.
__device__ __forceinline__ bool tail_steal(const TailPool& tp, TailCursor& c, int lw, int n_waves, int* victim_wave, int* piece_from_back) {
    while (c.misses <= tp.give_up) {
        if (c.victim != lw) {
            const unsigned old = tail_add(tp.words + static_cast<size_t>(c.victim) * kTailStride, 0x10000u);
            if ((old & 0xffffu) + (old >> 16) < static_cast<unsigned>(tp.pieces)) {
                c.misses = 0;
                *victim_wave = c.victim;
                *piece_from_back = static_cast<int>(old >> 16);
                return true;                          // (the next look starts at the same victim)
            }
            ++c.misses;
        }
        c.victim += c.step;
        if (c.victim >= n_waves) c.victim -= n_waves;
    }
    return false;
}

}  // namespace sg
