// rcx_divtab.hpp -- construction of divisor-table entries (host: the adaptive coder's table;
// device: the static coder's one divisor per block).
//
// Entry i serves total = 256 + i: the i-th symbol of a block sees exactly that total
// (cpprcoder.h:1096 init to 256, :1138 +1 per symbol) as long as no halving happens,
// so range / total (cpprcoder.h:703, :904) is a division by a value every lane of a
// wave shares and that is known ahead of time.
#pragma once
#include "rcx_lane.hpp"

// floor(n / d) == (u32)(((u64)n * mul + add) >> 32) >> shift for every n < 2^32
// (N-bit multiply-add division, Robison 2005: with k = 32 + floor(log2 d) either the
//  rounded-up reciprocal fits 32 bits and is exact, or the rounded-down one applied
//  to n+1 is).
RCX_HD DivEntry rcx_make_div_entry(u32 d)
{
    DivEntry e;
    e.total = d;
    u32 s = 31u - (u32)__builtin_clz(d);
    e.shift = s;
    if ((d & (d - 1)) == 0) { // power of two: ((n+1)*(2^32-1)) >> 32 == n
        e.mul = 0xFFFFFFFFu;
        e.add = 0xFFFFFFFFu;
        return e;
    }
    const u64 pow = (u64)1 << (32 + s); // d is not a power of two, so s <= 30 and this fits
    const u64 down = pow / d;
    const u64 rem = pow - down * d;
    if ((u64)d - rem <= ((u64)1 << s)) { // round-up magic fits and is exact
        e.mul = (u32)(down + 1);
        e.add = 0;
    } else { // round-down magic on n+1
        e.mul = (u32)down;
        e.add = (u32)down;
    }
    return e;
}
