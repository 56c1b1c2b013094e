"""Inputs whose adaptive coding holds back a long run of 0xFF output bytes and then carries through it.

The reference keeps such a run as a counter (cpprcoder.h:767-800: carry_); a coder that writes bytes eagerly
has to walk back through them.  The generator runs the coder arithmetic (cpprcoder.h:703-711, 764-802) on the
side and steers the interval along the wrap point of `low`: as long as [low, low + range) straddles 2^32 it picks
the symbol whose sub-interval contains that point (every renormalisation then emits 0xFF), and after `run_len`
such bytes a symbol that lies entirely above it (the carry).  Everything else is seeded random bytes.
"""
import numpy as np

M = 1 << 32


def carry_run_block(n: int, run_len: int, seed: int, after: int = 0) -> np.ndarray:
    """`after`: the first symbol index from which the interval may be steered (the run then begins at the next point at which
    the interval straddles the wrap point by itself: somewhere later in the block, at no particular output position)."""
    rs = np.random.RandomState(seed)
    cnt = [1] * 256
    total = 256
    low, rng = 0, 0xFFFFFF00
    out = []
    held = 0          # 0xFF bytes emitted while the interval straddles the wrap point
    done = False
    while len(out) < n:
        t = rng // total
        c = None
        if not done and len(out) >= after and low + total * t > M:  # [low, low + total*t) contains the wrap point
            cum = 0
            for s in range(256):
                lo_s, hi_s = low + cum * t, low + (cum + cnt[s]) * t
                if held < run_len and lo_s < M <= hi_s:
                    c = s      # stay on the wrap point
                    break
                if held >= run_len and lo_s >= M:
                    c = s      # entirely above it: this symbol carries
                    done = True
                    break
                cum += cnt[s]
        if c is None:
            c = int(rs.randint(256))
            if not done:
                held = 0
        cum = sum(cnt[:c])
        low = (low + cum * t) % M
        rng = cnt[c] * t
        while rng < (1 << 24):
            if (low >> 24) == 0xFF:
                held += 1
            low = (low << 8) % M
            rng <<= 8
        cnt[c] += 1
        total += 1
        out.append(c)
    assert done, "no carry run found: try another seed"
    return np.array(out, np.uint8)


def carry_run_block_static(n: int, run_len: int, seed: int) -> np.ndarray:
    """The same for the static coder (cpprcoder.h:400-436: fixed table, range starts at 0xFFFFFFFF).  The block
    uses every byte value n / 256 times, so the table is known before the symbols are chosen."""
    assert n % 256 == 0 and n // 256 < 0xFFFF
    rs = np.random.RandomState(seed)
    per = n // 256
    left = [per] * 256
    total = n
    low, rng = 0, 0xFFFFFFFF
    out = []
    held = 0
    done = False
    while len(out) < n:
        t = rng // total
        c = None
        if not done and low + total * t > M:
            for s in range(256):
                lo_s, hi_s = low + s * per * t, low + (s + 1) * per * t
                if held < run_len and lo_s < M <= hi_s and left[s]:
                    c = s
                    break
                if held >= run_len and lo_s >= M and left[s]:
                    c = s
                    done = True
                    break
        if c is None:
            pool = [s for s in range(256) if left[s]]
            c = pool[int(rs.randint(len(pool)))]
            if not done:
                held = 0
        left[c] -= 1
        low = (low + c * per * t) % M
        rng = per * t
        while rng < (1 << 24):
            if (low >> 24) == 0xFF:
                held += 1
            low = (low << 8) % M
            rng <<= 8
        out.append(c)
    assert done, "no carry run found: try another seed"
    return np.array(out, np.uint8)
