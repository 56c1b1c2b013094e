"""The adversarial inputs of tests/carry_runs.py do what they claim, checked on the CPU with the oracle:
the coded stream contains a run of run_len 0x00 bytes (0xFF bytes a carry went through), and it decodes."""
import re

import numpy as np
import pytest

import carry_runs


@pytest.mark.parametrize("run", [3, 40, 200])
def test_adaptive_carry_run(oracle, run):
    d = carry_runs.carry_run_block(4096, run, 7 + run)
    slots, sizes = oracle.encode_blocks(d, 4096)
    stream = bytes(slots[0, 5: int(sizes[0])])
    longest = max((len(m.group()) for m in re.finditer(b"\x00+", stream)), default=0)
    assert run <= longest <= run + 2
    back, ok = oracle.decode_blocks(slots, sizes, 4096, 4096)
    assert ok and np.array_equal(back, d)


@pytest.mark.parametrize("run", [5, 60])
def test_static_carry_run(oracle, run):
    d = None
    for seed in range(50):
        try:
            d = carry_runs.carry_run_block_static(8192, run, 1000 + seed)
            break
        except AssertionError:
            pass
    assert d is not None and np.all(np.bincount(d, minlength=256) == 32)
    slots, sizes = oracle.encode_blocks(d, 8192, coder=1)
    stream = bytes(slots[0, 517: int(sizes[0])])
    longest = max((len(m.group()) for m in re.finditer(b"\x00+", stream)), default=0)
    assert run <= longest <= run + 2
    back, ok = oracle.decode_blocks(slots, sizes, 8192, 8192, coder=1)
    assert ok and np.array_equal(back, d)
