"""CPU: the index algebra of the screening kernel (csrc/screen.hip: lane -> cell / planes, the DPP row moves, the B fragment packing, the
rotating accumulator slots, the emission coordinates), emulated lane by lane in numpy (tools/emulate_screen_walk.py): every window sum a
strip of 1..4 groups emits equals the direct sum over the window."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import emulate_screen_walk as emu  # noqa: E402


@pytest.mark.parametrize("ng", [1, 2, 3, 4])
def test_strip_walk_emits_every_window_sum(ng):
    n, expected, bad = emu.check(ng, fh=12, extra=5, seed=ng)
    assert n == expected and n > 0 and bad == 0
